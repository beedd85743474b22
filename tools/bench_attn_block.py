"""micro-benchmark of the fused attention block (b4r_attn_block_fwd / _bwd) at the ML-1M shape"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4rec_amd import _lib
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rate = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
H, N = 64, B * L
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
x, dz1, zprev = r(N, H), r(N, H), r(N, H)
Wqkv, bqkv, Wo, bo = r(H, 3 * H, sc=0.1), r(3 * H, sc=0.1), r(H, H, sc=0.1), r(H, sc=0.1)
g1, be1 = 1 + r(H, sc=0.1), r(H, sc=0.1)
meanp, rstdp = r(N, sc=0.1), 1 + r(N, sc=0.1).abs()
mask = torch.ones(B, L, dtype=torch.int64, device="cuda")
ctx, z1, x1, da = (torch.empty(N, H, device="cuda") for _ in range(4))
lse = torch.empty(B, 2, L, device="cuda"); dqkv = torch.empty(N, 3 * H, device="cuda"); dln = torch.empty(128, device="cuda")
bits = torch.zeros(lib.b4r_attn_keep_words(B, L, 2), dtype=torch.int32, device="cuda")
scratch = torch.empty(lib.b4r_attn_block_bwd_scratch_floats(B), device="cuda")
state = torch.zeros(16, dtype=torch.int32, device="cuda"); state[0] = 77
P = lambda t: t.data_ptr()
rng = P(state) if rate > 0 else None
fd = _lib.AttnBlockDesc()
fd.B, fd.L, fd.H, fd.heads, fd.x, fd.input_mask = B, L, H, 2, P(x), P(mask)
fd.Wqkv, fd.bqkv, fd.Wo, fd.bo, fd.ln_gamma, fd.ln_beta, fd.ln_eps = P(Wqkv), P(bqkv), P(Wo), P(bo), P(g1), P(be1), 1e-12
fd.rng, fd.probs_stream, fd.probs_rate, fd.out_stream, fd.out_rate = rng, 1, rate, 2, rate
fd.qkv, fd.ctx, fd.lse, fd.keep_bits, fd.z1, fd.x1 = None, P(ctx), P(lse), P(bits), P(z1), P(x1)
bd = _lib.AttnBlockBwdDesc()
bd.B, bd.L, bd.H, bd.heads = B, L, H, 2
bd.x, bd.dz1, bd.ctx, bd.lse, bd.keep_bits, bd.input_mask = P(x), P(dz1), P(ctx), P(lse), P(bits), P(mask)
bd.Wqkv, bd.bqkv, bd.Wo = P(Wqkv), P(bqkv), P(Wo)
bd.rng, bd.probs_stream, bd.probs_rate, bd.out_stream, bd.out_rate = rng, 1, rate, 2, rate
bd.prev_z, bd.prev_mean, bd.prev_rstd, bd.prev_gamma = P(zprev), P(meanp), P(rstdp), P(g1)
bd.dqkv, bd.dx_prev, bd.dprev_gamma, bd.scratch = P(dqkv), P(da), P(dln), P(scratch)
st = torch.cuda.current_stream().cuda_stream
fwd = lambda: _lib.check(lib.b4r_attn_block_fwd(C.byref(fd), st), "fwd")
bwd = lambda: _lib.check(lib.b4r_attn_block_bwd(C.byref(bd), st), "bwd")
def timeit(f, reps=100):
    for _ in range(10): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print("B %d L %d rate %.2f  attention block forward %.1f us  backward (+ LayerNorm partial reduce) %.1f us" % (B, L, rate, timeit(fwd), timeit(bwd)))
if hasattr(lib, "b4r_debug_ab_prof"):   # a -DAB_PROF build: phase stamps of workgroup 0 (shader clock cycles -> us at 2.4 GHz is only a guess: print ratios too)
    bwd(); torch.cuda.synchronize()
    buf = (C.c_longlong * 64)()
    lib.b4r_debug_ab_prof.argtypes = [C.c_void_p]
    assert lib.b4r_debug_ab_prof(buf) == 0
    t = list(buf)
    names = {0: "start", 1: "h0 region free", 2: "h0 weights staged", 3: "h0 dX of prev head", 4: "h0 qkv+dctx recomputed", 5: "h0 loads+barrier",
             6: "h0 images written+barrier", 7: "h0 sweep done", 11: "h1 region free (dqkv stored)", 12: "h1 weights staged", 13: "h1 dX",
             14: "h1 qkv+dctx", 15: "h1 loads+barrier", 16: "h1 images+barrier", 17: "h1 sweep done", 21: "tail region free", 22: "tail weights staged",
             23: "tail dX", 30: "before epilogue", 32: "epi: loads + row sums", 33: "epi: da stored", 34: "epi: column sums shuffled", 35: "epi: barrier", 31: "end"}
    prev = t[0]
    for k in sorted(names, key=lambda k: t[k]):
        print("%-32s +%7d cycles  (total %8d)" % (names[k], t[k] - prev, t[k] - t[0]))
        prev = t[k]
if hasattr(lib, "b4r_debug_a32_prof"):   # a -DA32_PROF build of b4r_attn32.hip: phase stamps of wave 0 of workgroup 0
    bwd(); torch.cuda.synchronize()
    buf = (C.c_longlong * 64)()
    lib.b4r_debug_a32_prof.argtypes = [C.c_void_p]
    assert lib.b4r_debug_a32_prof(buf) == 0
    t = list(buf)
    names = {0: "start"}
    for hd in (0, 1):
        for k, n in ((1, "x / dz1 loads issued"), (2, "barrier (region free)"), (3, "weights staged"), (4, "barrier"), (5, "q k v dctx projected"),
                     (6, "images / D / K^T written"), (7, "barrier"), (8, "sweep done"), (9, "dqkv stored + dX")):
            names[k + 10 * hd] = "h%d %s" % (hd, n)
    names.update({30: "epilogue start", 31: "epilogue: loads + column sums", 32: "end"})
    prev = t[0]
    for k in sorted(names, key=lambda k: t[k]):
        print("%-36s +%7d cycles  (total %8d)" % (names[k], t[k] - prev, t[k] - t[0]))
        prev = t[k]
if hasattr(lib, "b4r_debug_a32f_prof"):   # phase stamps of wave 0 of workgroup 0 of the forward
    fwd(); torch.cuda.synchronize()
    buf = (C.c_longlong * 32)()
    lib.b4r_debug_a32f_prof.argtypes = [C.c_void_p]
    assert lib.b4r_debug_a32f_prof(buf) == 0
    t = list(buf)
    names = {0: "start", 1: "x rows loaded / formed", 2: "weights + x images staged", 3: "barrier", 4: "q k v of both heads", 5: "barrier",
             6: "K / V images written", 7: "barrier", 8: "h0 scores", 9: "h0 softmax", 12: "h1 scores (after h0 dropout + P.V)", 13: "h1 softmax",
             16: "h1 dropout + P.V", 17: "output projection", 18: "barrier", 19: "ctx out (row layout)", 20: "end"}
    prev = t[0]
    for k in sorted(names, key=lambda k: t[k]):
        print("fwd %-36s +%7d cycles  (total %8d)" % (names[k], t[k] - prev, t[k] - t[0]))
        prev = t[k]
if hasattr(lib, "b4r_debug_a32_sweep"):
    buf = (C.c_longlong * 128)()
    lib.b4r_debug_a32_sweep.argtypes = [C.c_void_p]
    assert lib.b4r_debug_a32_sweep(buf) == 0
    t = list(buf)
    NT = (L + 31) // 32
    for w in (0, 1):
        tw = t[64 * w:64 * w + 64]
        print("sweep of head 0, wave %d: m12 %d" % (4 * w, tw[1] - tw[0]))
        for s in range(NT):
            a = tw[2 + 4 * s:6 + 4 * s]
            nxt = tw[2 + 4 * (s + 1)] if 2 + 4 * (s + 1) < 64 else a[3]
            print("   step %d: vector phase %6d  matrix phase issue %6d  flag wait + accumulate %6d  (-> next step %6d)   (start %d)" % (s, a[1] - a[0], a[2] - a[1], a[3] - a[2], nxt - a[3], a[0] - t[0]))
# the whole layer forward (attention half + feed-forward half in one launch)
I = 256
W1, b1, W2, b2 = r(H, I, sc=0.1), r(I, sc=0.1), r(I, H, sc=0.1), r(H, sc=0.1)
g2, be2 = 1 + r(H, sc=0.1), r(H, sc=0.1)
z2, x2 = torch.empty(N, H, device="cuda"), torch.empty(N, H, device="cuda")
mean1, rstd1, mean2, rstd2 = (torch.empty(N, device="cuda") for _ in range(4))
fd.mean1, fd.rstd1 = P(mean1), P(rstd1)
ff = _lib.FfnDesc()
ff.N, ff.H, ff.I = N, H, I
ff.x1, ff.W1, ff.b1, ff.W2, ff.b2 = P(x1), P(W1), P(b1), P(W2), P(b2)
ff.ln_gamma, ff.ln_beta, ff.ln_eps = P(g2), P(be2), 1e-12
ff.rng, ff.drop_stream, ff.drop_rate = rng, 3, rate
ff.z2, ff.x2, ff.mean2, ff.rstd2 = P(z2), P(x2), P(mean2), P(rstd2)
layer = lambda: _lib.check(lib.b4r_encoder_layer_fwd(C.byref(fd), C.byref(ff), st), "layer fwd")
ffn_only = lambda: _lib.check(lib.b4r_ffn_block_fwd(C.byref(ff), st), "ffn fwd")
print("whole layer forward %.1f us   (attention half alone %.1f us, feed-forward half alone %.1f us)" % (timeit(layer), timeit(fwd), timeit(ffn_only)))
if hasattr(lib, "b4r_debug_af_prof"):
    layer(); torch.cuda.synchronize()
    buf = (C.c_longlong * 16)()
    lib.b4r_debug_af_prof.argtypes = [C.c_void_p]
    assert lib.b4r_debug_af_prof(buf) == 0
    t = list(buf)
    names = ["start", "x loaded, weights staged", "barrier", "qkv computed", "barrier", "K/V images written", "barrier", "head 0 start", "head 1 start",
             "attention done", "ctx stored, out-proj done", "residual + LN stats"]
    for k in range(1, 13):
        print("fwd %-28s +%7d cycles  (total %8d)" % (names[k], t[k] - t[k - 1], t[k] - t[0]))
