"""micro-benchmark of the fused feed-forward block (b4r_ffn_block_fwd / _bwd) at the ML-1M token count"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4rec_amd import _lib
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
H, I = 64, 256
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
x1, z1, dz2 = r(N, H), r(N, H), r(N, H)
W1, b1, W2, b2 = r(H, I, sc=0.1), r(I, sc=0.1), r(I, H, sc=0.1), r(H, sc=0.1)
g1, g2, be2 = 1 + r(H, sc=0.1), 1 + r(H, sc=0.1), r(H, sc=0.1)
mean1, rstd1 = r(N, sc=0.1), 1 + r(N, sc=0.1).abs()
z2, x2, dz1 = (torch.empty(N, H, device="cuda") for _ in range(3))
mean2, rstd2 = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
dW1, db1, dW2, db2, dln = torch.empty(H, I, device="cuda"), torch.empty(I, device="cuda"), torch.empty(I, H, device="cuda"), torch.empty(H, device="cuda"), torch.empty(128, device="cuda")
scratch = torch.empty(lib.b4r_ffn_block_bwd_scratch_floats(N), device="cuda")
state = torch.zeros(16, dtype=torch.int32, device="cuda"); state[0] = 77
P = lambda t: t.data_ptr()
d = _lib.FfnDesc()
d.N, d.H, d.I = N, H, I
d.x1, d.W1, d.b1, d.W2, d.b2 = P(x1), P(W1), P(b1), P(W2), P(b2)
d.ln_gamma, d.ln_beta, d.ln_eps = P(g2), P(be2), 1e-12
d.rng, d.drop_stream, d.drop_rate = (P(state) if rate > 0 else None), 3, rate
d.z2, d.x2, d.mean2, d.rstd2 = P(z2), P(x2), P(mean2), P(rstd2)
d.dz2, d.z1, d.mean1, d.rstd1, d.ln1_gamma = P(dz2), P(z1), P(mean1), P(rstd1), P(g1)
d.dz1, d.dW1, d.db1, d.dW2, d.db2, d.dln1_gamma, d.scratch = P(dz1), P(dW1), P(db1), P(dW2), P(db2), P(dln), P(scratch)
st = torch.cuda.current_stream().cuda_stream
fwd = lambda: _lib.check(lib.b4r_ffn_block_fwd(C.byref(d), st), "fwd")
bwd = lambda: _lib.check(lib.b4r_ffn_block_bwd(C.byref(d), st), "bwd")
def timeit(f, reps=200):
    for _ in range(20): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print("N %d rate %.2f  ffn forward %.1f us  backward (dx + dw + 3 reductions) %.1f us" % (N, rate, timeit(fwd), timeit(bwd)))
if hasattr(lib, "b4r_debug_ff_prof"):   # -DFFN_PROF build: phase stamps of workgroup 0, thread 0
    fwd(); bwd(); torch.cuda.synchronize()
    buf = (C.c_longlong * 64)()
    lib.b4r_debug_ff_prof.argtypes = [C.c_void_p]
    assert lib.b4r_debug_ff_prof(buf) == 0
    t = list(buf)
    names = {0: "fwd start", 1: "fwd weights staged", 2: "fwd barrier", 3: "fwd x rows loaded", 4: "fwd kt loop", 5: "fwd residual + LN stats", 6: "fwd stores issued", 7: "fwd end",
             10: "dx start", 11: "dx staged + barrier", 12: "dx rows loaded", 13: "dx kt loop", 14: "dx LN' sums", 15: "dx stores issued", 16: "dx barrier",
             30: "dw start", 31: "dw operands in registers", 32: "dw first chunk staged", 33: "dw chunk 0", 34: "dw chunk 1", 35: "dw chunk 2", 36: "dw chunk 3",
             37: "dw chunk 4", 38: "dw chunk 5", 39: "dw chunk 6", 40: "dw chunk 7", 41: "dw loop done", 42: "dw end"}
    for base in (0, 10, 30):
        prev = t[base]
        for k in sorted(k for k in names if base <= k < base + 20 and t[k] >= t[base]):
            print("%-28s +%7d cycles  (total %8d)" % (names[k], t[k] - prev, t[k] - t[base]))
            prev = t[k]
