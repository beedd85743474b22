"""micro-benchmark of the fused masked-LM head (vocabulary sweep alone, whole forward, backward) under experiment switches"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4rec_amd import _lib
lib = _lib.load()
M, V, H = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (10240, 3709, 64)))
g = torch.Generator(device="cuda").manual_seed(1)
T = torch.randn(M, H, device="cuda", generator=g); E = torch.randn(V, H, device="cuda", generator=g) * 0.3
b = torch.randn(V, device="cuda", generator=g) * 0.1
y = torch.randint(1, V, (M,), device="cuda", generator=g)
scratch = torch.empty(lib.b4r_mlm_head_fused_scratch_floats(M, V, H), device="cuda")
dT = torch.empty(M, H, device="cuda"); rows = torch.empty(4 * M, device="cuda"); lse = torch.empty(M, device="cuda")
lab = torch.empty(M, dtype=torch.int32, device="cuda"); dE = torch.empty(V, H, device="cuda"); db = torch.empty(V, device="cuda")
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
fwd = lambda only: _lib.check(lib.b4r_mlm_head_fused_fwd(P(T), P(E), P(b), P(y), M, V, H, P(scratch), P(dT), P(rows), P(lse), P(lab), only, st), "fwd")
bwd = lambda: _lib.check(lib.b4r_mlm_head_fused_bwd(P(T), P(E), P(b), P(lse), P(lab), M, V, H, P(scratch), P(dE), P(db), st), "bwd")
def timeit(f, reps=100):
    for _ in range(10): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
fwd(0)
ref = torch.logsumexp(T.double() @ E.double().t() + b.double(), 1)
err = float((lse.double() - ref).abs().max())
print("occ", os.environ.get("B4R_HEAD_OCC", "-"), "fwd_wgs", os.environ.get("B4R_HEAD_FWD_WGS", "-"),
      "sweep %.1f us  forward %.1f us  backward %.1f us  lse err %.2e" % (timeit(lambda: fwd(1)), timeit(lambda: fwd(0)), timeit(bwd), err))

# per-launch times of one forward + one backward from the library's event timer
import ctypes as C
for name, f in (("forward", lambda: fwd(0)), ("backward", bwd)):
    torch.cuda.synchronize()
    _lib.check(lib.b4r_timing_begin(st, 64 * 20), "timing")
    for _ in range(20): f()
    n = C.c_int32(0); us = (C.c_float * (64 * 20))(); names = C.create_string_buffer(64 * 20 * 128)
    _lib.check(lib.b4r_timing_end(C.byref(n), us, names, 128, 64 * 20), "timing")
    per = n.value // 20
    rows = [(names.raw[j * 128:(j + 1) * 128].split(b"\0", 1)[0].decode(), sum(us[s_ * per + j] for s_ in range(20)) / 20) for j in range(per)]
    print(f"   {name}: " + " | ".join(f"{k} {v:.1f} us" for k, v in rows))
if hasattr(lib, "b4r_debug_h32_prof"):   # a -DH32_PROF build of b4r_head32.hip: shader-clock stamps of waves 0 / 5 of two workgroups of the forward
    import ctypes as C
    fwd(1); torch.cuda.synchronize()
    buf = (C.c_longlong * 512)()
    lib.b4r_debug_h32_prof.argtypes = [C.c_void_p]
    assert lib.b4r_debug_h32_prof(buf) == 0
    for w in range(4):
        t = [buf[128 * w + k] for k in range(128)]
        base = t[0]
        print(f"-- workgroup {'0' if w < 2 else '7'}, wave {'0' if w % 2 == 0 else '5'}: loads {t[1]-t[0]}, to first barrier {t[3]-t[1]} (wait+barrier {t[3]-t[2]})")
        its = []
        for i in range(28):
            a, b, c, d = t[4 + 4 * i: 8 + 4 * i]
            nxt = t[8 + 4 * i] if 8 + 4 * i < 116 and t[8 + 4 * i] > 0 else t[120]
            if a <= 0: break
            its.append((b - a, c - b, d - c, nxt - d))
        print("   per step (vmcnt wait, barrier, block A, main block):", its)
        print(f"   loop total {t[120]-t[3]}, last products {t[121]-t[120]}, epilogue {t[122]-t[121]}, kernel {t[122]-t[0]}")
