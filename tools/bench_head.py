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
