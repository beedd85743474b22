"""micro-benchmark of the materialising MLM-head projection (the roofline kernel) under experiment switches"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4rec_amd import _lib
lib = _lib.load()
M, V, H, Vp = 10240, 3709, 64, 3712
T = torch.randn(M, H, device="cuda"); E = torch.randn(V, H, device="cuda") * 0.05; b = torch.randn(V, device="cuda")
out = torch.empty(M, Vp, device="cuda")
d = _lib.GemmDesc(); d.A, d.lda, d.B, d.ldb, d.C, d.ldc = T.data_ptr(), H, E.data_ptr(), H, out.data_ptr(), Vp
d.M, d.N, d.K, d.b_is_nk, d.epilogue, d.bias = M, V, H, 1, _lib.EPI_BIAS, b.data_ptr()
d.c_pad_scratch = 1
st = torch.cuda.current_stream().cuda_stream
def run(mode, reps=50):
    lib.b4r_set_gemm_mode(mode)
    for _ in range(5): lib.b4r_gemm_f32(C.byref(d), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.b4r_gemm_f32(C.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print("variant", os.environ.get("B4R_RX_VARIANT", "0"), "target", os.environ.get("B4R_RX_TARGET", "-"), "f32 %.1f us  bf16x3 %.1f us" % (run(0), run(1)))
# plain fill of the same buffer for reference
x = torch.empty_like(out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): x.fill_(1.0)
e0.record()
for _ in range(20): x.fill_(1.0)
e1.record(); torch.cuda.synchronize(); print("torch fill of 152 MB: %.1f us" % (e0.elapsed_time(e1) * 1e3 / 20))
