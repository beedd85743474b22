"""micro-benchmark of one dense product through b4r_gemm_f32: python tools/bench_gemm.py M N K b_is_nk epi"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4rec_amd import _lib
lib = _lib.load()
M, N, K, nk, epi = (int(x) for x in sys.argv[1:6])
A = torch.randn(M, K, device="cuda"); B = torch.randn((N, K) if nk else (K, N), device="cuda") * 0.05
bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda"); out2 = torch.empty(M, N, device="cuda")
R = torch.randn(M, N, device="cuda")
d = _lib.GemmDesc(); d.A, d.lda, d.B, d.ldb, d.C, d.ldc = A.data_ptr(), K, B.data_ptr(), (K if nk else N), out.data_ptr(), N
d.M, d.N, d.K, d.b_is_nk, d.epilogue, d.bias = M, N, K, nk, epi, bias.data_ptr()
d.C2, d.ldc2, d.R, d.ldr, d.qscale, d.c_pad_scratch = out2.data_ptr(), N, R.data_ptr(), N, 1.0, 1
st = torch.cuda.current_stream().cuda_stream
def run(reps=50):
    for _ in range(5): _lib.check(lib.b4r_gemm_f32(C.byref(d), st), "gemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.b4r_gemm_f32(C.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
t = run()
print(f"M={M} N={N} K={K} nk={nk} epi={epi}: {t:.1f} us  ({t / M * 1e3:.3f} ns/row)")
