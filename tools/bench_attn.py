"""micro-benchmark of the attention core (forward, backward) through the C ABI: python tools/bench_attn.py [B L heads rate]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4rec_amd import _lib
lib = _lib.load()
B, L, heads = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (256, 200, 2)))
rate = float(sys.argv[4]) if len(sys.argv) > 4 else 0.2
H = 32 * heads
qkv = torch.randn(B * L, 3 * H, device="cuda") * 0.5
mask = torch.ones(B, L, dtype=torch.int64, device="cuda")
ctx = torch.empty(B * L, H, device="cuda"); lse = torch.empty(B * heads * L, device="cuda")
dctx = torch.randn(B * L, H, device="cuda"); dqkv = torch.empty(B * L, 3 * H, device="cuda")
bits = torch.empty(lib.b4r_attn_keep_words(B, L, heads), dtype=torch.int32, device="cuda")
state = torch.zeros(16, dtype=torch.int32, device="cuda"); state[0] = 1234
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
fwd = lambda: _lib.check(lib.b4r_attn_fwd(P(qkv), P(mask), B, L, heads, P(ctx), P(lse), P(state), 1, rate, P(bits), st), "fwd")
bwd = lambda: _lib.check(lib.b4r_attn_bwd(P(qkv), P(mask), P(ctx), P(lse), P(dctx), B, L, heads, 0.1767767, P(dqkv), P(state), 1, rate, P(bits), st), "bwd")
def timeit(f, reps=100):
    for _ in range(10): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print("stagger", os.environ.get("B4R_ATTN_STAGGER", "0"), "forward %.1f us  backward (dq + dkv) %.1f us  ctx checksum %.6f" % (timeit(fwd), timeit(bwd), float(ctx.double().abs().sum())))
