"""The forward / evaluation API path of the library for profiling runs (tools/prof_api.sh): b4r_forward with MATERIALISED masked-LM logits
(what model(batch) returns: rx_gemm_nk_kernel<BIAS> writes the [B*P, V] tensor) and the evaluator's ranking path, on the synthetic
S-full batch of a bench.py configuration.  The train step never runs these kernels, so tools/prof.sh never sees them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from bert4rec_amd.engine import Engine, make_model_config

name = sys.argv[1] if len(sys.argv) > 1 else "ml1m"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
V, H, NL, NH, I, L, P, B, od, ad, rate = bench.CONFIGS[name]
dev = torch.device("cuda", 0)
eng = Engine(make_model_config(V, H, NL, NH, L, I, od, ad), dev, seed=1)
eng.init_parameters(seed=3)
cb, keep = eng.prepare_batch(bench.synthetic_batch(B, L, P, V, rate, seed=0))
for _ in range(3):
    eng.forward(cb, training=False, pooler=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    eng.forward(cb, training=False, pooler=True)
torch.cuda.synchronize()
print(f"{name}: b4r_forward with materialised logits {(time.perf_counter() - t0) / reps * 1e6:.1f} us per call")
# the ranking path on the same batch: transform of the ranked rows + candidate ranking (101 candidates per slot)
seq = eng.region("sequence_output", B, L, P)
rows = (torch.arange(B, device=dev) * L + (L - 1)).to(torch.int64)
hidden = eng.mlm_transform_rows(seq, rows)
cand = torch.randint(3, V, (B, 101), device=dev)
gt = cand[:, 100].clone()
for _ in range(reps):
    eng.rank_candidates(hidden, None, cand, gt, want_ranking=False)
torch.cuda.synchronize()
