import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import synthetic_batch, CONFIGS
from bert4rec_amd.engine import Engine, make_adamw_config, make_model_config
V, H, NL, NH, I, L, P, B, od, ad, rate = CONFIGS["ml1m"]
def mk():
    eng = Engine(make_model_config(V, H, NL, NH, L, I, od, ad), "cuda", seed=1)
    eng.init_parameters(seed=3)
    return eng
hp = make_adamw_config()
batch = synthetic_batch(B, L, P, V, rate, seed=0)
e1, e2 = mk(), mk()
cb1, k1 = e1.prepare_batch(batch); cb2, k2 = e2.prepare_batch(batch)
# eager reference: 13 steps
for _ in range(13): e1.train_step(hp, cb1)
torch.cuda.synchronize()
# graphed: 3 eager warm-up steps on a side stream, capture 1, replay 9
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): e2.train_step(hp, cb2)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    e2.train_step(hp, cb2)
torch.cuda.synchronize()
# capture does not execute: state is after 3 steps; 10 replays -> 13 steps
for _ in range(10): g.replay()
torch.cuda.synchronize()
s1, s2 = e1.read_state(), e2.read_state()
print("steps", s1["step"], s2["step"], "loss", s1["loss_sum"], s2["loss_sum"], "max param diff", float((e1.params - e2.params).abs().max()))
t0 = time.perf_counter()
for _ in range(300): g.replay()
th = time.perf_counter() - t0
torch.cuda.synchronize()
ta = time.perf_counter() - t0
print(f"graph replay: host {th / 300 * 1e3:.3f} ms/step, wall {ta / 300 * 1e3:.3f} ms/step")
