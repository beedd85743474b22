import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import synthetic_batch, CONFIGS
from bert4rec_amd.engine import Engine, make_adamw_config, make_model_config
V, H, NL, NH, I, L, P, B, od, ad, rate = CONFIGS["ml1m"]
eng = Engine(make_model_config(V, H, NL, NH, L, I, od, ad), "cuda", seed=1)
eng.init_parameters(seed=3)
hp = make_adamw_config()
cb, keep = eng.prepare_batch(synthetic_batch(B, L, P, V, rate, seed=0))
for name, fn in (("train_step", lambda: eng.train_step(hp, cb)), ("dp_train_step", lambda: eng.dp_train_step(hp, cb))):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): fn()
    t_host = time.perf_counter() - t0          # enqueue time (the queue may throttle it)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{name}: host enqueue {t_host / 200 * 1e3:.3f} ms/step, wall {t_all / 200 * 1e3:.3f} ms/step")
# pure host cost: enqueue with the GPU idle-ish is not separable here; measure python+ctypes overhead with a tiny model
