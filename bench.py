#!/usr/bin/env python3
"""Benchmark of the BERT4Rec hot path on MI355X: full train steps (forward + masked CE + backward + clip + AdamW
[+ RCCL all-reduce]) on the ML-1M configuration of BASELINE.json (configs[1]):
B=256 per GPU, L=200, P=40, H=64, 2 layers, 2 heads, inner 256, V=3709, dropout 0.2/0.2, full-vocab masked-LM head.

    python bench.py --gpus 1 --steps 200 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` = masked positions (slots with masked_lm_ids != 0) consumed per second by the
whole job, inputs resident in HBM.

`roofline`: after the timed region a few more steps run under the library's launch timer (b4r_timing_begin / _end: a hipEvent on
the launch stream behind every kernel of the step).  The kernel with the largest time per step IS the roofline kernel -- it is
picked from the measurement, not named in advance -- and `achieved` = its algorithmic bytes (or FLOPs) per launch / its average
launch time.  `step_breakdown` lists every launch of one step, `step_hbm` puts the whole step against the HBM roof.
`roofline_materialising` keeps the HBM-bound logits[M,V] = T.E^T + b kernel of the forward / evaluation API (the north star's
">= 40 % of HBM roofline on the masked-LM head"; not part of the timed train step, which never writes logits).
`eval`: BASELINE.json's metric also names NDCG@10 -- a short train + evaluate run on a synthetic Zipf log with learnable
successor structure (labelled as synthetic), through the product's dataloader / trainer / evaluator surface, with
evaluation throughput in users/s.
`cpu_baseline`: the oracle (CPU restatement of the reference math; TF2 is not installed anywhere) timed on the host.
"""
import argparse
import collections
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

CONFIGS = {
    # name: (vocab, hidden, layers, heads, inner, L, P, B, out_drop, att_drop, mask rate)
    "ml1m": (3709, 64, 2, 2, 256, 200, 40, 256, 0.2, 0.2, 0.2),
    "ml20m": (26732, 256, 2, 8, 1024, 200, 40, 256, 0.1, 0.1, 0.2),      # layers as in ml-20m_256.json
    "ml20m_4l": (26732, 256, 4, 8, 1024, 200, 40, 256, 0.1, 0.1, 0.2),   # BASELINE.json configs[3]: the 4-layer variant
    "steam": (13047, 64, 2, 2, 256, 50, 20, 256, 0.1, 0.1, 0.4),
    # ml-1m_128.json: the encoder the reference's own ML-1M example trains (examples/bert4rec_ml_1m_example.py:21-25)
    "ml1m_128": (3709, 128, 2, 4, 512, 200, 40, 256, 0.5, 0.2, 0.2),
}
# BASELINE.json configs[3] / [4] (+ the reference's ML-1M example encoder): short driver-visible legs behind the headline number
OTHER_LEGS = (("steam", 150, 30), ("ml20m_4l", 24, 6), ("ml1m_128", 60, 15))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); 6.29 TB/s is the measured copy rate
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md); the split-precision kernels run on it
DTYPE = ("f32 (split products on the 16-bit matrix cores, fp32 accumulate: bf16 hi/lo x3 in the encoder, fp16 hi/lo x3 logits + x2 value "
         "products in the masked-LM head; softmax / LayerNorm / AdamW in fp32)")


# launch label of the library's timer -> kernel name in the rocprofv3 summaries
KERNEL_OF_LABEL = {"b4r_attn_block_fwd": "attn32_fwd_kernel", "b4r_attn_block_bwd": "attn32_bwd_kernel",
                   "b4r_ffn_block_fwd": "ffn_fwd_kernel", "b4r_ffn_block_bwd (dx)": "ffn_bwd_dx_kernel",
                   "b4r_ffn_block_bwd (dw)": "ffn_bwd_dw_kernel", "masked-LM head forward (fused)": ("head", "_fwd_kernel<"),
                   "masked-LM head dE (fused)": ("head", "_dE_kernel<")}   # (head32_fwd_kernel<2> / head32w_fwd_kernel<8, 4> / head_fwd_kernel<..>)
# matrix instructions executed per fp32-equivalent product of the masked-LM head: the 32 x 32-tile kernels (B4R_HEAD32 != 0) form the
# logits as a three-term fp16 hi / lo product and take the probabilities / softmax gradients as ONE fp16 operand (two terms):
# (3 + 2) / 2 per product on average; the 16-row-tile kernels run three bf16 terms everywhere
HEAD32 = os.environ.get("B4R_HEAD32", "1") != "0"
def executed_factor(label):
    return 2.5 if (HEAD32 and label.startswith("masked-LM head")) else 3.0


def library_hash():
    """sha256 of the libb4r_hip.so this process runs (B4R_LIB_PATH or the in-tree build): ties a number to a build"""
    import hashlib
    path = os.environ.get("B4R_LIB_PATH") or os.path.join(ROOT, "bert4rec_amd", "libb4r_hip.so")
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def profile_build(path):
    """the library hash a committed profile summary was taken with (tools/prof.sh writes <tag>_build_<config>.json next to it)"""
    import re
    m = re.match(r"(.*/r[0-9a-z_]*?)_(pmc|stepbytes)_(.*)\.(txt|json)$", path)
    if not m:
        return None
    try:
        return json.load(open(f"{m.group(1)}_build_{m.group(3)}.json")).get("lib_sha256")
    except (OSError, ValueError):
        return None


def stamp(entry, path):
    """source + the build it was measured on; stale = not the library loaded now (the figure then describes another build)"""
    taken = profile_build(os.path.join(ROOT, path))
    entry["source"] = path
    entry["source_lib_sha256"] = taken
    entry["stale"] = (taken is None) or (taken != library_hash())
    return entry


def profiled_traffic(kernel, config):
    """HBM bytes per launch of a kernel from the newest committed PMC summary of this config (profiles/r*_pmc_<config>.txt:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench, gfx950 x2 fetch correction applied by tools/pmc.py);
    the template instances of one kernel (e.g. attn_block_bwd_kernel<true> / <false>) are averaged by launch count.
    bench.py cannot collect PMC counters itself; None when no summary names the kernel."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{config}.txt")), reverse=True):
        try:
            lines = open(path).read().splitlines()
            hdr = lines[0].split()
            i_n, i_rd, i_wr = hdr.index("n"), hdr.index("rdMB"), hdr.index("wrMB")
            tot, n = 0.0, 0.0
            for ln in lines[1:]:
                if all(k_ in ln for k_ in ((kernel,) if isinstance(kernel, str) else kernel)):
                    cols = ln.split()
                    off = len(cols) - len(hdr)          # the kernel name may contain blanks
                    k = float(cols[i_n + off])
                    tot += k * (float(cols[i_rd + off]) + float(cols[i_wr + off]))
                    n += k
            if n > 0:
                return stamp({"bytes": round(tot / n * 1e6)}, os.path.relpath(path, ROOT))
        except (OSError, ValueError, IndexError):
            continue
    return None


def profiled_step_bytes(config):
    """HBM bytes of one whole train step: sum over the kernels of the newest committed PMC summary of (read + written) MB x
    launches per step (profiles/r*_stepbytes_<config>.json, written by tools/prof.sh); None if absent."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_stepbytes_{config}.json")), reverse=True):
        try:
            d = json.load(open(path))
            return stamp({"bytes": int(d["hbm_bytes_per_step"])}, os.path.relpath(path, ROOT))
        except (OSError, ValueError, KeyError):
            continue
    return None


def synthetic_batch(B, L, P, V, rate, seed, ragged=False):
    """S-full rows of SURVEY.md §8(d): every row has length L, n = min(P, max(1, int(L*rate))) masked positions.
    ragged (S-ragged, the realism check): row lengths ~ U{5..L}, right-padded with 0, same masking rule per row."""
    rng = np.random.default_rng(seed)
    ids = rng.integers(3, V, size=(B, L)).astype(np.int64)
    mask = np.ones((B, L), np.int64)
    pos = np.zeros((B, P), np.int64)
    mids = np.zeros((B, P), np.int64)
    if ragged:
        for b in range(B):
            nb = int(rng.integers(5, L + 1))
            ids[b, nb:] = 0
            mask[b, nb:] = 0
    inp = ids.copy()
    for b in range(B):
        nb = int(mask[b].sum())
        n = min(P, max(1, int(nb * rate)))
        p = np.sort(rng.choice(nb, size=n, replace=False))
        pos[b, :n] = p
        mids[b, :n] = ids[b, p]
        inp[b, p] = 1  # [MASK]
    w = (mids != 0).astype(np.int64)
    return {"input_word_ids": torch.from_numpy(inp), "input_mask": torch.from_numpy(mask),
            "labels": torch.from_numpy(ids), "masked_lm_positions": torch.from_numpy(pos),
            "masked_lm_ids": torch.from_numpy(mids), "masked_lm_weights": torch.from_numpy(w)}


def algorithmic_work(label, V, H, NL, NH, I, L, P, B):
    """(bound, unit work per launch, what) of a launch label of the library: ALGORITHMIC bytes for the HBM-bound kernels (fp32
    tensors each touched once), fp32-equivalent FLOPs for the matrix-pipe-bound ones (DESIGN.md §4 derives every entry).
    None for kernels this table does not model (small reductions, scatter, optimizer)."""
    N, M = B * L, B * P
    act = N * H * 4                                         # one [N, H] fp32 activation
    if label.startswith("b4r_ffn_block_fwd"):
        return "hbm", 3 * act + 4 * N * 4 + 2 * H * I * 4, "z1 (-> x1) in; z2, x2, statistics out; W1, W2"
    if label.startswith("b4r_ffn_block_bwd (dx)"):
        return "hbm", 3 * act + 2 * N * 4 + 2 * H * I * 4, "z1 (-> x1), dz2 in; dz1 out; statistics; W1, W2"
    if label.startswith("b4r_ffn_block_bwd (dw)"):
        chunks = -(-N // 32)
        slabs = -(-chunks // -(-chunks // 256))             # b4r_ffn_block_bwd: fewest workgroups with the same longest chunk run
        return "hbm", 2 * act + slabs * (2 * H * I + I + H) * 4, f"z1 (-> x1), dz2 in; {slabs} partial slabs of dW1, dW2, db1, db2 out"
    # the attention blocks: 9.5 GFLOP (fp32-equivalent, 3x that executed as bf16 MFMA) over ~110 MB = below the machine balance of
    # 312 FLOP/byte even counting the split products -> the HBM roof is the binding one
    small = B * NH * L * 4 + 2 * N * 4 + (B * NH * ((L + 15) // 16) * 2 * 64) * 4    # lse, mean / rstd, keep bits
    # the LAST layer's launches of a train step touch the masked-LM slots' rows only where the library sweeps the slots as the only
    # queries (hidden 64, 64 < L <= 224, P <= 64: b4r_attn32.hip's CQ instances): the label's figure is the MEAN over the NL launches
    slots_only = H == 64 and 64 < L <= 224 and P <= 64 and NL >= 1
    slot_rows = M * H * 4
    if label.startswith("b4r_attn_block_fwd"):
        dense = 3 * act + small
        if not slots_only:
            return "hbm", dense, "x in; ctx, z1 out (x1 is formed on load by the feed-forward kernels); lse, statistics, dropout bits"
        last = act + 2 * slot_rows + small * P // L
        return "hbm", ((NL - 1) * dense + last) // NL, ("mean of the layers' launches: x in; ctx, z1 out; lse, statistics, dropout bits -- the last "
                                                       "layer's launch writes the masked-LM slots' rows only")
    if label.startswith("b4r_attn_block_bwd"):
        nt = (L + 31) // 32
        small_b = B * NH * L * 4 + 2 * N * 4 + B * NH * nt * nt * 32 * 4                  # lse, mean / rstd, keep words
        slabs = B * (H * 3 * H + 3 * H + H * H + H) * 4                                   # dWqkv, dbqkv, dWo, dbo partials per sequence
        dense = 4 * act + act + slabs + small_b
        what = ("x, dz1, ctx, previous z in; dx_prev out; per-sequence partials of dWqkv / dbqkv / dWo / dbo out (no [N,3H] round trip); "
                "lse, statistics, dropout words")
        if not slots_only:
            return "hbm", dense, what
        last = 2 * act + 2 * slot_rows + act + slabs + (B * NH * L * 4 + B * NH * nt * nt * 32 * 4) * P // L + 2 * N * 4
        return "hbm", ((NL - 1) * dense + last) // NL, "mean of the layers' launches: " + what + " -- the last layer's launch reads dz1 / ctx at the masked-LM slots' rows only"
    if label.startswith("b4r_attn_fwd"):
        return "mfma", N * 4 * L * H, "QK^T + PV"
    if label.startswith("b4r_attn_bwd dq"):
        return "mfma", N * 6 * L * H, "S, dA recomputed + dQ"
    if label.startswith("b4r_attn_bwd dkv"):
        return "mfma", N * 8 * L * H, "S, dA recomputed + dK + dV"
    if label.startswith("masked-LM head forward (fused)"):
        return "mfma", 2 * (2 * M * V * H), "logit tiles + sum_v p E[v]"
    if label.startswith("masked-LM head dE (fused)"):
        return "mfma", 2 * (2 * M * V * H), "logit tiles recomputed + g^T T"
    return None


def measure_step_breakdown(eng, lib, hp, prepared, nb, n_steps=10):
    """Per-launch times of the train step from the library's own event timer (enqueue order, averaged over n_steps steps)."""
    from bert4rec_amd import _lib
    stream = torch.cuda.current_stream().cuda_stream
    cap = 256 * n_steps
    torch.cuda.synchronize()
    _lib.check(lib.b4r_timing_begin(stream, cap), "b4r_timing_begin")
    for i in range(n_steps):
        eng.train_step(hp, prepared[i % nb][0])
    n = C.c_int32(0)
    us = (C.c_float * cap)()
    stride = 128
    names = C.create_string_buffer(cap * stride)
    _lib.check(lib.b4r_timing_end(C.byref(n), us, names, stride, cap), "b4r_timing_end")
    per_step = n.value // n_steps
    assert per_step * n_steps == n.value, "launch count varies between steps"
    rows = []
    for j in range(per_step):
        label = names.raw[j * stride:(j + 1) * stride].split(b"\0", 1)[0].decode()
        t = float(np.mean([us[s * per_step + j] for s in range(n_steps)]))
        rows.append((label, t))
    return rows


def eval_leg(device, V_items=3706, users=6040, seed=0):
    """Train + evaluate through the product surface on a synthetic Zipf log with successor structure (datasets.make_synthetic,
    order=0.6): dataloader factory -> prepare_training(device_masking) -> trainer -> evaluator (100 popularity negatives per
    user, bert4rec_evaluator.py:60-120).  Returns NDCG@10 / HR@10 and the evaluation throughput."""
    from bert4rec_amd import config, dataloaders, datasets, evaluation, models, trainers
    from bert4rec_amd.dataloaders import dataloader_utils
    from bert4rec_amd.models.components import networks
    from bert4rec_amd.trainers import optimizers
    t0 = time.perf_counter()
    ds = datasets.synthetic_dataset(n_users=users, n_items=V_items, min_len=20, max_len=200, seed=seed, order=0.6)
    dl = dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(data_source=ds, input_duplication_factor=1)
    train, val, test = dl.prepare_training(device_masking=True)
    enc = networks.Bert4RecEncoder(dl.get_tokenizer().get_vocab_size(), seed=5, device=device, **config.get_encoder_config("ml-1m_64"))
    model = models.BERT4RecModel(enc)
    trainer = trainers.get(model=model)
    epochs, lr = 60, 1e-3
    steps = epochs * ((len(train) + 255) // 256)
    trainer.initialize_model(optimizer=optimizers.get("adamw", init_lr=lr, num_train_steps=steps, num_warmup_steps=50))
    tb = dataloader_utils.make_batches(train, batch_size=256, seed=1, remask_each_epoch=True)
    t_prep = time.perf_counter() - t0
    t0 = time.perf_counter()
    hist = model.fit(tb, epochs=epochs, verbose=0)
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    ev = evaluation.get(dataloader=dl, seed=3)
    test_b = dataloader_utils.make_batches(test, batch_size=256, seed=2).cache_on_device(device)
    ev.evaluate(model, test_b)                      # warm-up pass (also builds the sampler tables)
    ev.reset_metrics()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev.evaluate(model, test_b)
    res = ev.get_metrics_results()                  # one device -> host copy
    t_eval = time.perf_counter() - t0
    n_users = int(res["Valid Ranks"])
    return {"dataset": f"synthetic Zipf(1.2) log, {users} users x {V_items} items, lengths U{{20..200}}, successor structure 0.6 "
                       f"(NOT MovieLens: no dataset files on the box)",
            "train": f"{epochs} epochs = {steps} steps of B=256, AdamW lr {lr} (warm-up 50, linear decay), masks redrawn on the GPU "
                     f"every epoch, final loss {hist.history['loss'][-1]:.3f}",
            "NDCG@10": round(float(res["NDCG@10"]), 4), "HR@10": round(float(res["HR@10"]), 4),
            "NDCG@5": round(float(res["NDCG@5"]), 4), "MAP": round(float(res["MAP"]), 4), "users": n_users,
            "eval_users_per_s": round(n_users / t_eval, 1), "eval_ms": round(t_eval * 1e3, 2),
            "train_s": round(t_train, 2), "host_prepare_s": round(t_prep, 2),
            "chance_level_HR@10": round(10 / 101, 4)}


def launch_command(n_gpus, argv, port):
    """the command the driver itself uses for N > 1 (one rank per GPU over RCCL, rendezvous on 127.0.0.1)"""
    passed = [a for a in argv if a != "--dry-launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + passed


def launch_ranks(n_gpus, argv, dry=False):
    """Start the N ranks of `python bench.py --gpus N` as child processes and return their exit status."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = launch_command(n_gpus, argv, port)
    if dry:
        print(json.dumps({"launch": cmd, "ranks": n_gpus}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def other_config_leg(name, steps, warmup, device):
    """One more configuration of BASELINE.json timed the same way (full train steps on S-full synthetic batches resident in HBM,
    eager launches, device synchronised on both sides), short enough to ride behind the headline run: a few regions, the median."""
    from bert4rec_amd.engine import Engine, make_adamw_config, make_model_config
    V, H, NL, NH, I, L, P, B, od, ad, rate = CONFIGS[name]
    eng = Engine(make_model_config(V, H, NL, NH, L, I, od, ad), device, seed=4321)
    eng.init_parameters(seed=3)
    hp = make_adamw_config()
    batches = [synthetic_batch(B, L, P, V, rate, seed=7000 + i) for i in range(2)]
    prepared = [eng.prepare_batch(b) for b in batches]
    valid = float(sum(int((b["masked_lm_ids"] != 0).sum()) for b in batches)) / len(batches)
    for i in range(warmup):
        eng.train_step(hp, prepared[i % 2][0])
    torch.cuda.synchronize()
    regions = []
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(steps):
            eng.train_step(hp, prepared[i % 2][0])
        torch.cuda.synchronize()
        regions.append(time.perf_counter() - t0)
    regions.sort()
    el = regions[1]
    st = eng.read_state()
    loss = st["loss_sum"] / max(st["valid_count"], 1.0)
    out = {"ms_per_step": round(el / steps * 1e3, 4), "value": round(valid * steps / el, 1), "unit": "masked positions/s",
           "steps": steps, "warmup": warmup, "regions": 3, "final_loss": round(loss, 4), "finite": bool(np.isfinite(loss)),
           "workload": f"{name}: full train step, B={B} L={L} P={P} H={H} layers={NL} heads={NH} inner={I} V={V} dropout {od}/{ad}"}
    del eng, prepared
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", default="ml1m", choices=list(CONFIGS))
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed oracle steps for cpu_baseline (0 disables)")
    ap.add_argument("--no-eval", action="store_true", help="skip the train + evaluate leg (NDCG@10 on the synthetic log)")
    ap.add_argument("--no-other", action="store_true", help="skip the short legs on the other BASELINE configurations (other_configs)")
    ap.add_argument("--no-breakdown", action="store_true", help="skip the event-timed extra steps (profiling runs: tools/prof.sh)")
    ap.add_argument("--phases", action="store_true", help="also print per-phase timings to stderr")
    ap.add_argument("--ragged", action="store_true", help="diagnostic only: S-ragged rows (lengths U{5..L}); reports the padding "
                                                          "penalty, NOT the headline configuration")
    ap.add_argument("--bucketed", action="store_true", help="with --ragged: what make_batches(bucket_by_length=4, trim_padding=True) does to "
                                                            "the same rows -- the 4 batches are re-formed by length and cut to their longest "
                                                            "sequence (timing line only)")
    ap.add_argument("--no-dropout", action="store_true", help="diagnostic only: dropout 0 (NOT the headline configuration)")
    ap.add_argument("--graph", action="store_true", help="replay each batch's step from captured hipGraphs (same GPU time, "
                                                         "~7x less host time per step; N > 1: two graphs around the all-reduce)")
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL code path (init, broadcast, all-reduce, barriers) with "
                                                              "ONE rank: the only way to rehearse it on a one-GPU box")
    ap.add_argument("--dry-launch", action="store_true", help="with --gpus N > 1 and no WORLD_SIZE in the environment: print the "
                                                              "torch.distributed.run command the launcher would start (JSON) and exit")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the LAUNCHER.  It has made no HIP / torch.cuda call (importing torch
        # initialises nothing), starts N fresh rank processes as children -- never an exec of a process that touched the GPU -- lets
        # their stdout through (rank 0 prints the one JSON line) and exits with their status.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], args.dry_launch))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs the GPU (the product has no CPU path)"
    # one rank per GPU; when the launcher isolates each rank's card (HIP_VISIBLE_DEVICES per rank) the only visible index is 0
    dev_index = local_rank if local_rank < torch.cuda.device_count() else 0
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.force_dist
    # RCCL prints a version banner on stdout when its first communicator is created: until the warm-up is over, stdout (fd 1) is
    # pointed at stderr so that the one JSON line stays the only thing on stdout
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    from bert4rec_amd import _lib
    from bert4rec_amd.distributed import broadcast_parameters
    from bert4rec_amd.engine import Engine, make_adamw_config, make_model_config

    V, H, NL, NH, I, L, P, B, od, ad, rate = CONFIGS[args.config]
    if args.no_dropout:
        od = ad = 0.0
    cfg = make_model_config(V, H, NL, NH, L, I, od, ad)
    eng = Engine(cfg, device, seed=1234 + rank)   # per-rank dropout stream: ranks hold different rows
    eng.init_parameters(seed=3)
    eng.rehearse_collectives = args.force_dist
    broadcast_parameters(eng.params)
    hp = make_adamw_config()
    nb = 4
    batches = [synthetic_batch(B, L, P, V, rate, seed=1000 * rank + i, ragged=args.ragged) for i in range(nb)]
    if args.bucketed:   # same rows, batches re-formed by length and trimmed (dataloader_utils.bucket_order / trimmed_length)
        from bert4rec_amd.dataloaders import dataloader_utils as du
        pool = {k: torch.cat([b[k] for b in batches]) for k in batches[0]}
        lens = pool["input_mask"].sum(1).numpy()
        order = du.bucket_order(np.arange(nb * B), lens, B, nb, seed=0)
        batches = []
        for s_ in range(0, nb * B, B):
            idx = torch.from_numpy(order[s_:s_ + B])
            cols = du.trimmed_length(lens[order[s_:s_ + B]].max(), L)
            slots = du.trimmed_length(int((pool["masked_lm_weights"][idx] != 0).sum(1).max()), P, 4)
            batches.append({k: v[idx][:, :(cols if k in du.PER_TOKEN_KEYS else slots)].contiguous() for k, v in pool.items()})
        args.no_breakdown = True
    prepared = [eng.prepare_batch(b) for b in batches]
    valid_per_step = float(sum(int((b["masked_lm_ids"] != 0).sum()) for b in batches)) / nb
    graphs = args.graph or (use_dist and world > 1)   # N > 1: the host must not become the bottleneck of the step

    def step(i):
        cb, _ = prepared[i % nb]
        if use_dist:
            (eng.dp_train_step_graphed if graphs else eng.dp_train_step)(hp, cb)
        elif args.graph:
            eng.train_step_graphed(hp, cb)
        else:
            eng.train_step(hp, cb)

    for i in range(max(args.warmup, 3 * nb if graphs else 0)):   # graphs: eager, capture, replay once per batch before timing
        step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    def timed_region(offset):
        """EXACTLY args.steps steps between barriers + device synchronisation; the max over ranks"""
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(offset + i)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # a region shorter than 200 ms (the driver's 20 steps are 12 ms) is repeated and the MEDIAN region reported: one region of that
    # length moves by several per cent with the clock state it starts in.  Every rank takes the same decision (the first region's
    # time is already the max over ranks).
    regions = [timed_region(args.warmup)]
    repeats = 1 if (regions[0] >= 0.2 or args.no_breakdown) else max(7, min(51, int(0.2 / max(regions[0], 1e-6)) | 1))   # (profiling runs: one region)
    for k in range(1, repeats):
        if use_dist:
            dist.barrier()
        regions.append(timed_region(args.warmup + k * args.steps))
    regions.sort()
    elapsed = regions[len(regions) // 2]
    st = eng.read_state()
    loss = st["loss_sum"] / max(st["valid_count"], 1.0)
    assert np.isfinite(loss), "training diverged"

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = valid_per_step * world * args.steps / elapsed
        lib = _lib.load()
        cb, _ = prepared[0]
        M = B * P
        stream = torch.cuda.current_stream().cuda_stream

        # ---- roofline: the dominant kernel of the step, found by timing every launch of the step ------------------------------
        roofline, roofline_head, breakdown, step_hbm = None, None, None, None
        if not args.no_breakdown:
            rows = measure_step_breakdown(eng, lib, hp, prepared, nb)
            per_label = collections.OrderedDict()
            for label, t in rows:
                a = per_label.setdefault(label, [0.0, 0])
                a[0] += t
                a[1] += 1
            total_us = sum(t for _, t in rows)
            breakdown = {"launches_per_step": len(rows), "event_timed_us_per_step": round(total_us, 1),
                         "kernels": [{"launch": k, "n_per_step": v[1], "us_per_step": round(v[0], 1)} for k, v in
                                     sorted(per_label.items(), key=lambda kv: -kv[1][0])]}
            def roof_of(label, tot, cnt, selection):
                work = algorithmic_work(label, V, H, NL, NH, I, L, P, B)
                if work is None:
                    return None
                bound, amount, what = work
                avg_us = tot / cnt
                if bound == "hbm":
                    ach, peak, unit = amount / (avg_us * 1e-6) / 1e9, HBM_PEAK_GBS, "GB/s"
                else:
                    ach, peak, unit = amount / (avg_us * 1e-6) / 1e12, BF16_PEAK_TFLOPS, "TFLOP/s"
                kname = next((v for k, v in KERNEL_OF_LABEL.items() if label.startswith(k)), None)
                tr = profiled_traffic(kname, args.config) if kname else None
                r = {"kernel": label, "selection": selection,
                     "bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
                     "traffic": tr["bytes"] if tr else None, "traffic_source": tr["source"] if tr else None,
                     "traffic_stale": tr["stale"] if tr else None, "traffic_lib_sha256": tr["source_lib_sha256"] if tr else None,
                     "algorithmic_" + ("bytes" if bound == "hbm" else "flops"): int(amount), "what_is_counted": what,
                     "avg_launch_us": round(avg_us, 2), "launches_per_step": cnt,
                     "timer": "hipEvents on the launch stream behind every launch of 10 extra steps (b4r_timing_begin/_end)"}
                if bound == "mfma":
                    xf = executed_factor(label)
                    r["executed_mfma_flops"] = int(xf * amount)
                    r["frac_executed"] = round(xf * ach / peak, 4)
                    r["executed_per_product"] = xf
                return r

            for label, (tot, cnt) in sorted(per_label.items(), key=lambda kv: -kv[1][0]):
                roofline = roof_of(label, tot, cnt, f"largest time per step of the {len(per_label)} distinct launches "
                                                    f"({tot:.1f} us = {100 * tot / total_us:.1f} % of the step's kernel time)")
                if roofline is not None:
                    break
            for label, (tot, cnt) in per_label.items():   # second entry: the vocabulary sweep of the train step's masked-LM head
                if label.startswith("masked-LM head forward (fused)") and (roofline is None or roofline["kernel"] != label):
                    roofline_head = roof_of(label, tot, cnt, "the masked-LM head's vocabulary sweep (north star: the head's roofline)")
            sb = profiled_step_bytes(args.config)
            if sb:
                gbs = sb["bytes"] / (ms * 1e-3) / 1e9
                step_hbm = {"hbm_bytes_per_step": sb["bytes"], "source": sb["source"], "source_lib_sha256": sb["source_lib_sha256"],
                            "stale": sb["stale"], "achieved_GBs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}

        # ---- the materialising masked-LM-head projection (forward / evaluation API), replayed on live buffers ----------------
        if args.no_breakdown:      # profiling runs (tools/prof.sh): the train step only, no replays in the kernel statistics
            note = "--no-breakdown: timing only"
            if args.bucketed:
                note += "; batches re-formed by length and trimmed to " + "/".join(
                    f"{b['input_word_ids'].shape[1]}x{b['masked_lm_ids'].shape[1]}" for b in batches) + " token x slot columns"
            print(json.dumps({"metric": "masked positions/sec", "value": round(value, 1), "ms_per_step": round(ms, 4),
                              "note": note}), flush=True)
            if use_dist:
                dist.barrier()
                dist.destroy_process_group()
            return
        eng.forward(cb, training=False, pooler=False)        # fills mlm_hidden / mlm_logits of this batch shape
        t_h = eng.region("mlm_hidden", B, L, P)
        logits = eng.region("mlm_logits", B, L, P)
        d = _lib.GemmDesc()
        d.A, d.lda = t_h.data_ptr(), t_h.stride(0)
        d.B, d.ldb = eng.view("word_embeddings/embeddings").data_ptr(), H
        d.C, d.ldc = logits.data_ptr(), logits.stride(0)
        d.M, d.N, d.K, d.b_is_nk, d.epilogue = M, V, H, 1, _lib.EPI_BIAS
        d.bias = eng.view("cls/predictions/output_bias/bias").data_ptr()
        d.c_pad_scratch = 1  # exactly as b4r_forward launches it: the pad columns V..Vp-1 of the logits rows are scratch
        for _ in range(5):
            _lib.check(lib.b4r_gemm_f32(C.byref(d), stream))
        reps = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            _lib.check(lib.b4r_gemm_f32(C.byref(d), stream))
        e1.record()
        torch.cuda.synchronize()
        k_us = e0.elapsed_time(e1) * 1e3 / reps
        alg_bytes = M * V * 4 + M * H * 4 + V * H * 4 + V * 4 + M * 8
        achieved = alg_bytes / (k_us * 1e-6) / 1e9
        roofline_mat = {"kernel": "mlm_logits = T.E^T + b (b4r_gemm_f32, B4R_EPI_BIAS, bf16x3)", "bound": "hbm",
                        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes": alg_bytes,
                        "avg_launch_us": round(k_us, 2), "in_timed_region": not eng.fused_head_supported()}
        # (the train step never launches this kernel: its counters come from the API-path profile of the same build, tools/prof_api.sh)
        tr = profiled_traffic("rx_gemm_nk_kernel<1, false, 4>", args.config + "_api") or profiled_traffic("rx_gemm_nk_kernel<1, false, 4>", args.config)
        if tr:
            roofline_mat["traffic"], roofline_mat["traffic_source"] = tr["bytes"], tr["source"]
            roofline_mat["traffic_stale"], roofline_mat["traffic_lib_sha256"] = tr["stale"], tr["source_lib_sha256"]
        if roofline is None:
            roofline = roofline_mat

        if args.phases:
            def timed(fn, n=20):
                fn()
                torch.cuda.synchronize()
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(n):
                    fn()
                b_.record()
                torch.cuda.synchronize()
                return a.elapsed_time(b_) / n
            fused = eng.fused_head_supported()
            eng.begin_step()
            f_ms = timed(lambda: eng.forward(cb, training=True, pooler=False, fused_head=fused))
            l_ms = timed(lambda: (eng.forward(cb, training=True, pooler=False, fused_head=fused), eng.loss(cb, True, fused_head=fused)))
            b_ms = timed(lambda: eng.backward(cb, training=True, fused_head=fused))
            o_ms = timed(lambda: eng.optimizer_step(hp, cb))
            print(f"[phases] forward {f_ms:.3f} ms, loss {l_ms - f_ms:.3f} ms, backward {b_ms:.3f} ms, optimizer {o_ms:.3f} ms",
                  file=sys.stderr)

        # ---- train + evaluate leg: NDCG@10 (synthetic log) and evaluation throughput ----------------------------------------
        ev = None
        if world == 1 and not args.no_eval and args.config == "ml1m":
            ev = eval_leg(device)

        # ---- the other BASELINE configurations, a few dozen steps each (N = 1, headline config only) -------------------------------
        others = None
        if world == 1 and not args.no_other and args.config == "ml1m" and not args.ragged:
            others = {}
            for name, k_, w_ in OTHER_LEGS:
                try:
                    others[name] = other_config_leg(name, k_, w_, device)
                except Exception as e:   # a leg that fails must not take the headline line with it; it is reported as failed
                    others[name] = {"error": f"{type(e).__name__}: {e}"[:300]}

        # ---- CPU baseline: the oracle's train step on the host cores (rank 0, N=1 only) ---------------------------
        cpu = None
        topk = None
        if world == 1 and args.cpu_steps > 0 and args.config in ("ml1m", "steam"):
            from oracle import bert4rec_oracle as orc
            # the GPU box exposes every host core but a one-GPU job owns a 16-core share: more threads only thrash
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            torch.set_num_threads(int(os.environ.get("B4R_CPU_THREADS", min(avail, 16))))
            cfg_o = orc.OracleConfig(vocab_size=V, hidden_size=H, num_layers=NL, num_attention_heads=NH,
                                     max_sequence_length=L, inner_dim=I, output_dropout=od, attention_dropout=ad)
            params = orc.init_params(cfg_o, seed=3)
            m, v = orc.zeros_like_params(params), orc.zeros_like_params(params)
            hp_o = orc.AdamWConfig()
            orc.train_step(params, m, v, batches[0], cfg_o, hp_o, step=0, training=True, rng=(1234, 0))
            c0 = time.perf_counter()
            for i in range(args.cpu_steps):
                orc.train_step(params, m, v, batches[(i + 1) % nb], cfg_o, hp_o, step=i + 1, training=True, rng=(1234, i + 1))
            c_el = time.perf_counter() - c0
            cpu = {"value": round(valid_per_step * args.cpu_steps / c_el, 1), "unit": "masked positions/s",
                   "cores": torch.get_num_threads(), "kind": "port",
                   "sample": f"{args.cpu_steps} full train steps of the same S-full batches (B={B}, L={L}, P={P}) after 1 warm-up "
                             f"step, torch-CPU fp32 restatement of the reference math (TF2 unavailable), "
                             f"{c_el / args.cpu_steps * 1e3:.0f} ms/step"}

            # ---- the checker once more: top-10 item lists of the GPU forward against the oracle's, same weights, one batch ------
            # ("ranked top-k bit-exact" holds for the ranking step given equal hidden states; across two float implementations of the
            # encoder the lists can differ where two logits are closer than their rounding: this is how often, and by how much)
            w_now = eng.export_named()
            for n_, p_ in params.items():
                if n_ in w_now:
                    params[n_] = w_now[n_].detach().cpu().reshape(p_.shape).clone()
            with torch.no_grad():
                lo = orc.model_forward(params, batches[0], cfg_o)["mlm_logits"].reshape(B * P, -1)[:, :V]
            eng.forward(prepared[0][0], training=False, pooler=False)
            lg = eng.region("mlm_logits", B, L, P)[:, :V].detach().cpu()
            valid = (batches[0]["masked_lm_ids"].reshape(-1) != 0)
            to, tg = torch.topk(lo[valid], 10).indices, torch.topk(lg[valid], 10).indices
            same = (to == tg).all(1)
            gap = 0.0
            if not bool(same.all()):
                rows_ = torch.nonzero(~same).reshape(-1)
                first = (to[rows_] != tg[rows_]).float().argmax(1)
                lo_v = lo[valid]
                a_, b_ = to[rows_, first], tg[rows_, first]
                gap = float((lo_v[rows_, a_] - lo_v[rows_, b_]).abs().max())
            topk = {"slots": int(valid.sum()), "top10_identical_share": round(float(same.float().mean()), 6),
                    "max_logit_gap_where_different": gap, "max_abs_logit_diff": round(float((lo[valid] - lg[valid]).abs().max()), 7),
                    "against": "oracle forward (CPU fp32 restatement; parity with TF2 unpinned, DESIGN.md §1), same weights, batch 0"}

        result = {"metric": "masked positions/sec", "value": round(value, 1), "unit": "masked positions/s",
                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
                  "repeats": len(regions), "ms_min": round(regions[0] / args.steps * 1e3, 4), "ms_max": round(regions[-1] / args.steps * 1e3, 4),
                  "timing": f"median of {len(regions)} timed regions of {args.steps} steps each" if len(regions) > 1 else "one timed region",
                  "lib_sha256": library_hash(),
                  "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
                  "config": {"workload": f"{args.config}: full train step, B={B}/GPU L={L} P={P} H={H} layers={NL} heads={NH} "
                                         f"inner={I} V={V} dropout {od}/{ad}, full-vocab masked-LM head, {int(valid_per_step)} "
                                         f"masked positions/GPU/step" + (" (S-ragged rows: lengths U{5..L})" if args.ragged else ""),
                             "global_batch": B * world, "seq_len": L, "parallelism": f"dp{world}",
                             "launch_mode": "hipGraph replay" if graphs else "eager"},
                  "per_gpu": round(value / world, 1), "final_loss": round(loss, 5),
                  "roofline": roofline, "roofline_head": roofline_head, "step_hbm": step_hbm, "roofline_materialising": roofline_mat, "eval": ev, "topk_agreement": topk,
                  "cpu_baseline": cpu, "other_configs": others, "step_breakdown": breakdown}
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
