#!/usr/bin/env python3
"""Benchmark of the BERT4Rec hot path on MI355X: full train steps (forward + masked CE + backward + clip + AdamW
[+ RCCL all-reduce]) on the ML-1M configuration of BASELINE.json (configs[1]):
B=256 per GPU, L=200, P=40, H=64, 2 layers, 2 heads, inner 256, V=3709, dropout 0.2/0.2, full-vocab masked-LM head.

    python bench.py --gpus 1 --steps 200 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` = masked positions (slots with masked_lm_ids != 0) consumed per second by the
whole job, inputs resident in HBM.  `roofline` is measured on the longest kernel of the step, replayed stand-alone on the
live buffers between two HIP events on the launch stream: with hidden size 64 that is the vocabulary sweep of the fused
masked-LM head (head_fwd_kernel: logits tiles -> online softmax -> sum_v p E[v], nothing of size [M,V] touches HBM, so
the bound is the matrix pipe); `roofline_materialising` reports the HBM-bound logits[M,V] = T.E^T + b kernel that the
forward / evaluation API still uses (and that larger hidden sizes train with).
`cpu_baseline` is the oracle (CPU restatement of the reference math; TF2 is not installed anywhere) timed on the host.
"""
import argparse
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
}

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); 6.29 TB/s is the measured copy rate
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md); the split-precision kernels run on it


def profiled_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel from the newest committed PMC summary (profiles/r01_*_pmc_ml1m*.txt: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench, gfx950 x2 fetch correction applied by tools/pmc.py).
    bench.py cannot collect PMC counters itself; None when no summary names the kernel."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r01_*_pmc_ml1m*.txt")), reverse=True):
        try:
            lines = open(path).read().splitlines()
            hdr = lines[0].split()
            i_rd, i_wr = hdr.index("rdMB"), hdr.index("wrMB")
            for ln in lines[1:]:
                if kernel_prefix in ln:
                    cols = ln.split()
                    off = len(cols) - len(hdr)          # the kernel name may contain blanks
                    return {"bytes": round((float(cols[i_rd + off]) + float(cols[i_wr + off])) * 1e6),
                            "source": os.path.relpath(path, ROOT)}
        except (OSError, ValueError, IndexError):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", default="ml1m", choices=list(CONFIGS))
    ap.add_argument("--cpu-steps", type=int, default=2, help="timed oracle steps for cpu_baseline (0 disables)")
    ap.add_argument("--phases", action="store_true", help="also print per-phase timings to stderr")
    ap.add_argument("--ragged", action="store_true", help="diagnostic only: S-ragged rows (lengths U{5..L}); reports the padding "
                                                          "penalty, NOT the headline configuration")
    ap.add_argument("--no-dropout", action="store_true", help="diagnostic only: dropout 0 (NOT the headline configuration)")
    ap.add_argument("--graph", action="store_true", help="replay each batch's step from captured hipGraphs (same GPU time, "
                                                         "~7x less host time per step; N > 1: two graphs around the all-reduce)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs the GPU (the product has no CPU path)"
    # one rank per GPU; when the launcher isolates each rank's card (HIP_VISIBLE_DEVICES per rank) the only visible index is 0
    dev_index = local_rank if local_rank < torch.cuda.device_count() else 0
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # B4R_BENCH_FORCE_DIST=1: run the RCCL code path (init, broadcast, all-reduce, barriers) with one rank -- the only way to
    # rehearse it on a one-GPU box
    use_dist = world > 1 or os.environ.get("B4R_BENCH_FORCE_DIST") == "1"
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
    eng = Engine(cfg, device, seed=1234)
    eng.init_parameters(seed=3)
    broadcast_parameters(eng.params)
    hp = make_adamw_config()
    nb = 4
    batches = [synthetic_batch(B, L, P, V, rate, seed=1000 * rank + i, ragged=args.ragged) for i in range(nb)]
    prepared = [eng.prepare_batch(b) for b in batches]
    valid_per_step = float(sum(int((b["masked_lm_ids"] != 0).sum()) for b in batches)) / nb

    def step(i):
        cb, _ = prepared[i % nb]
        if use_dist:
            (eng.dp_train_step_graphed if args.graph else eng.dp_train_step)(hp, cb)
        elif args.graph:
            eng.train_step_graphed(hp, cb)
        else:
            eng.train_step(hp, cb)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = eng.read_state()
    loss = st["loss_sum"] / max(st["valid_count"], 1.0)
    assert np.isfinite(loss), "training diverged"

    result = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = valid_per_step * world * args.steps / elapsed
        # ---- roofline: the materialising masked-LM-head projection, replayed on the live buffers ------------------
        cb, _ = prepared[0]
        lib = _lib.load()
        M = B * P
        t_h = eng.region("mlm_hidden", B, L, P)
        logits = eng.region("mlm_logits", B, L, P)
        d = _lib.GemmDesc()
        d.A, d.lda = t_h.data_ptr(), t_h.stride(0)
        d.B, d.ldb = eng.view("word_embeddings/embeddings").data_ptr(), H
        d.C, d.ldc = logits.data_ptr(), logits.stride(0)
        d.M, d.N, d.K, d.b_is_nk, d.epilogue = M, V, H, 1, _lib.EPI_BIAS
        d.bias = eng.view("cls/predictions/output_bias/bias").data_ptr()
        d.c_pad_scratch = 1  # exactly as b4r_forward launches it: the pad columns V..Vp-1 of the logits rows are scratch
        stream = torch.cuda.current_stream().cuda_stream
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
        roofline_mat = {"kernel": "rx_gemm_nk_kernel<BIAS> (mlm_logits = T.E^T + b, bf16x3)", "bound": "hbm",
                        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes": alg_bytes,
                        "avg_launch_us": round(k_us, 2), "in_timed_region": not eng.fused_head_supported()}
        tr = profiled_traffic("rx_gemm_nk_kernel<1, false, 4>") if args.config == "ml1m" else None
        if tr:
            roofline_mat["traffic"], roofline_mat["traffic_source"] = tr["bytes"], tr["source"]
        roofline = roofline_mat
        if eng.fused_head_supported():
            # the train step's head: replay the vocabulary sweep (the longest kernel of the step) on the live buffers
            _, keep0 = prepared[0]
            scratch = torch.empty(lib.b4r_mlm_head_fused_scratch_floats(M, V, H), dtype=torch.float32, device=device)
            hargs = (t_h.data_ptr(), eng.view("word_embeddings/embeddings").data_ptr(),
                    eng.view("cls/predictions/output_bias/bias").data_ptr(), keep0["masked_lm_ids"].data_ptr(), M, V, H,
                    scratch.data_ptr(), None, None, None, None, 1, stream)
            for _ in range(5):
                _lib.check(lib.b4r_mlm_head_fused_fwd(*hargs))
            e0.record()
            for _ in range(reps):
                _lib.check(lib.b4r_mlm_head_fused_fwd(*hargs))
            e1.record()
            torch.cuda.synchronize()
            h_us = e0.elapsed_time(e1) * 1e3 / reps
            alg_flops = 2 * (2 * M * V * H)          # x = T.E^T and sum_v p[m,v] E[v,:], fp32-equivalent multiply-adds
            tf = alg_flops / (h_us * 1e-6) / 1e12
            roofline = {"kernel": "head_fwd_kernel (masked-LM head: logit tiles -> online softmax -> p.E, bf16x3)",
                        "bound": "mfma", "achieved": round(tf, 1), "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tf / BF16_PEAK_TFLOPS, 4), "traffic": None, "algorithmic_flops": alg_flops,
                        "executed_mfma_flops": 3 * alg_flops, "frac_executed": round(3 * tf / BF16_PEAK_TFLOPS, 4),
                        "avg_launch_us": round(h_us, 2)}
            tr = profiled_traffic("head_fwd_kernel") if args.config == "ml1m" else None
            if tr:
                roofline["traffic"], roofline["traffic_source"] = tr["bytes"], tr["source"]

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
            eng.begin_step()
            f_ms = timed(lambda: eng.forward(cb, training=True, pooler=False))
            l_ms = timed(lambda: (eng.forward(cb, training=True, pooler=False), eng.loss(cb, True)))
            b_ms = timed(lambda: eng.backward(cb, training=True))
            o_ms = timed(lambda: eng.optimizer_step(hp, cb))
            print(f"[phases] forward {f_ms:.3f} ms, loss {l_ms - f_ms:.3f} ms, backward {b_ms:.3f} ms, optimizer {o_ms:.3f} ms",
                  file=sys.stderr)

        # ---- CPU baseline: the oracle's train step on the host cores (rank 0, N=1 only) ---------------------------
        cpu = None
        if world == 1 and args.cpu_steps > 0 and args.config == "ml1m":
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

        result = {"metric": "masked positions/sec", "value": round(value, 1), "unit": "masked positions/s",
                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
                  "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                  "config": {"workload": f"{args.config}: full train step, B={B}/GPU L={L} P={P} H={H} layers={NL} heads={NH} "
                                         f"inner={I} V={V} dropout {od}/{ad}, full-vocab masked-LM head, {int(valid_per_step)} "
                                         f"masked positions/GPU/step" + (" (S-ragged rows: lengths U{5..L})" if args.ragged else ""),
                             "global_batch": B * world, "seq_len": L, "parallelism": f"dp{world}"},
                  "per_gpu": round(value / world, 1), "final_loss": round(loss, 5),
                  "roofline": roofline, "roofline_materialising": roofline_mat, "cpu_baseline": cpu}
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
