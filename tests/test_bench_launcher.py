"""`python bench.py --gpus N` without a launcher around it (SURVEY.md §8e readiness): the process must start its own N ranks
through torch.distributed.run BEFORE it touches the GPU, relay rank 0's JSON line and exit with the children's status."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_dry_launch_prints_the_torchrun_command_for_n_ranks():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-launch"],
                         env=_env(), capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    cmd = d["launch"]
    assert d["ranks"] == 8 and cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]          # the ranks get the caller's arguments, not the flag


def test_a_rank_under_a_launcher_does_not_launch_again():
    """with WORLD_SIZE in the environment the process IS a rank: it must go on to the GPU work (which fails here: no GPU) and not
    start children of its own -- so no launch record is printed and the exit status is the rank's failure"""
    env = dict(_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-launch"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert '"launch"' not in out.stdout
    import torch
    if not torch.cuda.is_available():
        assert out.returncode != 0 and "needs the GPU" in out.stderr


def test_attention_roofline_bytes_are_the_mean_over_the_layers_launches():
    """bench.py prices the attention block launches of a step with ONE figure per label: where the library sweeps the last layer's
    slots only (hidden 64, 64 < L <= 224, P <= 64) that figure is the mean of a dense launch and the smaller last-layer one."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dense_like = bench.algorithmic_work("b4r_attn_block_bwd", 13047, 64, 2, 2, 256, 50, 20, 256)[1]      # Steam: L = 50, every launch dense
    assert dense_like == 5 * 256 * 50 * 64 * 4 + 256 * (64 * 192 + 192 + 64 * 64 + 64) * 4 + 256 * 2 * 50 * 4 + 2 * 256 * 50 * 4 + 256 * 2 * 2 * 2 * 32 * 4
    two = bench.algorithmic_work("b4r_attn_block_bwd", 3709, 64, 2, 2, 256, 200, 40, 256)[1]
    one = bench.algorithmic_work("b4r_attn_block_bwd", 3709, 64, 1, 2, 256, 200, 40, 256)[1]            # a single layer: the small figure
    four = bench.algorithmic_work("b4r_attn_block_bwd", 3709, 64, 4, 2, 256, 200, 40, 256)[1]
    dense = 2 * two - one
    assert one < two < four < dense and abs(four - (3 * dense + one) // 4) <= 1
    fwd_two = bench.algorithmic_work("b4r_attn_block_fwd", 3709, 64, 2, 2, 256, 200, 40, 256)[1]
    fwd_wide = bench.algorithmic_work("b4r_attn_block_fwd", 3709, 64, 2, 2, 256, 200, 80, 256)[1]        # more than 64 slots: dense
    assert fwd_two < fwd_wide
