"""Data-parallel train steps ON the GPU: two ranks share the one card of the test box, each runs
Engine.dp_train_step (HIP forward / loss / backward of the loss SUM on its half of the batch, one flat all-reduce of
[grads | loss sums], optimizer on the reduced buffer).  RCCL refuses two ranks on one device, so the exchange goes through
gloo here; everything else is the path bench.py --gpus N runs.  The result must equal single-process train steps on the
whole batch (trainer_utils.py:19-22 normalises by the batch-global count) -- dropout off, because dropout indices are local
to a rank's batch."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

V, H, NL, NH, L, I, B, P, STEPS, EPOCHS = 203, 64, 2, 2, 32, 128, 16, 6, 3, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _engine():
    from bert4rec_amd.engine import Engine, make_model_config
    eng = Engine(make_model_config(V, H, NL, NH, L, I, 0.0, 0.0), "cuda", seed=5)
    eng.init_parameters(seed=9)
    return eng


def _batches():
    from oracle import bert4rec_oracle as orc
    return [orc.synthetic_batch(B, L, P, V, seed=70 + i, ragged=True) for i in range(STEPS)]


def _worker(rank, world, port, out_path, graphed=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bert4rec_amd.distributed import broadcast_parameters, shard_rows
    from bert4rec_amd.engine import make_adamw_config
    torch.cuda.set_device(0)
    eng = _engine()
    broadcast_parameters(eng.params)
    hp = make_adamw_config(num_warmup_steps=2, num_train_steps=10)
    losses = []
    sl = shard_rows(B, rank, world)
    prepared = [eng.prepare_batch({k: v[sl] for k, v in full.items()}) for full in _batches()]   # kept alive: graphs hold pointers
    for epoch in range(EPOCHS):      # graphed: epoch 0 runs eagerly, epoch 1 captures, epoch 2 replays
        for cb, keep in prepared:
            (eng.dp_train_step_graphed if graphed else eng.dp_train_step)(hp, cb)
            torch.cuda.synchronize()
            st = eng.read_state()
            losses.append((st["loss_sum"], st["valid_count"], st["grad_norm"]))
    if rank == 0:
        torch.save({"params": eng.params.cpu(), "losses": losses}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("graphed", [False, True])
def test_two_rank_dp_steps_equal_single_process_steps_on_the_whole_batch(graphed):
    from bert4rec_amd.engine import make_adamw_config
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "rank0.pt")
        mp.spawn(_worker, args=(2, _free_port(), out, graphed), nprocs=2, join=True)
        got = torch.load(out)
    eng = _engine()
    hp = make_adamw_config(num_warmup_steps=2, num_train_steps=10)
    want_losses = []
    prepared = [eng.prepare_batch(full) for full in _batches()]
    for epoch in range(EPOCHS):
        for cb, keep in prepared:
            eng.train_step(hp, cb)
            torch.cuda.synchronize()
            st = eng.read_state()
            want_losses.append((st["loss_sum"], st["valid_count"], st["grad_norm"]))
    for (ls, vc, gn), (wls, wvc, wgn) in zip(got["losses"], want_losses):
        assert vc == wvc and abs(ls - wls) < 1e-3 * abs(wls) and abs(gn - wgn) < 2e-3 * abs(wgn)
    a, b = got["params"].double(), eng.params.cpu().double()
    assert float((a - b).abs().max()) < 5e-5, float((a - b).abs().max())   # 9 Adam steps of lr <= 1e-4


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graphed"])
def test_bench_runs_its_rccl_path_with_one_rank(graph):
    """bench.py with the "nccl" (= RCCL) process group forced on for a single rank: init, parameter broadcast, the
    [grads | sums] all-reduce between backward and the optimizer, barriers and the MAX reduction of the timing -- the
    code the driver runs at N = 2, 4, 8, rehearsed on the one GPU this box has."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "3", "--cpu-steps", "0", "--no-eval",
           "--force-dist"]
    if graph:
        cmd.append("--graph")
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0 and 0 < res["final_loss"] < 12
