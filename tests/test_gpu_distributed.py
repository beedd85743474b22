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


def _fit_model():
    from bert4rec_amd import config, models
    from bert4rec_amd.models.components import networks
    from bert4rec_amd.trainers import optimizers
    cfgd = {**config.get_encoder_config("ml-1m_64"), "max_sequence_length": L, "output_dropout": 0.0, "attention_dropout": 0.0}
    model = models.BERT4RecModel(networks.Bert4RecEncoder(V, seed=4, **cfgd))
    model.compile(optimizer=optimizers.get("adamw", init_lr=1e-3, num_warmup_steps=2, num_train_steps=50))
    return model


def _fit_batches():
    from oracle import bert4rec_oracle as orc
    from bert4rec_amd.dataloaders.dataloader_utils import BatchedDataset
    return BatchedDataset([orc.synthetic_batch(B // 2, L, P, V, seed=90 + i, ragged=True) for i in range(5)])   # 5: the last round has a batch for rank 0 only


def _fit_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    model = _fit_model()
    hist = model.fit(_fit_batches(), epochs=2, verbose=0)
    torch.cuda.synchronize()
    torch.save({"weights": model.get_weights(), "loss": hist.history["loss"], "steps": model._trained_steps},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_fit_shards_the_batches_and_equals_single_process_training_on_the_joined_batches():
    """BERT4RecModel.fit under an initialised process group: rank r trains on batches r, r + 2, ... of every epoch, the gradients
    are summed over the ranks -- i.e. one step on the two batches joined.  The 5th batch of the epoch has no partner: rank 0 trains
    on it, rank 1 joins that round with zeros (no batch is dropped).  Both ranks end with the same weights, and those are the
    single-process weights after training on [b0+b1, b2+b3, b4]."""
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_fit_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(tmp, "rank0.pt")), torch.load(os.path.join(tmp, "rank1.pt"))
    assert r0["steps"] == r1["steps"] == 3 * 2                      # 5 batches -> 3 rounds per epoch (the last one padded), 2 epochs
    for k in r0["weights"]:
        assert torch.equal(r0["weights"][k], r1["weights"][k]), k   # identical updates on every rank
    assert r0["loss"] == r1["loss"]                                 # the logged sums are the all-reduced ones
    bs = _fit_batches().batches
    joined = [{k: torch.cat([bs[2 * j][k], bs[2 * j + 1][k]]) for k in bs[0]} for j in range(2)] + [bs[4]]   # every batch consumed
    from bert4rec_amd.dataloaders.dataloader_utils import BatchedDataset
    single = _fit_model()
    hist = single.fit(BatchedDataset(joined), epochs=2, verbose=0)
    w = single.get_weights()
    for k in w:
        scale = max(1e-3, float(w[k].abs().max()))
        assert float((w[k] - r0["weights"][k]).abs().max()) < 2e-5 * scale + 2e-6, k
    assert max(abs(a - b) for a, b in zip(hist.history["loss"], r0["loss"])) < 1e-4
