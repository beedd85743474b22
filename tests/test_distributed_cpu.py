"""Data-parallel exchange on CPU: world_size 2, gloo.  Each rank back-propagates the loss SUM of its half
of the batch (oracle autograd stands in for the HIP backward, which needs a GPU), the product's one flat all-reduce
combines gradients and loss sums, and the result must equal the single-process step on the whole batch."""
import datetime
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bert4rec_amd import _lib
from bert4rec_amd.distributed import allreduce_step, shard_rows
from bert4rec_amd.engine import Engine, make_model_config
from oracle import bert4rec_oracle as orc

CFG = orc.OracleConfig(vocab_size=41, hidden_size=64, num_layers=1, num_attention_heads=2, max_sequence_length=12, inner_dim=64)
B, L, P = 6, 12, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sum_loss_grads(params, batch):
    leaf = {n: p.detach().clone().requires_grad_(orc.is_trainable(n)) for n, p in params.items()}
    out = orc.model_forward(leaf, batch, CFG, training=False)
    y = batch["masked_lm_ids"]
    lse = torch.logsumexp(out["mlm_logits"], -1)
    picked = torch.gather(out["mlm_logits"], -1, y.unsqueeze(-1)).squeeze(-1)
    mask = (y != 0).float()
    loss_sum = ((lse - picked) * mask).sum()
    names = [n for n in leaf if orc.is_trainable(n)]
    gs = torch.autograd.grad(loss_sum, [leaf[n] for n in names])
    return float(loss_sum), float(mask.sum()), dict(zip(names, gs))


def _rows_for(world):
    return B if world == 2 else 10     # 10 rows over 4 ranks: 3 / 3 / 3 / 1 -- uneven shards, no empty one


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    params = orc.init_params(CFG, 3)
    Bw = _rows_for(world)
    full = orc.synthetic_batch(Bw, L, P, CFG.vocab_size, seed=4, ragged=True)
    sl = shard_rows(Bw, rank, world)
    local = {k: v[sl] for k, v in full.items()}
    loss_sum, count, grads = _sum_loss_grads(params, local)
    eng = Engine(make_model_config(CFG.vocab_size, 64, 1, 2, 12, 64, 0.0, 0.0), "cpu")
    eng.ensure_training_buffers()
    for n, g in grads.items():
        v = eng.view(n, eng.grads)
        v.copy_(g.reshape(v.shape))
    # what b4r_backward(B4R_FLAG_GRAD_TAIL) leaves behind the gradients: [loss_sum, valid_count, correct_masked, correct_all,
    # slots_all, 0, 0, 0]; what b4r_optimizer_step_reduced reads back after the ONE collective
    tail = eng.grad_ext[eng.n_params:]
    tail[:5] = torch.tensor([loss_sum, count, 1.0 + rank, 2.0, float(local["masked_lm_ids"].numel())])
    allreduce_step(eng.grad_ext)
    f = eng.state.view(torch.float32)
    f[_lib.ST_LOSS_SUM:_lib.ST_LOSS_SUM + 5] = tail[:5]
    if rank == 0:
        ret["state"] = eng.state.view(torch.float32).clone()
        ret["grads"] = {n: eng.view(n, eng.grads).clone() for n in grads}
        ret["pad"] = float(eng.grad_ext[eng.n_params + 5:].abs().sum())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_allreduce_of_the_gradient_tail_equals_single_process_step(world):
    """world 4: 10 rows in uneven shards, so the ranks' valid counts differ -- the division happens after the reduction"""
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        state, grads, pad = ret["state"], ret["grads"], ret["pad"]
    params = orc.init_params(CFG, 3)
    Bw = _rows_for(world)
    full = orc.synthetic_batch(Bw, L, P, CFG.vocab_size, seed=4, ragged=True)
    loss_ref, grads_ref, _ = orc.loss_and_grads(params, full, CFG, training=False)
    count = float((full["masked_lm_ids"] != 0).sum())
    assert float(state[_lib.ST_VALID]) == count and float(state[_lib.ST_SLOTS_ALL]) == Bw * P
    assert float(state[_lib.ST_CORRECT_MASKED]) == sum(1.0 + r for r in range(world)) and float(state[_lib.ST_CORRECT_ALL]) == 2.0 * world
    assert pad == 0.0
    assert abs(float(state[_lib.ST_LOSS_SUM]) / count - float(loss_ref)) < 1e-5
    for n, g in grads_ref.items():
        got = grads[n] / count          # the optimizer kernel applies 1/valid_count after the reduction
        assert float((got - g.reshape(got.shape)).abs().max()) < 1e-6 + 1e-4 * float(g.abs().max()), n


def test_shard_rows_partitions_without_overlap():
    for n, w in [(256, 8), (10, 4), (3, 8)]:
        rows = [i for r in range(w) for i in range(*shard_rows(n, r, w).indices(n))]
        assert rows == list(range(n))


# ---- evaluation: users dealt to the ranks, one all-reduce of the metric sums -------------------------------
class _RanksModel:
    """stands in for the ranking model: the 'rank of the ground truth' of every user is carried by the batch itself"""
    engine = None

    def rank_items_tensor(self, batch, candidates, ground_truth, want_ranking=True):
        return None, torch.as_tensor(batch["ranks"], dtype=torch.int32), None, None


def _eval_batches(sizes=(7, 5, 9, 4, 6)):
    g = torch.Generator().manual_seed(11)
    return [{"ranks": torch.randint(1, 60, (n,), generator=g)} for n in sizes]


def _eval_worker(rank, world, port, ret, sizes=(7, 5, 9, 4, 6)):
    from bert4rec_amd import evaluation
    from bert4rec_amd.dataloaders import samplers
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ev = evaluation.get("bert4rec", sampler=samplers.get("random", vocab=list(range(3, 200)), sample_size=10))
    ev.evaluate_batch = lambda model, batch: evaluation.BERT4RecEvaluator.evaluate_batch(
        ev, model, batch, candidates=[[0]] * len(batch["ranks"]), ground_truth=[0] * len(batch["ranks"]))
    ev.evaluate(_RanksModel(), _eval_batches(sizes))
    ret[rank] = {k: float(v) for k, v in ev.get_metrics_results().items()}
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_evaluation_equals_single_process_metrics():
    from bert4rec_amd import evaluation
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_eval_worker, args=(2, port, ret), nprocs=2, join=True)
        r0, r1 = dict(ret[0]), dict(ret[1])
    ms = evaluation.default_metrics()
    for b in _eval_batches():
        for m in ms:
            m.update(b["ranks"].numpy())
    want = {m.name: float(m.result()) for m in ms}
    assert r0 == r1                                   # every rank ends with the metrics of ALL users
    assert want["Valid Ranks"] == 31 == r0["Valid Ranks"]
    for k, v in want.items():
        assert abs(r0[k] - v) < 1e-12, k


@pytest.mark.parametrize("sizes", [(7, 5, 9, 4, 6), (8, 3, 5)])
def test_four_rank_evaluation_with_a_batch_count_not_divisible_by_the_world(sizes):
    """5 batches over 4 ranks (rank 0 takes two), and 3 batches over 4 ranks: rank 3 evaluates NOTHING and must still join the one
    all-reduce with a tensor of the right size and device"""
    from bert4rec_amd import evaluation
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_eval_worker, args=(4, port, ret, sizes), nprocs=4, join=True)
        rs = [dict(ret[r]) for r in range(4)]
    ms = evaluation.default_metrics()
    for b in _eval_batches(sizes):
        for m in ms:
            m.update(b["ranks"].numpy())
    want = {m.name: float(m.result()) for m in ms}
    assert all(r == rs[0] for r in rs)
    assert rs[0]["Valid Ranks"] == sum(sizes)
    for k, v in want.items():
        assert abs(rs[0][k] - v) < 1e-12, k


def _short_worker(rank, world, port, ret):
    from bert4rec_amd import evaluation
    from bert4rec_amd.dataloaders import samplers
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    ev = evaluation.get("bert4rec", sampler=samplers.get("random", vocab=list(range(3, 200)), sample_size=10))
    ev.evaluate_batch = lambda model, batch: evaluation.BERT4RecEvaluator.evaluate_batch(
        ev, model, batch, candidates=[[0]] * len(batch["ranks"]), ground_truth=[0] * len(batch["ranks"]))
    ev.evaluate(_RanksModel(), _eval_batches((4, 6)))             # a clean evaluation first: its metrics must survive the failed one
    good = {k: float(v) for k, v in ev.get_metrics_results().items()}
    n = len(ev.get_metrics())
    ev._dev = (None, torch.zeros(n, dtype=torch.float64), torch.zeros(1, dtype=torch.int64))   # accumulators as the device path keeps them
    ev._short = torch.tensor([rank == 1])                        # the sampler kernel's flag, raised on rank 1 ONLY
    try:
        ev.evaluate(_RanksModel(), _eval_batches())
        ret[rank] = "no error"
    except ValueError as e:
        after = {k: float(v) for k, v in ev.get_metrics_results().items()}
        ret[rank] = ("ValueError", "exclusion lists" in str(e), after == good)
    dist.barrier()
    dist.destroy_process_group()


def test_a_short_row_on_one_rank_raises_on_every_rank_instead_of_hanging():
    """The sampler kernel flags a row with fewer drawable items than the sample size on the DEVICE; evaluate() reads the flag once
    at the end.  With world > 1 a rank that raised before the metric all-reduce left the others waiting in it forever: the flag
    now travels inside that collective and every rank raises the reference's ValueError (popular_random_sampler.py:56-58) after
    it, with the metrics of the failed evaluation rolled back everywhere."""
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_short_worker, args=(2, port, ret), nprocs=2, join=True)
        got = [ret[0], ret[1]]
    assert got == [("ValueError", True, True)] * 2, got
