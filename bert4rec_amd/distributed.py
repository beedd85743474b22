"""Data-parallel exchange for the train step: ONE all-reduce per step over one flat fp32 buffer
[ gradient of loss_SUM | loss_sum | valid_count | correct_masked | correct_all | slots_all ].

Every rank back-propagates the un-normalised sum of its local per-slot losses; after the reduction the optimizer kernel
divides by the reduced valid_count, so the update equals the reference's single-process step on the concatenated global
batch (trainer_utils.py:19-22 normalises by the batch-global count; adam_w_optimizer.py:111-112 clips the global norm
of that gradient).  Backend: torch.distributed "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

N_SUMS = 5  # loss_sum, valid_count, correct_masked, correct_all, slots_all: state words ST_LOSS_SUM .. ST_SLOTS_ALL
TAIL = 8    # floats reserved behind the gradients (keeps the buffer a multiple of 4 floats)


def alloc_grad_buffer(n_params: int, device) -> torch.Tensor:
    """Flat buffer holding the gradients followed by TAIL floats for the per-step sums."""
    return torch.zeros(n_params + TAIL, dtype=torch.float32, device=device)


def allreduce_step(grad_ext: torch.Tensor, group: Optional[dist.ProcessGroup] = None, single_rank_too: bool = False) -> None:
    """Sum [gradients | per-step sums] over the data-parallel group, in place: one collective, nothing else.  The backward wrote
    the sums into the tail (B4R_FLAG_GRAD_TAIL) and the optimizer step reads them back from there (b4r_optimizer_step_reduced).
    single_rank_too: issue the collective even in a group of one (rehearsal of the RCCL path on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_world_size(group) == 1 and not single_rank_too:
        return
    dist.all_reduce(grad_ext, op=dist.ReduceOp.SUM, group=group)


def broadcast_parameters(params: torch.Tensor, src: int = 0, group: Optional[dist.ProcessGroup] = None) -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(params, src=src, group=group)


def shard_rows(n_rows: int, rank: int, world: int) -> slice:
    """Contiguous shard of a global batch / user list for this rank (weak scaling keeps the per-rank batch fixed)."""
    per = (n_rows + world - 1) // world
    return slice(min(n_rows, rank * per), min(n_rows, (rank + 1) * per))
