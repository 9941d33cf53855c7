"""Multi-GPU batched synthesis: utterances are independent (reference: sequential slices of 4 with no
cross-slice state, pipelines/chattts_plus_pipeline.py:391-397), so the path shards by utterance with
one process per GPU and NO data-path collective.  The only exchange is one broadcast of the
speaker-embedding table from rank 0 per request (RCCL over xGMI when the backend is "nccl"; gloo on CPU
in the tests) plus an optional gather of per-utterance lengths for aggregate throughput.

Whenever a process group is initialised the collectives RUN, also at world size 1: a one-rank "nccl" group takes the same RCCL code
path (communicator creation, device-tensor broadcast / all-reduce) as an eight-rank one, so it can be exercised on a one-GPU box
(tests/test_gpu_rccl.py, `bench.py --force-pg`).  Without a group the functions are plain local copies.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def partition(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Length-balanced static split: sort by length (desc) and snake-assign, so every rank's padded batch
    has a similar max length and token count.  Returns utterance indices per rank (each sorted ascending)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    shards: List[List[int]] = [[] for _ in range(world)]
    for j, idx in enumerate(order):
        r = j % (2 * world)
        r = r if r < world else 2 * world - 1 - r
        shards[r].append(idx)
    return [sorted(s) for s in shards]


def broadcast_speakers(table: Optional[torch.Tensor], n_spk: int, dim: int, device, src: int = 0) -> torch.Tensor:
    """rank `src` holds table [n_spk, dim] fp32; every rank returns a copy on `device`.  <= 786 KB even for
    256 distinct speakers: latency-bound on xGMI, done once per request before the decode loop."""
    if not (dist.is_available() and dist.is_initialized()):
        assert table is not None
        return table.to(device=device, dtype=torch.float32)
    if dist.get_rank() == src:
        buf = table.to(device=device, dtype=torch.float32).contiguous()
        assert tuple(buf.shape) == (n_spk, dim)
    else:
        buf = torch.empty(n_spk, dim, dtype=torch.float32, device=device)
    dist.broadcast(buf, src=src)
    return buf


def broadcast_seed(seed: int, device, src: int = 0) -> int:
    """The request's noise seed as drawn on rank `src` (8 bytes, same process group as the speaker table); world 1: the local draw."""
    if not (dist.is_available() and dist.is_initialized()):
        return int(seed)
    buf = torch.tensor([int(seed)], dtype=torch.int64, device=device)
    dist.broadcast(buf, src=src)
    return int(buf.item())


def sharded_generate(lengths: Sequence[int], speaker_index: Sequence[int], speaker_table: Optional[torch.Tensor],
                     n_spk: int, dim: int, device,
                     run_local: Callable[[List[int], torch.Tensor], List[int]]) -> Tuple[List[int], List[int]]:
    """Drives one sharded request.  `run_local(indices, speaker_rows)` synthesises this rank's utterances
    (speaker_rows[i] is the embedding of utterance indices[i]) and returns the generated length of each.
    Returns (my utterance indices, all generated lengths in global order on every rank)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    table = broadcast_speakers(speaker_table, n_spk, dim, device)
    mine = partition(lengths, world)[rank]
    rows = table[torch.as_tensor([int(speaker_index[i]) for i in mine], dtype=torch.long, device=table.device)] if mine else table[:0]
    out_len = run_local(mine, rows)
    assert len(out_len) == len(mine)
    total = len(lengths)
    full = torch.zeros(total, dtype=torch.int64, device=device)
    if mine:
        full[torch.as_tensor(mine, dtype=torch.long, device=device)] = torch.as_tensor(out_len, dtype=torch.int64, device=device)
    if dist.is_initialized():
        dist.all_reduce(full, op=dist.ReduceOp.SUM)          # disjoint supports -> gather of lengths
    return mine, full.cpu().tolist()
