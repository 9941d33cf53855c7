"""The N>1 path on CPU: world_size-2 gloo processes exercise partition + speaker broadcast + length gather
(chatttsplus_amd/dist.py) with a stub per-rank synthesis function."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from chatttsplus_amd import dist as cdist


def test_partition_balanced_and_complete():
    lengths = [5, 100, 7, 64, 64, 3, 90, 12, 33]
    for world in (1, 2, 4, 8):
        shards = cdist.partition(lengths, world)
        assert sorted(i for s in shards for i in s) == list(range(len(lengths)))
        loads = [sum(lengths[i] for i in s) for s in shards]
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
        if world == 2:
            assert abs(loads[0] - loads[1]) <= max(lengths)
    assert cdist.partition([], 4) == [[], [], [], []]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lengths = [10, 40, 20, 30, 25, 5, 60]
        spk_idx = [0, 1, 2, 0, 1, 2, 0]
        table = torch.arange(3 * 8, dtype=torch.float32).view(3, 8) if rank == 0 else None

        def run_local(indices, rows):
            # stub synthesis: generated length = prompt length + first speaker-row element (checks the broadcast)
            return [lengths[i] + int(rows[j, 0].item()) for j, i in enumerate(indices)]

        mine, full = cdist.sharded_generate(lengths, spk_idx, table, 3, 8, torch.device("cpu"), run_local)
        q.put((rank, mine, full))
    finally:
        dist.destroy_process_group()


def test_sharded_generate_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lengths = [10, 40, 20, 30, 25, 5, 60]
    spk_first = [0, 8, 16, 0, 8, 16, 0]
    expect = [l + s for l, s in zip(lengths, spk_first)]
    seen = []
    for rank, mine, full in res:
        assert full == expect
        seen += mine
    assert sorted(seen) == list(range(len(lengths)))
