"""world_size-2 gloo rehearsal of the multi-GPU path: chunk sharding is disjoint and complete, and
the benchmark's reduction (MAX elapsed, SUM units) behaves; no data-path collective exists."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from margin_amd import sharding, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = sharding.chunk_seeds(rank, 3)
    units = 0
    for s in seeds:
        c = synth.make_ont_chunk(seed=s, region_bp=10_000, n_sites=20, coverage=6)
        units += c.units
    elapsed, total = sharding.reduce_elapsed_and_units(dist, 1.0 + rank, float(units))
    mine = sharding.shard_chunks(10, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, (seeds, mine, units))
    dist.barrier()
    if rank == 0:
        out.put((elapsed, total, gathered))
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    elapsed, total, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert elapsed == 2.0                                  # MAX over ranks
    assert total == sum(g[2] for g in gathered)            # SUM over ranks
    seeds0, seeds1 = gathered[0][0], gathered[1][0]
    assert not set(seeds0) & set(seeds1)                   # ranks own different chunks
    assert sorted(gathered[0][1] + gathered[1][1]) == list(range(10))
