"""Frame sharding with world_size 2 over gloo on the CPU (the N>1 path of SURVEY 8(e)).

The per-frame compute injected here is the CPU oracle -- this test exercises the sharding /
gather plumbing of fsgm_amd.batch, not the kernels (those are covered by the -m gpu tests; on a
GPU box the same plumbing runs with calc_cost_sgm_batch(device=local_rank))."""
import os
import socket
import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from fsgm_amd import synth, batch


def test_shard_indices_partition_the_batch():
    for n in (0, 1, 7, 8, 9):
        for world in (1, 2, 3, 8):
            parts = [batch.shard_indices(n, r, world) for r in range(world)]
            assert sorted(i for p in parts for i in p) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        batch.shard_indices(4, 2, 2)


def _make_frames(n, W=40, H=28, D=16):
    out = []
    for s in range(n):
        I1, I2 = synth.image_pair(W, H, D, seed=50 + s)
        pd0, nd, off = synth.epi_maps(W, H, "general", seed=60 + s)
        out.append((I1, I2, pd0, nd, off))
    return out


def _worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle
        frames = _make_frames(n_frames)
        seen = []

        def compute(fs):
            seen.append(len(fs))
            return [pyoracle.calc_cost_sgm(I1, I2, 16, 0.3, pd0, nd, off, 6, 64, 8) for (I1, I2, pd0, nd, off) in fs]

        full = batch.run_sharded(frames, compute)                       # env-derived rank/world, gathered
        mine = batch.run_sharded(frames, compute, gather=False)
        dist.barrier()
        q.put((rank, seen, [int(bd.sum()) + int(mc.sum()) for bd, mc in full], sorted(mine.keys())))
    finally:
        dist.destroy_process_group()


def test_two_ranks_shard_frames_and_gather_over_gloo(oracle):
    n_frames, world = 5, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frames = _make_frames(n_frames)
    want = []
    for (I1, I2, pd0, nd, off) in frames:
        bd, mc = oracle.calc_cost_sgm(I1, I2, 16, 0.3, pd0, nd, off, 6, 64, 8)
        want.append(int(bd.sum()) + int(mc.sum()))
    assert got[0][1] == [3, 3] and got[1][1] == [2, 2]                  # frames 0,2,4 | 1,3 -- twice
    assert got[0][2] == want and got[1][2] == want                      # every rank holds the full ordered result
    assert got[0][3] == [0, 2, 4] and got[1][3] == [1, 3]
