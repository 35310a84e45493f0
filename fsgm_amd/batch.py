"""Frame-batch sharding across the GPUs of one node.

Frames are independent, so the path shards embarrassingly: frame i belongs to rank i mod G, each
rank runs its frames on its own GPU through the C ABI, and there is NO collective in the data
path.  torch.distributed is used only as plumbing -- rendezvous, and an optional gather of the
small per-frame results (bestD / minC maps) to rank 0.  Backend "nccl" (= RCCL over xGMI) on GPUs,
"gloo" in the CPU tests.
"""
import os


def shard_indices(n_frames, rank, world):
    """Frame indices owned by `rank` (round-robin: frame i -> rank i mod world)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_frames, world))


def dist_env():
    """(rank, world, local_rank) from the torchrun environment; (0, 1, 0) when not launched by it."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def run_sharded(frames, compute, *, rank=None, world=None, gather=True, group=None):
    """Run `compute(list_of_frames) -> list_of_results` on this rank's shard of `frames`.

    compute is normally a closure over fsgm_amd.calc_cost_sgm_batch(..., device=local_rank).
    With gather=True every rank returns the full, frame-ordered result list (all_gather_object of
    the per-rank lists -- host-side, off the data path); otherwise each rank returns
    {frame_index: result} for its own frames.
    """
    if rank is None or world is None:
        r, w, _ = dist_env()
        rank = r if rank is None else rank
        world = w if world is None else world
    mine = shard_indices(len(frames), rank, world)
    results = compute([frames[i] for i in mine]) if mine else []
    if len(results) != len(mine):
        raise RuntimeError("compute() must return one result per frame")
    local = dict(zip(mine, results))
    if not gather or world == 1:
        return [local[i] for i in range(len(frames))] if world == 1 else local
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, local, group=group)
    merged = {}
    for part in parts:
        merged.update(part)
    return [merged[i] for i in range(len(frames))]
