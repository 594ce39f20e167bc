"""Multi-GPU CCD: one process per GPU, pairs shard naturally, one min-reduce of the TOI.

The reference has no working multi-GPU path (its _multigpu prototype is not compiled,
src/scalable_ccd/cuda/broad_phase/CMakeLists.txt:21).  Here every rank builds and sorts the
(small) box lists redundantly, sweeps only its candidate-balanced share of the sorted rows,
runs the narrow phase on the pairs it emitted, and the ranks exchange exactly one scalar per
pass: an all-reduce(min) of the time of impact over RCCL/xGMI (or gloo on CPU in the tests).
"""
import numpy as np


def balanced_bounds(weights, parts):
    """Split rows with the given weights into `parts` contiguous shards of nearly equal total
    weight.  Mirrors shard_rows() of csrc/api.hip (weight = candidates + 1 per row).
    Returns parts+1 boundaries."""
    w = np.asarray(weights, dtype=np.uint64) + np.uint64(1)
    total = int(w.sum())
    run = np.concatenate([[0], np.cumsum(w, dtype=np.uint64)[:-1]]).astype(np.uint64) if len(w) else np.zeros(0, np.uint64)
    bounds = [0]
    for r in range(1, parts):
        target = total * r // parts
        # first row whose running prefix reaches the target
        idx = int(np.searchsorted(run, np.uint64(target), side="left")) if len(w) else 0
        bounds.append(min(idx, len(w)))
    bounds.append(len(w))
    return bounds


def allreduce_min(value, group=None, device=None):
    """min over all ranks of a python float (RCCL when the tensor is on a GPU, gloo on CPU)."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return float(t.item())


def ccd_sharded(run_pass, rank, world, group=None, device=None, prepare=None):
    """ccd() across `world` ranks.

    run_pass(is_vf, toi) -> (toi, stats) runs this rank's share of the VF or EE pass starting
    from the bound `toi` (sccd.ccd_mesh_pass on a context configured with SHARD_RANK/COUNT).
    The VF result seeds the EE pass on every rank (ccd.cu:125-143), hence two reductions.
    """
    if prepare is not None:
        prepare()
    toi = 1.0
    stats = {}
    for is_vf in (True, False):
        toi, st = run_pass(is_vf, toi)
        for k, v in (st or {}).items():
            stats[k] = stats.get(k, 0) + v
        toi = allreduce_min(toi, group=group, device=device)
        if toi <= 0:  # nothing can beat 0 (narrow_phase.cu:136)
            pass
    return toi, stats
