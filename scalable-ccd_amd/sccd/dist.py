"""Multi-GPU CCD: one process per GPU, grid cells shard naturally, one min-reduce of the TOI.

The reference has no working multi-GPU path (its _multigpu prototype is not compiled,
src/scalable_ccd/cuda/broad_phase/CMakeLists.txt:21).  Here every rank builds the vertex boxes and
derives the same cell grid (from statistics every rank samples alike), is dealt a contiguous window
of grid cells holding an equal share of the sort entries ON THE DEVICE (csrc/build.hip bp_build,
boxes.hip shard_window_k), builds the edge and face boxes of that window only, and sorts, sweeps,
culls and narrows only that window.  A pair is reported from exactly one cell, hence by exactly one rank, and the
ranks exchange exactly one scalar per pass: an all-reduce(min) of the time of impact over
RCCL/xGMI (or gloo on CPU in the tests).
"""
import numpy as np


def balanced_bounds(weights, parts):
    """Cut weighted items (grid cells weighted by their entry counts) into `parts` contiguous
    windows of nearly equal weight: an item goes to the window its midpoint in running weight
    falls into.  Line-for-line mirror of shard_bounds() in csrc/api.hip (C ABI:
    sccd_shard_bounds); tests/test_sharding.py holds the two against each other.
    Returns parts+1 boundaries."""
    w = [int(x) for x in np.asarray(weights).ravel()]
    n = len(w)
    total = sum(w)
    bounds = [0]
    run = 0
    k = 0
    for r in range(1, parts):
        target = total * r // parts
        while k < n and run + w[k] // 2 < target:
            run += w[k]
            k += 1
        bounds.append(k)
    bounds.append(n)
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


class DeviceMin:
    """all-reduce(min) of a step's TOI WITHOUT a host round trip: the TOI sits in a persistent device tensor (written by
    sccd.ccd_mesh_dev on the context's stream), the collective is enqueued stream-ordered behind the step (an event recorded on
    the context's stream, torch's current stream waits for it), reduces the tensor in place, and nobody calls .item() until the
    caller wants the number (value()).  Round 3 built a tensor from a host float and read the result back on every step."""

    def __init__(self, ctx, device, group=None):
        import torch

        self.torch = torch
        self.group = group
        self.word = torch.ones(1, dtype=torch.float64, device=device)
        self.ext = torch.cuda.ExternalStream(ctx.stream_ptr(), device=device)
        self.event = torch.cuda.Event()
        self.done = torch.cuda.Event()

    def ptr(self):
        return self.word.data_ptr()

    def reduce(self):
        import torch.distributed as dist

        # (a process group of ONE rank still runs the collective: the one-rank RCCL run of bench.py exercises this very path)
        if not dist.is_available() or not dist.is_initialized():
            return
        # The collective is issued with the CONTEXT'S stream as torch's current one: the process group then orders its own stream behind
        # that stream's work (the step, and the copy of its TOI into the word) and that stream behind the collective -- the next
        # step's result cannot land in the word before the collective has read it -- with the two event hops it makes anyway.
        # (Round 5 went through torch's default stream: four more event calls per step, ~0.1 ms of host time in front of the next
        # step's first launch on every rank.)  Asynchronous w.r.t. the host.
        with self.torch.cuda.stream(self.ext):
            dist.all_reduce(self.word, op=dist.ReduceOp.MIN, group=self.group)

    def value(self):
        with self.torch.cuda.stream(self.ext):
            return float(self.word.item())


class GlobalPrior:
    """The speculative bound of a multi-GPU job: every rank starts a step from 1.125 x the REDUCED result of the job's last step on the
    mesh (a rank's own last result is a worse bound -- its shard's earliest impact may lie far behind the job's -- and a rank must not
    redo a step just because ITS shard has nothing below the bound).  run(bound) -> (toi, stats) is one ccd() of this rank from that bound
    (sccd.ccd_mesh_from on a context with SHARD_RANK / SHARD_COUNT set).  The reduced minimum is below the bound: exact, done.  It IS the
    bound (no rank found anything below it): every rank knows, and every rank redoes the step from 1.  Exact either way; one more
    collective only for a bound that broke."""

    def __init__(self, group=None, device=None, scalar_f32=False):
        self.group, self.device = group, device
        self.bound = 1.0
        self.hits = self.misses = 0
        # the float build (sccd.OPT_SCALAR = 1) starts its kernels from (float)bound and returns THAT value when nothing lies below it:
        # the bound compared with the reduced result must be the rounded one, or a bound that rounded down reads as a hit
        self.scalar_f32 = bool(scalar_f32)

    def step(self, run):
        bound = self.bound
        if self.scalar_f32:
            import numpy as np

            bound = float(np.float32(bound))
            if not 0.0 < bound < 1.0:
                bound = 1.0
        toi, st = run(bound)
        toi = allreduce_min(toi, group=self.group, device=self.device)
        if bound < 1.0:
            if toi < bound:
                self.hits += 1
            else:  # nothing below the bound anywhere: the earliest impact lies at or beyond it
                self.misses += 1
                toi, st = run(1.0)
                toi = allreduce_min(toi, group=self.group, device=self.device)
        self.bound = min(1.0, 1.125 * toi) if 0.0 < toi < 1.0 else 1.0
        return toi, st

    def forget(self):
        self.bound = 1.0


def ccd_sharded(run_pass, rank, world, group=None, device=None, prepare=None, reduce_between_passes=False):
    """ccd() across `world` ranks.

    run_pass(is_vf, toi) -> (toi, stats) runs this rank's share of the VF or EE pass starting
    from the bound `toi` (sccd.ccd_mesh_pass on a context configured with SHARD_RANK/COUNT).
    The result is the minimum over all accepted domains of all ranks, so ONE all-reduce(min) at
    the end is enough: a rank's own VF result seeds its EE pass (ccd.cu:125-143) and only prunes.
    reduce_between_passes=True also shares the VF result before the EE pass (a tighter pruning
    bound on every rank for the price of a second collective -- not worth it at millisecond steps).
    """
    if prepare is not None:
        prepare()
    toi = 1.0
    stats = {}
    for is_vf in (True, False):
        toi, st = run_pass(is_vf, toi)
        for k, v in (st or {}).items():
            stats[k] = stats.get(k, 0) + v
        if reduce_between_passes and is_vf:
            toi = allreduce_min(toi, group=group, device=device)
    toi = allreduce_min(toi, group=group, device=device)
    return toi, stats
