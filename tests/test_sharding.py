"""Multi-GPU host logic on CPU: one process per rank over gloo (world_size 2), exactly the code
path bench.py uses with RCCL -- entry-balanced windows of grid cells, per-rank VF/EE passes, one
all-reduce(min) of the time of impact per pass.  The per-rank compute is stubbed with the CPU
oracle here (tests may use it); the GPU version of the same flow is tests/test_gpu_parity.py::
test_sharded_sweeps_partition_the_pair_set and bench.py --gpus N."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_balanced_bounds_cover_and_balance():
    from sccd import dist as sdist

    rng = np.random.default_rng(0)
    w = rng.integers(0, 1000, size=5000)
    for parts in (1, 2, 3, 8):
        b = sdist.balanced_bounds(w, parts)
        assert b[0] == 0 and b[-1] == len(w) and all(x <= y for x, y in zip(b, b[1:]))
        sums = [int(w[b[i]:b[i + 1]].sum()) for i in range(parts)]
        assert sum(sums) == int(w.sum())
        assert max(sums) - min(sums) <= 2 * 1000  # within one item's weight of the ideal split
    assert sdist.balanced_bounds([], 4) == [0, 0, 0, 0, 0]
    # one huge item cannot be split: it lands in exactly one window
    b = sdist.balanced_bounds([1, 10**9, 1, 1], 2)
    assert b[0] == 0 and b[-1] == 4 and b[1] in (1, 2)
    # all the weight in one cell: the other windows are empty, nothing is lost
    b = sdist.balanced_bounds([0, 0, 7, 0], 3)
    assert b == sorted(b) and b[0] == 0 and b[-1] == 4


def test_shard_bounds_abi_matches_python_mirror():
    """The split the library applies to the per-cell histogram (host code, runs without a GPU)."""
    import sccd
    from sccd import dist as sdist

    rng = np.random.default_rng(3)
    cases = [rng.integers(0, 5000, size=n).astype(np.uint32) for n in (1, 2, 7, 64, 1000, 1024)]
    cases += [np.zeros(16, np.uint32), np.array([0, 0, 9, 0], np.uint32), np.full(1024, 0xFFFFFFFF, np.uint32)]
    for w in cases:
        for parts in (1, 2, 3, 4, 8, 13):
            assert sccd.shard_bounds(w, parts) == sdist.balanced_bounds(w, parts)
    assert sccd.shard_bounds(np.zeros(0, np.uint32), 3) == [0, 0, 0, 0]
    with pytest.raises(ValueError):
        sccd.shard_bounds(np.zeros(4, np.uint32), 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "scalable-ccd_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist

    import orc
    from sccd import dist as sdist
    from sccd import scenes

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        V0, V1, E, F = scenes.triangle_soup(150, seed=21)
        vb, eb, fb = orc.build_boxes(V0, V1, E, F)
        vf, _, _ = orc.sort_and_sweep(vb, fb)
        ee, _, _ = orc.sort_and_sweep(eb)
        checked = {"n": 0}

        def run_pass(is_vf, toi):
            pairs = vf if is_vf else ee
            # this rank's share (weights = 1 per pair here)
            b = sdist.balanced_bounds(np.ones(len(pairs), np.int64), world)
            mine = pairs[b[rank]:b[rank + 1]]
            checked["n"] += len(mine)
            t, _ = orc.narrow_phase_mt(V0, V1, E, F, mine, is_vf, toi=toi, nthreads=2)
            return t, {"n_pairs": len(mine)}

        toi, stats = sdist.ccd_sharded(run_pass, rank, world)
        want, n_vf, n_ee = orc.ccd(V0, V1, E, F)
        q.put((rank, toi, want, checked["n"], n_vf + n_ee, sdist.allreduce_min(float(rank + 5))))
    finally:
        dist.destroy_process_group()


def test_sharded_ccd_two_ranks_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    (r0, toi0, want0, n0, tot0, m0), (r1, toi1, want1, n1, tot1, m1) = out
    assert toi0 == toi1 == want0 == want1 and toi0 < 1  # every rank ends with the global minimum
    assert n0 + n1 == tot0 == tot1                      # the shards partition the queries
    assert m0 == m1 == 5.0                               # all-reduce(min) really reduces over ranks


def _prior_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "scalable-ccd_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist

    import orc
    from sccd import dist as sdist
    from sccd import scenes

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        V0, V1, E, F = scenes.triangle_soup(150, seed=21)
        gp = sdist.GlobalPrior()
        log = []
        for scale in (1.0, 1.0, 0.4, 0.4, 1.0):  # (the step cut to 0.4: every impact moves to 2.5 x the time -- the bound breaks; then back)
            W1 = V0 + scale * (V1 - V0)
            vb, eb, fb = orc.build_boxes(V0, W1, E, F)
            vf, _, _ = orc.sort_and_sweep(vb, fb)
            ee, _, _ = orc.sort_and_sweep(eb)
            calls = []

            def run(bound):  # this rank's ccd() from the bound: both passes on its share of the pairs, toi in / out
                calls.append(bound)
                t = bound
                for is_vf, pairs in ((True, vf), (False, ee)):
                    b = sdist.balanced_bounds(np.ones(len(pairs), np.int64), world)
                    t, _ = orc.narrow_phase_mt(V0, W1, E, F, pairs[b[rank]:b[rank + 1]], is_vf, toi=t, nthreads=2)
                return t, {}

            toi, _ = gp.step(run)
            log.append((toi, orc.ccd(V0, W1, E, F)[0], tuple(calls), gp.bound))
        q.put((rank, log, gp.hits, gp.misses))
    finally:
        dist.destroy_process_group()


def test_global_prior_two_ranks_gloo():
    """sccd.dist.GlobalPrior over gloo: both ranks start from 1.125 x the last REDUCED result, a bound that breaks is redone from 1 on
    every rank, and every step ends with the oracle's TOI on both."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prior_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=480) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    (_, log0, h0, m0), (_, log1, h1, m1) = out
    assert log0 == log1 and (h0, m0) == (h1, m1)  # the ranks agree on every result, every bound and every redo
    for toi, want, calls, bound in log0:
        assert toi == want
        assert bound == (min(1.0, 1.125 * want) if 0 < want < 1 else 1.0)
    assert log0[0][2] == (1.0,) and len(log0[1][2]) == 1 and log0[1][2][0] < 1.0  # first from 1, then from the bound
    assert len(log0[2][2]) == 2 and log0[2][2][1] == 1.0  # the broken bound: again from 1
    assert h0 >= 2 and m0 >= 1


def test_global_prior_rounds_its_bound_like_the_float_build():
    """The float build starts from (float)bound and hands THAT back when nothing lies below it: with the unrounded bound a result that
    is the rounded-down bound would read as an impact.  One rank, a fake run() that behaves like the library."""
    from sccd import dist as sdist

    t_true = float.fromhex("0x1.a1eb220000000p-2")  # the job's earliest impact, a float whose 1.125-fold rounds DOWN to float
    seen = []

    def run_f32(bound):  # min(float(bound), earliest impact below it) -- the earliest impact is beyond the second step's bound
        b = float(np.float32(bound))
        seen.append(bound)
        hit = t_true if len(seen) == 1 else 0.47
        return (hit if hit < b else b), {}

    gp = sdist.GlobalPrior(scalar_f32=True)
    assert gp.step(run_f32)[0] == t_true
    assert gp.bound == 1.125 * t_true
    bound_f32 = float(np.float32(gp.bound))
    assert bound_f32 != gp.bound and bound_f32 < 0.47  # (the case: the double bound is no float)
    toi, _ = gp.step(run_f32)
    assert seen[1] == bound_f32 and seen[2] == 1.0 and toi == 0.47 and (gp.hits, gp.misses) == (0, 1)  # the rounded bound came back: redone from 1


def test_single_rank_needs_no_process_group():
    from sccd import dist as sdist

    assert sdist.allreduce_min(0.25) == 0.25
    toi, st = sdist.ccd_sharded(lambda is_vf, toi: (min(toi, 0.5 if is_vf else 0.75), {"x": 1}), 0, 1)
    assert toi == 0.5 and st == {"x": 2}


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu_and_agree_with_one_rank():
    """bench.py's N > 1 path end to end on the GPU box: two processes (gloo, both on cuda:0), each
    building / sorting / sweeping / narrowing its window of grid cells, one all-reduce(min).
    Queries must add up to the single-rank count and the TOI must be the same."""
    import json
    import subprocess

    env = dict(os.environ, SCCD_BENCH_BACKEND="gloo")
    common = ["bench.py", "--cloth-n", "160", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]

    def last_json(out):
        lines = [ln for ln in out.splitlines() if ln.startswith("{\"metric\"")]
        assert lines, out[-2000:]
        return json.loads(lines[-1])

    one = subprocess.run([sys.executable] + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    port = _free_port()
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port)] + common + ["--gpus", "2"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = last_json(one.stdout), last_json(two.stdout)
    assert b["n_gpus"] == 2 and a["n_gpus"] == 1
    assert b["config"]["queries_per_step"] == a["config"]["queries_per_step"] > 0
    assert b["config"]["toi"] == a["config"]["toi"]
    # (the candidates are a work metric of the cell grid, and a sharded rank sizes its grid from a sample of the edge and
    # face boxes -- it never builds them all: the same pair set from a slightly different grid)
    assert abs(b["config"]["candidates_per_step"] - a["config"]["candidates_per_step"]) < 0.25 * a["config"]["candidates_per_step"]


def _warmup_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import time

    import torch
    import torch.distributed as dist

    import bench

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        calls = {"n": 0}

        def step():  # a step holds a collective, like bench.py's; rank 1 is the slower rank
            time.sleep(0.001 * (1 + 3 * rank))
            t = torch.tensor([float(rank)])
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            calls["n"] += 1

        def agree(n):
            t = torch.tensor([n], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item())

        n = bench.clock_warmup(step, 0.2, agree)
        # what follows the warm-up in bench.py: more collectives, which must still pair up
        for _ in range(5):
            step()
        dist.barrier()
        q.put((rank, n, calls["n"]))
    finally:
        dist.destroy_process_group()


def test_clock_warmup_runs_the_same_number_of_steps_on_every_rank():
    """bench.py warms the clocks with its own steps for a fixed TIME; with several ranks a step holds an all-reduce, so the ranks
    must agree on a COUNT -- each rank ending the loop by its own clock left them out of step and a 2-rank job hung (round 4)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_warmup_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, c0), (_, n1, c1) = out
    assert n0 == n1 >= 3 and c0 == c1 == n0 + 5


def test_clock_warmup_single_rank_is_time_bound_and_can_be_switched_off():
    sys.path.insert(0, ROOT)
    import time

    import bench

    calls = {"n": 0}

    def step():
        time.sleep(0.002)
        calls["n"] += 1

    assert bench.clock_warmup(step, 0) == 0 and calls["n"] == 0
    t0 = time.perf_counter()
    n = bench.clock_warmup(step, 0.05)
    assert n == calls["n"] >= 2 and time.perf_counter() - t0 < 1.0
    assert bench.clock_warmup(step, 0.05, lambda k: 4) == 7  # three timed steps + what the ranks agreed on
