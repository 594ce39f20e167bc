"""GPU parity tests proper: the HIP path through the C ABI against the CPU oracle on the same
seeded inputs.  Bars: boxes and overlap-pair sets bit-identical (compared as sorted sets);
time of impact bit-equal (the stated tolerance of north_star is |dTOI| <= 1e-6; both arithmetic
contracts are expected to agree exactly with the oracle run under the same contract)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from sccd import scenes

pytestmark = pytest.mark.gpu
TOI_TOL = 1e-6
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "golden.json")


def _sorted(p):
    p = np.ascontiguousarray(p, dtype=np.int32).reshape(-1, 2)
    if len(p) == 0:
        return p
    return p[np.lexsort((p[:, 1], p[:, 0]))]


def _scene(name):
    return {
        "cloth_ball_10k": lambda: scenes.cloth_ball(),
        "cloth_ball_small": lambda: scenes.cloth_ball(20, 1, seed=3),
        "soup_400": lambda: scenes.triangle_soup(400, seed=11),
        "soup_dense": lambda: scenes.triangle_soup(1500, seed=5, size=0.08, motion=0.2),
        "folded_120": lambda: scenes.folded_cloth(120),
    }[name]()


# ---- boxes ------------------------------------------------------------------------------------
@pytest.mark.parametrize("inflation", [0.0, 1e-3, 0.25])
def test_build_boxes_bit_exact(sccd, ctx, orc, inflation):
    V0, V1, E, F = scenes.cloth_ball(30, 2, seed=5)
    # awkward coordinates: zeros, negatives, tiny and huge magnitudes
    V0[0] = [0.0, -0.0, 1e-310]
    V1[0] = [-1e-308, 0.0, -1e-310]
    V0[1] = [1e300, -1e300, 5e-324]
    V1[1] = [1e300, -1e300, 0.0]
    vb, eb, fb = orc.build_boxes(V0, V1, E, F, inflation)
    g_vb = sccd.build_vertex_boxes(V0, V1, inflation, ctx=ctx)
    g_eb = sccd.build_edge_boxes(g_vb, E, ctx=ctx)
    g_fb = sccd.build_face_boxes(g_vb, F, ctx=ctx)
    assert g_vb.tobytes() == vb.tobytes()
    assert g_eb.tobytes() == eb.tobytes()
    assert g_fb.tobytes() == fb.tobytes()


def test_device_boxes_from_mesh(sccd, ctx, orc):
    V0, V1, E, F = _scene("cloth_ball_small")
    vb, eb, fb = orc.build_boxes(V0, V1, E, F, 1e-3)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    for dev, ref in zip(sccd.DeviceAABBs.from_mesh(mesh, 1e-3), (vb, eb, fb)):
        assert len(dev) == len(ref)
        assert dev.download().tobytes() == ref.tobytes()  # element order, bit for bit


# ---- broad phase ------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["cloth_ball_10k", "soup_400", "soup_dense", "folded_120"])
@pytest.mark.parametrize("algo", [0, 1, 2, 3])  # auto, plain SAP, filter/queue/confirm STQ, direct exact sweep
def test_overlap_pairs_identical(sccd, ctx, orc, name, algo):
    V0, V1, E, F = _scene(name)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    want_vf, _, _ = orc.sort_and_sweep(vb, fb, nthreads=8)
    want_ee, _, _ = orc.sort_and_sweep(eb, nthreads=8)
    ctx.set_option(sccd.OPT_SWEEP_ALGO, algo)
    try:
        bp = sccd.BroadPhase(ctx)
        bp.build(sccd.DeviceAABBs(vb, ctx), sccd.DeviceAABBs(fb, ctx))
        got_vf = bp.detect_overlaps()
        assert bp.is_complete()
        bp.build(sccd.DeviceAABBs(eb, ctx))
        got_ee = bp.detect_overlaps()
    finally:
        ctx.set_option(sccd.OPT_SWEEP_ALGO, 0)
    assert np.array_equal(_sorted(got_vf), want_vf)
    assert np.array_equal(_sorted(got_ee), want_ee)


@pytest.mark.parametrize("cell_factor", ["0", "1", "2.5", "16"])
def test_pair_set_independent_of_cell_grid(sccd, ctx, orc, cell_factor, request):
    """the composite (cell, x) key changes the work, never the result (SCCD_OPT_CELL_FACTOR_MILLI; "0": the grid switched off)"""
    ctx.set_option(sccd.OPT_CELL_FACTOR_MILLI, int(float(cell_factor) * 1000) or -1)
    request.addfinalizer(lambda: ctx.set_option(sccd.OPT_CELL_FACTOR_MILLI, 0))
    V0, V1, E, F = _scene("soup_dense")
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    want_vf, _, _ = orc.sort_and_sweep(vb, fb, nthreads=8)
    want_ee, _, _ = orc.sort_and_sweep(eb, nthreads=8)
    bp = sccd.BroadPhase(ctx)
    bp.build(sccd.DeviceAABBs(vb, ctx), sccd.DeviceAABBs(fb, ctx))
    assert np.array_equal(_sorted(bp.detect_overlaps()), want_vf)
    bp.build(sccd.DeviceAABBs(eb, ctx))
    assert np.array_equal(_sorted(bp.detect_overlaps()), want_ee)
    # boxes of very different sizes: the big ones are listed in many cells
    b = scenes.random_boxes(6000, seed=12, max_extent=0.01)
    big = scenes.random_boxes(40, seed=13, max_extent=0.8)
    big["element_id"] += 6000
    big["vertex_ids"] += 3 * 6000
    mix = np.concatenate([b, big])
    want, _, _ = orc.sort_and_sweep(mix, nthreads=8)
    bp.build(sccd.DeviceAABBs(mix, ctx))
    assert np.array_equal(_sorted(bp.detect_overlaps()), want)


def test_scan_build_path(sccd, ctx, orc, request):
    """SCCD_OPT_BUILD_SCAN: count -> device-wide prefix scan -> fill instead of the one-pass append
    (entries in box order).  Same pair sets, one and two lists."""
    ctx.set_option(sccd.OPT_BUILD_SCAN, 1)
    request.addfinalizer(lambda: ctx.set_option(sccd.OPT_BUILD_SCAN, 0))
    V0, V1, E, F = _scene("cloth_ball_10k")
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    bp = sccd.BroadPhase(ctx)
    bp.build(sccd.DeviceAABBs(eb, ctx))
    want, _, _ = orc.sort_and_sweep(eb, nthreads=8)
    assert np.array_equal(_sorted(bp.detect_overlaps()), want)
    bp.build(sccd.DeviceAABBs(vb, ctx), sccd.DeviceAABBs(fb, ctx))
    want, _, _ = orc.sort_and_sweep(vb, fb, nthreads=8)
    assert np.array_equal(_sorted(bp.detect_overlaps()), want)
    b = scenes.random_boxes(50_000, seed=3, max_extent=0.5)  # big boxes: the replication budget coarsens the grid
    want, _, _ = orc.sort_and_sweep(b[:4000], nthreads=8)
    bp.build(sccd.DeviceAABBs(b[:4000], ctx))
    assert np.array_equal(_sorted(bp.detect_overlaps()), want)
    ctx.set_option(sccd.OPT_BUILD_SCAN, 0)
    bp.build(sccd.DeviceAABBs(b[:4000], ctx))  # ... and the same through the append path
    assert np.array_equal(_sorted(bp.detect_overlaps()), want)


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_pair_set_independent_of_sort_axis(sccd, ctx, orc, axis):
    boxes = scenes.random_boxes(20000, seed=3, max_extent=0.06)
    want, _, _ = orc.sort_and_sweep(boxes, nthreads=8)
    ctx.set_option(sccd.OPT_SORT_AXIS, axis)
    try:
        bp = sccd.BroadPhase(ctx)
        bp.build(sccd.DeviceAABBs(boxes, ctx))
        got = bp.detect_overlaps()
    finally:
        ctx.set_option(sccd.OPT_SORT_AXIS, 0)
    assert np.array_equal(_sorted(got), want)


def test_broad_phase_edge_cases(sccd, ctx, orc):
    empty = np.zeros(0, sccd.AABB_DTYPE)
    one = scenes.random_boxes(1)
    bp = sccd.BroadPhase(ctx)
    with pytest.raises(RuntimeError):  # broad_phase.cu:123-126
        bp.detect_overlaps()
    for a, b in ((empty, None), (one, None), (empty, one), (one, empty)):
        bp.build(sccd.DeviceAABBs(a, ctx), sccd.DeviceAABBs(b, ctx) if b is not None else None)
        assert len(bp.detect_overlaps()) == 0 and bp.is_complete()
    # ties on the sort key, touching faces (inclusive test), identical boxes, shared vertices
    b = np.zeros(6, sccd.AABB_DTYPE)
    b["min"] = [[0, 0, 0], [1, 0, 0], [0, 0, 0], [2.5, 0, 0], [0, 1, 1], [0, 0, 0]]
    b["max"] = [[1, 1, 1], [2, 1, 1], [1, 1, 1], [3, 1, 1], [1, 2, 2], [1, 1, 1]]
    b["vertex_ids"] = [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11], [12, 13, 14], [0, 20, 21]]
    b["element_id"] = np.arange(6)
    want, _, _ = orc.sort_and_sweep(b)
    assert [0, 1] in want.tolist() and [0, 5] not in want.tolist()  # touching counts, shared vertex does not
    bp.build(sccd.DeviceAABBs(b, ctx))
    assert np.array_equal(_sorted(bp.detect_overlaps()), want)


@pytest.mark.parametrize("algo", [0, 2, 3])
def test_crowded_boxes_overflow_retry(sccd, ctx, orc, algo):
    # everything overlaps everything: exercises the crowded-block path of the candidate queue and
    # the overlap-buffer overflow -> exact-size rerun (broad_phase.cu:142-203)
    n = 1500
    b = scenes.random_boxes(n, seed=8, max_extent=0.9)
    want, _, _ = orc.sort_and_sweep(b, nthreads=8)
    assert len(want) > 0.25 * n * (n - 1) / 2
    ctx.set_option(sccd.OPT_OVERLAP_CAPACITY, 1000)
    ctx.set_option(sccd.OPT_SWEEP_ALGO, algo)
    try:
        bp = sccd.BroadPhase(ctx)
        bp.build(sccd.DeviceAABBs(b, ctx))
        got = bp.detect_overlaps()
    finally:
        ctx.set_option(sccd.OPT_OVERLAP_CAPACITY, 0)
        ctx.set_option(sccd.OPT_SWEEP_ALGO, 0)
    assert np.array_equal(_sorted(got), want)


def test_partial_cursor_covers_everything_once(sccd, ctx, orc):
    V0, V1, E, F = _scene("soup_400")
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    want, _, _ = orc.sort_and_sweep(vb, fb)
    ctx.set_option(sccd.OPT_MAX_OVERLAP_CUTOFF, 333)
    try:
        bp = sccd.BroadPhase(ctx)
        bp.build(sccd.DeviceAABBs(vb, ctx), sccd.DeviceAABBs(fb, ctx))
        calls = 0
        while not bp.is_complete():
            bp.detect_overlaps_partial()
            calls += 1
        assert calls >= -(-(len(vb) + len(fb)) // 333)  # one row per (box, overlapped cell)
        bp.build(sccd.DeviceAABBs(vb, ctx), sccd.DeviceAABBs(fb, ctx))
        got = bp.detect_overlaps()
    finally:
        ctx.set_option(sccd.OPT_MAX_OVERLAP_CUTOFF, 0)
    assert np.array_equal(_sorted(got), want)


def test_memory_limit_halves_the_swept_range(sccd, ctx, orc):
    """MemoryHandler behaviour (memory_handler.cpp:55-79): when the pairs of a chunk do not fit the
    budget, the swept box range is halved and the cursor advances by what was done."""
    V0, V1, E, F = _scene("soup_dense")
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    want, _, _ = orc.sort_and_sweep(eb, nthreads=8)
    assert len(want) > 100_000
    ctx.set_option(sccd.OPT_MEMORY_LIMIT_MB, 1)  # room for 65,536 pairs: several chunks are needed
    try:
        bp = sccd.BroadPhase(ctx)
        bp.build(sccd.DeviceAABBs(eb, ctx))
        chunks, parts = 0, []
        while not bp.is_complete():
            ptr, n = bp.detect_overlaps_partial()
            assert n <= 65536
            chunks += 1
        assert chunks >= 2
        bp.build(sccd.DeviceAABBs(eb, ctx))
        got = bp.detect_overlaps()
        toi = sccd.ccd(V0, V1, E, F, 0.0, -1, 1e-6, True, ctx=ctx)  # the driver loops over the chunks too (ccd.cu:55)
    finally:
        ctx.set_option(sccd.OPT_MEMORY_LIMIT_MB, 0)
    assert np.array_equal(_sorted(got), want)
    assert toi == sccd.ccd(V0, V1, E, F, 0.0, -1, 1e-6, True, ctx=ctx)


def test_overflow_behind_a_speculative_build_resweeps_the_settled_rows(sccd, orc):
    """A step whose pair list outgrows its buffer right after a SPECULATIVE build (lists sized by the previous build's counts plus
    a margin): the re-sweep must cover the settled rows, not the padded bounds -- rows behind a list's real end hold whatever an
    earlier build left there.  On a context of its own (the suite's shared one has large buffers by now): the first call under a
    1 MB memory limit leaves 65,536-pair buffers, the second call without the limit builds speculatively and overflows them.
    Round 4: a memory fault in the vertex-face narrow kernel when test_memory_limit_halves_the_swept_range ran first in a process."""
    V0, V1, E, F = _scene("soup_dense")
    want, n_vf, n_ee = orc.ccd(V0, V1, E, F, 0.0, -1, 1e-6, True)
    own = sccd.Context(0)
    try:
        mesh = sccd.Mesh(V0, V1, E, F, ctx=own)
        own.set_option(sccd.OPT_MEMORY_LIMIT_MB, 1)
        toi, st = sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True, want_stats=True)
        assert toi == want and st["n_vf_pairs"] == n_vf and st["n_ee_pairs"] == n_ee
        own.set_option(sccd.OPT_MEMORY_LIMIT_MB, 0)
        for _ in range(3):  # (the first of these overflows behind a speculative build; the others find room)
            toi, st = sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True, want_stats=True)
            assert toi == want and st["n_vf_pairs"] == n_vf and st["n_ee_pairs"] == n_ee
        # (under the lab switches that turn the speculative build off the scene is still checked, the path is not met: tools/jobs/env_matrix.sh)
        if os.environ.get("SCCD_SPECULATE") != "0":
            assert own.get_option(sccd.OPT_SPEC_HITS) > 0
    finally:
        own.close()


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("two_lists", [False, True])
def test_sharded_sweeps_partition_the_pair_set(sccd, ctx, orc, world, two_lists):
    """Every rank sweeps its window of grid cells (or, on a grid too small to deal out, its slice
    of the rows): the union over the ranks is the reference's pair set, with no pair twice."""
    V0, V1, E, F = _scene("cloth_ball_10k")
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    if two_lists:
        want, _, _ = orc.sort_and_sweep(vb, fb, nthreads=8)
    else:
        want, _, _ = orc.sort_and_sweep(eb, nthreads=8)
    parts = []
    try:
        for r in range(world):
            ctx.set_option(sccd.OPT_SHARD_COUNT, world)
            ctx.set_option(sccd.OPT_SHARD_RANK, r)
            bp = sccd.BroadPhase(ctx)
            if two_lists:
                bp.build(sccd.DeviceAABBs(vb, ctx), sccd.DeviceAABBs(fb, ctx))
            else:
                bp.build(sccd.DeviceAABBs(eb, ctx))
            parts.append(bp.detect_overlaps().reshape(-1, 2))
    finally:
        ctx.set_option(sccd.OPT_SHARD_COUNT, 1)
        ctx.set_option(sccd.OPT_SHARD_RANK, 0)
    assert sum(len(p) > 0 for p in parts) >= 2
    assert np.array_equal(_sorted(np.concatenate(parts)), want)  # disjoint and complete


def test_sharded_sweep_on_a_single_cell_grid_splits_rows(sccd, ctx, orc, request):
    """SCCD_OPT_CELL_FACTOR_MILLI < 0 switches the grid off: the shards fall back to slices of the rows."""
    ctx.set_option(sccd.OPT_CELL_FACTOR_MILLI, -1)
    request.addfinalizer(lambda: ctx.set_option(sccd.OPT_CELL_FACTOR_MILLI, 0))
    b = scenes.random_boxes(20_000, seed=5, max_extent=0.05)
    want, _, _ = orc.sort_and_sweep(b, nthreads=8)
    parts = []
    try:
        for r in range(4):
            ctx.set_option(sccd.OPT_SHARD_COUNT, 4)
            ctx.set_option(sccd.OPT_SHARD_RANK, r)
            bp = sccd.BroadPhase(ctx)
            bp.build(sccd.DeviceAABBs(b, ctx))
            parts.append(bp.detect_overlaps().reshape(-1, 2))
    finally:
        ctx.set_option(sccd.OPT_SHARD_COUNT, 1)
        ctx.set_option(sccd.OPT_SHARD_RANK, 0)
    assert all(len(p) > 0 for p in parts)
    assert np.array_equal(_sorted(np.concatenate(parts)), want)


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_cpu_entry_point_sort_and_sweep(sccd, ctx, orc, axis):
    """scalable_ccd::sort_and_sweep (sort_and_sweep.hpp:28-42) served by the device path: the pair
    set of the CPU code along any sort axis, and the arg-max-variance axis it hands back."""
    V0, V1, E, F = _scene("cloth_ball_small")
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    want, want_axis, _ = orc.sort_and_sweep(eb, sort_axis=axis)
    got, got_axis = sccd.sort_and_sweep(eb, sort_axis=axis, ctx=ctx)
    assert np.array_equal(_sorted(got), want) and got_axis == want_axis
    want, want_axis, _ = orc.sort_and_sweep(vb, fb, sort_axis=axis)
    got, got_axis = sccd.sort_and_sweep(vb, fb, sort_axis=axis, ctx=ctx)
    assert np.array_equal(_sorted(got), want) and got_axis == want_axis
    # a thin slab: the variance rule must pick the long axis whatever axis was swept
    b = scenes.random_boxes(5000, seed=9, max_extent=0.02)
    b["min"][:, 0] *= 0.01
    b["max"][:, 0] *= 0.01
    b["min"][:, 2] *= 7.0
    b["max"][:, 2] *= 7.0
    want, want_axis, _ = orc.sort_and_sweep(b, sort_axis=axis)
    got, got_axis = sccd.sort_and_sweep(b, sort_axis=axis, ctx=ctx)
    assert np.array_equal(_sorted(got), want) and got_axis == want_axis == 2
    assert ctx.get_option(sccd.OPT_SORT_AXIS) == 0  # the context's own setting is restored
    empty, ax = sccd.sort_and_sweep(b[:0], sort_axis=axis, ctx=ctx)
    assert len(empty) == 0 and ax == axis
    # the reference's two-step form (sort_and_sweep.cpp:126-141,143-195,221-240): sort_along_axis, then sweep<> on the sorted
    # boxes -- two lists merged, the first list's ids flipped to -id - 1
    se = sccd.sort_along_axis(axis, eb)
    assert np.all(np.diff(se["min"][:, axis]) >= 0)
    want, want_axis, _ = orc.sort_and_sweep(eb, sort_axis=axis)
    got, got_axis = sccd.sweep(se, axis, two_lists=False, ctx=ctx)
    assert np.array_equal(_sorted(got), want) and got_axis == want_axis
    fa = sccd.sort_along_axis(axis, vb).copy()
    fa["element_id"] = -fa["element_id"] - 1
    merged = np.concatenate([fa, sccd.sort_along_axis(axis, fb)])
    merged = merged[np.argsort(merged["min"][:, axis], kind="stable")]
    want, want_axis, _ = orc.sort_and_sweep(vb, fb, sort_axis=axis)
    got, got_axis = sccd.sweep(merged, axis, two_lists=True, ctx=ctx)
    assert np.array_equal(_sorted(got), want) and got_axis == want_axis


@pytest.mark.parametrize("two_lists", [False, True])
def test_speculative_builds_follow_the_scene_and_survive_a_wrong_guess(sccd, ctx, orc, two_lists):
    """The second build of the same number of boxes on a BroadPhase is enqueued on the FIRST build's entry counts (sort, records
    and sweep right behind the fill, real counts read on the device: csrc/api.hip bp_build).  Boxes that barely change keep the
    guess; boxes that grow past its margin, or change the grid's key width, must fall back to the exact build -- the pair set
    is the oracle's every time."""
    n = 30_000
    bp = sccd.BroadPhase(ctx)

    def scene(seed, ext, jitter=0.0):
        a = scenes.random_boxes(n, seed=seed, max_extent=ext)
        if jitter:
            rng = np.random.default_rng(seed + 1)
            d = rng.uniform(-jitter, jitter, (n, 3))
            a["min"] += d
            a["max"] += d
        if not two_lists:
            return a, None
        b = scenes.random_boxes(n // 2, seed=seed + 100, max_extent=ext)
        return a, b

    steps = [(5, 0.02, 0.0), (5, 0.02, 1e-4), (5, 0.02, 2e-4),  # the guess holds (a moving scene)
             (5, 0.05, 0.0),                                     # many more entries: past the margin
             (6, 0.004, 0.0), (6, 0.004, 1e-5),                  # far fewer, another grid
             (5, 0.02, 0.0)]
    for seed, ext, jitter in steps:
        a, b = scene(seed, ext, jitter)
        bp.build(sccd.DeviceAABBs(a, ctx), sccd.DeviceAABBs(b, ctx) if b is not None else None)
        got = _sorted(bp.detect_overlaps())
        want = orc.sort_and_sweep(a, b)[0] if b is not None else orc.sort_and_sweep(a)[0]
        assert np.array_equal(got, want), (seed, ext, jitter, len(got), len(want))
        assert bp.is_complete()
    # options changed BETWEEN a (speculative) build and its sweep to ones a speculative sweep does not serve: the plain
    # sweep, rows in chunks -- the build is settled first (csrc/api.hip bp_detect_partial)
    a, b = scene(5, 0.02, 3e-4)
    want = orc.sort_and_sweep(a, b)[0] if b is not None else orc.sort_and_sweep(a)[0]
    da, db = sccd.DeviceAABBs(a, ctx), (sccd.DeviceAABBs(b, ctx) if b is not None else None)
    for option, value in ((sccd.OPT_SWEEP_ALGO, 1), (sccd.OPT_MAX_OVERLAP_CUTOFF, 7000)):
        bp.build(da, db)
        ctx.set_option(option, value)
        try:
            assert np.array_equal(_sorted(bp.detect_overlaps()), want), option
        finally:
            ctx.set_option(option, 0)


def test_random_100k_matches_golden_hash(sccd, ctx):
    G = json.load(open(GOLDEN))["random_100k"]
    b = scenes.random_boxes(100_000, seed=42, max_extent=0.027)
    bp = sccd.BroadPhase(ctx)
    bp.build(sccd.DeviceAABBs(b, ctx))
    got = _sorted(bp.detect_overlaps())
    assert len(got) == G["n"]
    assert hashlib.sha256(got.tobytes()).hexdigest() == G["sha256"]


# ---- narrow phase -----------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["cloth_ball_small", "soup_400", "cloth_ball_10k"])
@pytest.mark.parametrize("arith", [0, 1])
@pytest.mark.parametrize("algo", [0, 1])
def test_toi_matches_oracle(sccd, ctx, orc, name, arith, algo):
    V0, V1, E, F = _scene(name)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    vf, _, _ = orc.sort_and_sweep(vb, fb, nthreads=8)
    ee, _, _ = orc.sort_and_sweep(eb, nthreads=8)
    want_vf, _ = orc.narrow_phase_mt(V0, V1, E, F, vf, True, arith=arith, nthreads=8)
    want_ee, _ = orc.narrow_phase_mt(V0, V1, E, F, ee, False, arith=arith, toi=want_vf, nthreads=8)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    ctx.set_option(sccd.OPT_ARITH, arith)
    ctx.set_option(sccd.OPT_NARROW_ALGO, algo)
    try:
        got_vf = sccd.narrow_phase(mesh, vf, True)
        got_ee = sccd.narrow_phase(mesh, ee, False, toi=got_vf)
    finally:
        ctx.set_option(sccd.OPT_ARITH, sccd.ARITH_DEFAULT)
        ctx.set_option(sccd.OPT_NARROW_ALGO, 0)
    assert abs(got_vf - want_vf) <= TOI_TOL and abs(got_ee - want_ee) <= TOI_TOL
    assert got_vf == want_vf and got_ee == want_ee  # bit-equal under the same arithmetic contract


@pytest.mark.parametrize("name", ["cloth_ball_small", "soup_400", "cloth_ball_10k"])
@pytest.mark.parametrize("max_iter", [-1, 5000, 50])
@pytest.mark.parametrize("tol", [1e-6, 1e-9])
def test_narrow_phase_on_a_callers_list_with_the_cull_in_front(sccd, ctx, orc, name, max_iter, tol):
    # sccd_narrow_phase culls the caller's list first (from 100,000 pairs on; forced here): the same TOI as the whole list gives in the
    # oracle, with and without a check limit, from 1 and from a bound the first pass found, with a minimum separation
    V0, V1, E, F = _scene(name)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    vf, _, _ = orc.sort_and_sweep(vb, fb, nthreads=8)
    ee, _, _ = orc.sort_and_sweep(eb, nthreads=8)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    ctx.set_option(sccd.OPT_CULL, 2)
    try:
        for ms in ((0.0, 1e-3) if tol == 1e-6 else (0.0,)): # (a minimum separation at 1e-9, unlimited: half a minute of oracle)
            if max_iter < 0: # (no limit: the order of the traversal decides nothing -- the threaded depth-first oracle)
                want_vf = orc.narrow_phase_mt(V0, V1, E, F, vf, True, ms, max_iter, tol, True, nthreads=8)[0]
                want_ee = orc.narrow_phase_mt(V0, V1, E, F, ee, False, ms, max_iter, tol, True, toi=want_vf, nthreads=8)[0]
            else:
                want_vf = orc.narrow_phase(V0, V1, E, F, vf, True, ms, max_iter, tol, True)[0]
                want_ee = orc.narrow_phase(V0, V1, E, F, ee, False, ms, max_iter, tol, True, toi=want_vf)[0]
            got_vf = sccd.narrow_phase(mesh, vf, True, max_iter, tol, ms, True)
            got_ee = sccd.narrow_phase(mesh, ee, False, max_iter, tol, ms, True, toi=got_vf)
            assert got_vf == want_vf and got_ee == want_ee, (ms, got_vf, want_vf, got_ee, want_ee)
    finally:
        ctx.set_option(sccd.OPT_CULL, 1)


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (1e-6, 0.0), (1e4, -3.7e6), (3.0, 1.0e9)])
def test_pair_set_under_translation_and_scale(sccd, ctx, orc, scale, offset):
    """The composite key quantises coordinates relative to the scene bounds; tiny scenes, huge
    offsets (where neighbouring doubles are 2e-7 apart) and negative coordinates must not change
    the pair set.  The oracle runs on the SAME transformed boxes."""
    b = scenes.random_boxes(30_000, seed=13, max_extent=0.04)
    b["min"] = b["min"] * scale + offset
    b["max"] = b["max"] * scale + offset
    assert (b["min"] <= b["max"]).all()
    want, _, _ = orc.sort_and_sweep(b, nthreads=8)
    assert len(want) > 1000
    bp = sccd.BroadPhase(ctx)
    bp.build(sccd.DeviceAABBs(b, ctx))
    assert np.array_equal(_sorted(bp.detect_overlaps()), want)
    # two lists built from the same boxes, every second box flipped onto one plane (zero extent)
    a2, b2 = b[::2].copy(), b[1::2].copy()
    a2["max"][:, 2] = a2["min"][:, 2]
    b2["vertex_ids"] += 10_000_000
    want, _, _ = orc.sort_and_sweep(a2, b2, nthreads=8)
    bp.build(sccd.DeviceAABBs(a2, ctx), sccd.DeviceAABBs(b2, ctx))
    assert np.array_equal(_sorted(bp.detect_overlaps()), want)


@pytest.mark.parametrize("world", [1, 5])
def test_many_small_boxes_use_thousands_of_cells(sccd, ctx, orc, world):
    """A volumetric scene of small boxes: the grid wants ~15,000 cells (more than the 1024 the first
    version allowed).  One and two lists, unsharded and as 5 cell windows."""
    b = scenes.random_boxes(200_000, seed=21, max_extent=0.004)
    a2 = scenes.random_boxes(60_000, seed=22, max_extent=0.006)
    a2["vertex_ids"] += 10_000_000
    want1, _, _ = orc.sort_and_sweep(b, nthreads=8)
    want2, _, _ = orc.sort_and_sweep(a2, b, nthreads=8)
    assert len(want1) > 1000 and len(want2) > 1000
    got1, got2 = [], []
    try:
        for r in range(world):
            ctx.set_option(sccd.OPT_SHARD_COUNT, world)
            ctx.set_option(sccd.OPT_SHARD_RANK, r)
            bp = sccd.BroadPhase(ctx)
            bp.build(sccd.DeviceAABBs(b, ctx))
            got1.append(bp.detect_overlaps().reshape(-1, 2))
            bp.build(sccd.DeviceAABBs(a2, ctx), sccd.DeviceAABBs(b, ctx))
            got2.append(bp.detect_overlaps().reshape(-1, 2))
    finally:
        ctx.set_option(sccd.OPT_SHARD_COUNT, 1)
        ctx.set_option(sccd.OPT_SHARD_RANK, 0)
    assert np.array_equal(_sorted(np.concatenate(got1)), want1)
    assert np.array_equal(_sorted(np.concatenate(got2)), want2)


def test_degenerate_box_lists(sccd, ctx, orc):
    """Point boxes, thousands of identical boxes, one box covering everything."""
    n = 3000
    b = np.zeros(n, sccd.AABB_DTYPE)
    rng = np.random.default_rng(4)
    p = rng.random((n, 3))
    b["min"] = p
    b["max"] = p  # zero extent on every axis: the mean extent the grid is sized from is 0
    b["vertex_ids"] = np.arange(3 * n).reshape(n, 3)
    b["element_id"] = np.arange(n)
    b["min"][:500] = 0.5
    b["max"][:500] = 0.5  # 500 coincident points: 124,750 pairs from one spot
    b["min"][-1] = -1.0
    b["max"][-1] = 2.0  # and one box that contains all the others
    want, _, _ = orc.sort_and_sweep(b, nthreads=8)
    assert len(want) == 500 * 499 // 2 + (n - 1)
    bp = sccd.BroadPhase(ctx)
    bp.build(sccd.DeviceAABBs(b, ctx))
    assert np.array_equal(_sorted(bp.detect_overlaps()), want)


def test_resting_and_grazing_contacts(sccd, ctx, orc):
    """Queries that start in contact (TOI 0 with allow_zero_toi, > 0 without), parallel sliding,
    and a mesh that does not move at all."""
    V0, F = scenes.cloth_grid(6)  # (coplanar sliding is Tight-Inclusion's worst case: keep the oracle's work small)
    V0 = V0.copy()
    E = scenes.edges_from_faces(F)
    # a second, identical sheet lying exactly on the first one, sliding sideways
    Va0 = np.concatenate([V0, V0 + [0.013, 0.007, 0.0]])
    Va1 = np.concatenate([V0, V0 + [0.113, 0.007, 0.0]])
    Fa = np.concatenate([F, F + len(V0)]).astype(np.int32)
    Ea = np.concatenate([E, E + len(V0)]).astype(np.int32)
    for allow_zero in (True, False):
        want, _, _ = orc.ccd(Va0, Va1, Ea, Fa, 0.0, -1, 1e-6, allow_zero, nthreads=8)
        got = sccd.ccd(Va0, Va1, Ea, Fa, 0.0, -1, 1e-6, allow_zero, ctx=ctx)
        assert got == want
        if allow_zero:
            assert got == 0.0
    # nothing moves: every query is static (infinite time tolerance), still the oracle's answer
    want, _, _ = orc.ccd(Va0, Va0, Ea, Fa, 0.0, -1, 1e-6, True, nthreads=8)
    assert sccd.ccd(Va0, Va0, Ea, Fa, 0.0, -1, 1e-6, True, ctx=ctx) == want
    # sheets 1e-3 apart moving in parallel: minimum separation decides
    Vb0 = np.concatenate([V0, V0 + [0.013, 0.007, 1e-3]])
    Vb1 = np.concatenate([V0 + [0.2, 0, 0], V0 + [0.213, 0.007, 1e-3]])
    for ms in (0.0, 5e-4, 2e-3):
        want, _, _ = orc.ccd(Vb0, Vb1, Ea, Fa, ms, -1, 1e-6, True, nthreads=8)
        assert sccd.ccd(Vb0, Vb1, Ea, Fa, ms, -1, 1e-6, True, ctx=ctx) == want


@pytest.mark.parametrize("ms", [0.0, 1e-3])
@pytest.mark.parametrize("allow_zero", [True, False])
def test_narrow_phase_parameters(sccd, ctx, orc, ms, allow_zero):
    V0, V1, E, F = _scene("cloth_ball_small")
    vb, eb, fb = orc.build_boxes(V0, V1, E, F, ms)
    ee, _, _ = orc.sort_and_sweep(eb)
    want, _ = orc.narrow_phase_mt(V0, V1, E, F, ee, False, ms=ms, allow_zero_toi=allow_zero, nthreads=8)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    assert sccd.narrow_phase(mesh, ee, False, ms=ms, allow_zero_toi=allow_zero) == want


def test_narrow_phase_known_answers(sccd, ctx):
    V0 = np.array([[0.25, 0.25, 1.0], [0, 0, 0], [1, 0, 0], [0, 1, 0]], float)
    V1 = V0.copy()
    V1[0, 2] = -1.0
    E = np.array([[1, 2], [2, 3], [1, 3]], np.int32)
    F = np.array([[1, 2, 3]], np.int32)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    t = sccd.narrow_phase(mesh, [[0, 0]], True)
    assert 0.5 - 1e-5 <= t <= 0.5
    assert sccd.narrow_phase(mesh, [[0, 0]], True, toi=0.25) == 0.25  # in/out bound (narrow_phase.cu:124)
    assert sccd.narrow_phase(mesh, [[0, 0]], True, toi=0.0) == 0.0    # loop guard toi > 0 (:136)
    assert sccd.narrow_phase(mesh, np.zeros((0, 2), np.int32), True) == 1.0
    with pytest.raises(RuntimeError):
        sccd.narrow_phase(mesh, [[0, 0]], True, toi=-1.0)  # assert(toi >= 0), narrow_phase.cu:126
    with pytest.raises(RuntimeError):
        sccd.narrow_phase(mesh, [[7, 0]], True)
    V1[0, 2] = 0.5  # never reaches the triangle
    mesh.update_vertices(V0, V1)
    assert sccd.narrow_phase(mesh, [[0, 0]], True) == 1.0


def test_narrow_lists_around_the_deal_boundaries(sccd, ctx, orc):
    """np_walk_k deals a list in batches of 21 queries, as one list (fewer than 32 batches or fewer than eight blocks)
    or as eight interleaved sub-lists of segments of 2^k batches with the rest on sub-list 0: list lengths on and around
    every one of those boundaries, each against the oracle, each with per-query output (every query must be checked
    exactly once: a query dealt twice or not at all shows in the collision records)."""
    V0, V1, E, F = _scene("cloth_ball_10k")
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    ee, _, _ = orc.sort_and_sweep(eb, nthreads=8)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    lengths = set()
    for nb in (1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025):  # batches
        for d in (-1, 0, 1):
            lengths.add(21 * nb + d)
    lengths |= {1, 20, 22, 41, 43, 21 * 8 * 32 * 4 + 5, len(ee)}
    for n in sorted(x for x in lengths if 0 < x <= len(ee)):
        pairs = ee[:n]
        want = orc.narrow_phase_mt(V0, V1, E, F, pairs, False, nthreads=8)[0]
        assert sccd.narrow_phase(mesh, pairs, False) == want, n
    for n in (21 * 32 - 1, 21 * 32, 21 * 256 + 1, min(len(ee), 21 * 1024 + 22)):
        pairs = ee[:n]
        _, want_pq, _ = orc.narrow_phase(V0, V1, E, F, pairs, False, per_query=True)
        _, col = sccd.narrow_phase(mesh, pairs, False, want_collisions=True)
        hits = want_pq < 1
        assert len(col) == int(hits.sum()), n
        assert np.array_equal(col["toi"], want_pq[hits]), n


@pytest.mark.parametrize("algo", [0, 1])  # work-queue kernel (per-lane bound, atomicMin per query), level order
@pytest.mark.parametrize("is_vf", [True, False])
def test_per_query_collisions(sccd, ctx, orc, algo, is_vf):
    """SCALABLE_CCD_TOI_PER_QUERY: (aid, bid, toi) of every query with toi < 1; a query is pruned by
    its own earliest impact only (root_finder.cu:297), so every value is the oracle's, bit for bit."""
    V0, V1, E, F = scenes.triangle_soup(300, seed=6, size=0.12, motion=0.35)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    pairs, _, _ = orc.sort_and_sweep(vb, fb) if is_vf else orc.sort_and_sweep(eb)
    want_t, want_pq, _ = orc.narrow_phase(V0, V1, E, F, pairs, is_vf, per_query=True)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    try:
        ctx.set_option(sccd.OPT_NARROW_ALGO, algo)
        t, col = sccd.narrow_phase(mesh, pairs, is_vf, want_collisions=True)
    finally:
        ctx.set_option(sccd.OPT_NARROW_ALGO, 0)
    assert t == want_t
    hits = want_pq < 1
    assert hits.sum() > 10 and len(col) == hits.sum()
    got = {(int(a), int(b)): float(x) for a, b, x in zip(col["aid"], col["bid"], col["toi"])}
    for (a, b), x in zip(pairs[hits], want_pq[hits]):
        assert got[(int(a), int(b))] == x
    assert all(t <= x for x in got.values())  # tests/test_narrow_phase.cu:60-62


# ---- end to end ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["cloth_ball_10k", "cloth_ball_10k_ms", "soup_400", "cloth_ball_small"])
@pytest.mark.parametrize("arith", [0, 1])
def test_ccd_matches_golden(sccd, ctx, name, arith):
    G = json.load(open(GOLDEN))[name]
    V0, V1, E, F = _scene(name.replace("_ms", ""))
    ms = 1e-3 if name.endswith("_ms") else 0.0
    ctx.set_option(sccd.OPT_ARITH, arith)
    try:
        toi = sccd.ccd(V0, V1, E, F, ms, -1, 1e-6, True, ctx=ctx)
        mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
        toi2, st = sccd.ccd_mesh(mesh, ms, want_stats=True)
    finally:
        ctx.set_option(sccd.OPT_ARITH, sccd.ARITH_DEFAULT)
    want = float.fromhex(G["toi_fma" if arith else "toi_strict"])
    assert toi == want and toi2 == want
    assert st["n_vf_pairs"] == G["n_vf"] and st["n_ee_pairs"] == G["n_ee"]


@pytest.mark.parametrize("arith", [0, 1])
def test_golden_case_that_tells_the_arithmetic_contracts_apart(sccd, ctx, arith):
    """Every other golden case has toi_fma == toi_strict: a build whose fused path were silently off would pass them.  In
    this one (tests/golden/make_contract_case.py) the minimum separation sits where the two contracts decide differently."""
    G = json.load(open(GOLDEN))["contract_split"]
    V0 = np.array([[float.fromhex(x) for x in r] for r in G["V0"]])
    V1 = np.array([[float.fromhex(x) for x in r] for r in G["V1"]])
    E, F, ms = np.array(G["E"], np.int32), np.array(G["F"], np.int32), float.fromhex(G["ms"])
    assert G["toi_strict"] != G["toi_fma"]
    ctx.set_option(sccd.OPT_ARITH, arith)
    try:
        toi = sccd.ccd(V0, V1, E, F, ms, -1, 1e-6, True, ctx=ctx)
        ctx.set_option(sccd.OPT_NARROW_ALGO, 1)  # ... and the level-order kernels
        toi_level = sccd.ccd(V0, V1, E, F, ms, -1, 1e-6, True, ctx=ctx)
    finally:
        ctx.set_option(sccd.OPT_ARITH, sccd.ARITH_DEFAULT)
        ctx.set_option(sccd.OPT_NARROW_ALGO, 0)
    want = float.fromhex(G["toi_fma" if arith else "toi_strict"])
    assert toi == want and toi_level == want


@pytest.mark.parametrize("seed", list(range(12)))
def test_randomised_scenes_against_the_oracle(sccd, ctx, orc, seed):
    """Triangle soups and cloth-ball scenes with seeded random size, density, motion, minimum
    separation, zero-TOI policy and arithmetic contract: pair sets and TOI against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    if seed % 3 == 2:
        V0, V1, E, F = scenes.cloth_ball(int(rng.integers(8, 40)), 1, seed=int(rng.integers(1, 10**6)))
    else:
        V0, V1, E, F = scenes.triangle_soup(int(rng.integers(50, 900)), seed=int(rng.integers(1, 10**6)),
                                            size=float(rng.uniform(0.03, 0.2)), motion=float(rng.uniform(0.0, 0.4)))
    ms = float(rng.choice([0.0, 0.0, 1e-4, 3e-3]))
    allow_zero = bool(rng.integers(0, 2))
    arith = int(rng.integers(0, 2))
    vb, eb, fb = orc.build_boxes(V0, V1, E, F, ms)
    want_vf, _, _ = orc.sort_and_sweep(vb, fb, nthreads=8)
    want_ee, _, _ = orc.sort_and_sweep(eb, nthreads=8)
    want, _, _ = orc.ccd(V0, V1, E, F, ms, -1, 1e-6, allow_zero, arith=arith, nthreads=8)
    try:
        ctx.set_option(sccd.OPT_ARITH, arith)
        mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
        dv, de, df = sccd.DeviceAABBs.from_mesh(mesh, ms)
        bp = sccd.BroadPhase(ctx)
        bp.build(dv, df)
        assert np.array_equal(_sorted(bp.detect_overlaps()), want_vf)
        bp.build(de)
        assert np.array_equal(_sorted(bp.detect_overlaps()), want_ee)
        assert sccd.ccd_mesh(mesh, ms, -1, 1e-6, allow_zero) == want
    finally:
        ctx.set_option(sccd.OPT_ARITH, sccd.ARITH_DEFAULT)


def test_odd_meshes(sccd, ctx, orc):
    """One triangle, no faces, no edges, nothing at all, an absurdly fast vertex, and one context
    reused for scenes of very different sizes."""
    V0 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0.2, 0.2, 1.0]], float)
    V1 = V0.copy()
    V1[3, 2] = -1
    F = np.array([[0, 1, 2]], np.int32)
    E = np.array([[0, 1], [1, 2], [0, 2]], np.int32)
    noE, noF = np.zeros((0, 2), np.int32), np.zeros((0, 3), np.int32)
    for e, f in ((E, F), (E, noF), (noE, F), (noE, noF)):
        assert sccd.ccd(V0, V1, e, f, ctx=ctx) == orc.ccd(V0, V1, e, f)[0]
    assert sccd.ccd(np.zeros((0, 3)), np.zeros((0, 3)), noE, noF, ctx=ctx) == 1.0
    Vfast = V0.copy()
    Vfast[3] = [1e8, -1e8, -1e8]
    assert sccd.ccd(V0, Vfast, E, F, ctx=ctx) == orc.ccd(V0, Vfast, E, F)[0]
    for n in (3, 40, 7, 90):
        scene = scenes.cloth_ball(n, 1, seed=n)
        assert sccd.ccd(*scene, ctx=ctx) == orc.ccd(*scene, nthreads=8)[0]


def test_ccd_argument_errors(sccd, ctx):
    V0, V1, E, F = _scene("cloth_ball_small")
    with pytest.raises(RuntimeError):
        sccd.ccd(V0, V1[:-1], E, F, ctx=ctx)  # ccd.cu:94-95
    bad = F.copy()
    bad[0, 0] = len(V0)
    with pytest.raises(RuntimeError):
        sccd.ccd(V0, V1, E, bad, ctx=ctx)
    # vertex indices are validated on the device while E and F are packed (csrc/boxes.hip pack_*_k): every entry that
    # takes index matrices refuses a bad one -- first, last, negative -- and the context works on afterwards
    want = sccd.ccd(V0, V1, E, F, ctx=ctx)
    for where, value in ((0, len(V0)), (-1, len(V0) + 7), (len(F) // 2, -1)):
        badF, badE = F.copy(), E.copy()
        badF[where, 2] = value
        badE[where % len(E), 1] = value
        for args in ((E, badF), (badE, F)):
            with pytest.raises(RuntimeError, match="out of range"):
                sccd.ccd(V0, V1, *args, ctx=ctx)
            with pytest.raises(RuntimeError, match="out of range"):
                sccd.Mesh(V0, V1, *args, ctx=ctx)
            with pytest.raises(RuntimeError, match="out of range"):
                sccd.ipc_ccd_strategy(V0, V1, *args, 0.0, -1, 1e-6, ctx=ctx)
            with pytest.raises(RuntimeError, match="out of range"):
                sccd.ccd(V0, V1, *args, ctx=ctx, want_collisions=True)
        vb = sccd.build_vertex_boxes(V0, V1, 0.0, ctx)
        with pytest.raises(RuntimeError, match="out of range"):
            sccd.build_face_boxes(vb, badF, ctx)
        with pytest.raises(RuntimeError, match="out of range"):
            sccd.build_edge_boxes(vb, badE, ctx)
        assert sccd.ccd(V0, V1, E, F, ctx=ctx) == want
    # nothing moves: the reference's EE tol[1] quirk (root_finder.cu:82-85) makes coplanar static
    # edges "collide" at t = 0; the restatement and the HIP path reproduce that
    from orc import ccd as oracle_ccd

    assert sccd.ccd(V0, V0, E, F, ctx=ctx) == oracle_ccd(V0, V0, E, F)[0] == 0.0


def test_ipc_ccd_strategy(sccd, ctx, orc):
    V0, V1, E, F = _scene("cloth_ball_small")
    t = sccd.ipc_ccd_strategy(V0, V1, E, F, 0.0, -1, 1e-6, ctx=ctx)
    want, _, _ = orc.ccd(V0, V1, E, F)
    assert t == want  # toi >= 1e-6: no conservative re-run (ipc_ccd_strategy.cu:72)
    # resting contact at t = 0 forces the re-run: ms = 0, no zero toi, result scaled by 0.8 (:72-91)
    V0 = np.array([[0.25, 0.25, 0.0], [0, 0, 0], [1, 0, 0], [0, 1, 0]], float)
    V1 = V0.copy()
    V1[0, 2] = -1.0
    E = np.array([[1, 2], [2, 3], [1, 3]], np.int32)
    F = np.array([[1, 2, 3]], np.int32)
    t = sccd.ipc_ccd_strategy(V0, V1, E, F, 0.0, -1, 1e-6, ctx=ctx)
    assert 0.0 <= t <= 1e-5
    want, reran = orc.ipc_ccd_strategy(V0, V1, E, F, 0.0, -1, 1e-6, want_branches=True)
    assert reran and t == want  # the oracle twin took the re-run branch too: same value, bit for bit


@pytest.mark.parametrize("case", ["resting_mesh", "ms_contact", "late_hit"])
def test_ipc_ccd_strategy_matches_the_oracle_twin(sccd, ctx, orc, case):
    """ipc_ccd_strategy.cu:43-91 against orc.ipc_ccd_strategy, bit for bit, on scenes that take the conservative
    re-run in the vertex-face pass, in the edge-edge pass, in both, or in neither."""
    if case == "resting_mesh":  # a small cloth lying ON a static one, pushed through it: toi = 0 in both passes
        V0a, Fa = scenes.cloth_grid(6)
        Ea = scenes.edges_from_faces(Fa)
        nA = len(V0a)  # second copy of the whole sheet, shifted in x/y by a fraction of a cell
        V0 = np.vstack([V0a, V0a + np.array([0.013, 0.007, 0.0])])
        V1 = V0.copy()
        V1[nA:, 2] -= 0.05
        E = np.vstack([Ea, Ea + nA]).astype(np.int32)
        F = np.vstack([Fa, Fa + nA]).astype(np.int32)
        ms, mi = 0.0, -1
    elif case == "ms_contact":  # closer than the minimum separation at t = 0: re-run WITHOUT ms finds the real hit
        V0 = np.array([[0.3, 0.3, 1e-4], [0, 0, 0], [1, 0, 0], [0, 1, 0]], float)
        V1 = V0.copy()
        V1[0, 2] = -0.5
        E = np.array([[1, 2], [2, 3], [1, 3]], np.int32)
        F = np.array([[1, 2, 3]], np.int32)
        ms, mi = 1e-3, 10_000_000
    else:
        V0, V1, E, F = scenes.triangle_soup(300, seed=21)
        ms, mi = 0.0, 10_000_000
    want, reran = orc.ipc_ccd_strategy(V0, V1, E, F, ms, mi, 1e-6, want_branches=True)
    got = sccd.ipc_ccd_strategy(V0, V1, E, F, ms, mi, 1e-6, ctx=ctx)
    assert got == want, (case, got, want, reran)
    try:  # the cull in front of both runs of a chunk (the list kept under ms serves the re-run without it: drivers.hip ipc_pass)
        ctx.set_option(sccd.OPT_CULL, 2)
        assert sccd.ipc_ccd_strategy(V0, V1, E, F, ms, mi, 1e-6, ctx=ctx) == want, (case, "cull forced")
    finally:
        ctx.set_option(sccd.OPT_CULL, 1)
    if case != "late_hit":
        assert reran, "the scene is meant to take the conservative re-run branch"


def test_check_limit_that_no_query_reaches_runs_on_the_fast_kernel(sccd, ctx, orc):
    """A limit no query comes near (the IPC Toolkit passes 1e7) changes nothing, and the library proves that instead of
    paying for the reference's level order: the fast kernel runs without the limit, then ONE query -- the one that holds
    the earliest impact -- is bisected alone in level order with the limit on the host (csrc/ti_census.cpp).  Same TOI,
    a check count like max_iter = -1 (level order needs several times as many).  SCCD_OPT_LIMIT_LEVEL_ORDER = 1 runs the
    level-synchronous kernels regardless: the same answers."""
    V0, V1, E, F = _scene("cloth_ball_10k")
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    try:  # (the count to compare with: a call without the projection cull -- under the default, 1, a mesh this small is not culled either)
        ctx.set_option(sccd.OPT_CULL, 0)
        t_free, st_free = sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True, want_stats=True)
    finally:
        ctx.set_option(sccd.OPT_CULL, 1)
    assert sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True) == t_free
    free_checks = st_free["n_vf_checks"] + st_free["n_ee_checks"]
    for limit in (4096, 10_000_000):
        t, st = sccd.ccd_mesh(mesh, 0.0, limit, 1e-6, True, want_stats=True)
        assert t == t_free
        # depth-first with pruning: the count depends a little on timing, never by a factor
        assert st["n_vf_checks"] + st["n_ee_checks"] < 1.5 * free_checks
    t_ipc = sccd.ipc_ccd_strategy(V0, V1, E, F, 0.0, 10_000_000, 1e-6, ctx=ctx)
    assert t_ipc == sccd.ipc_ccd_strategy(V0, V1, E, F, 0.0, -1, 1e-6, ctx=ctx)
    try:  # (round 6) the projection cull in front of a call with a limit: the same TOI on a fraction of the pairs, the certificate on the kept list
        ctx.set_option(sccd.OPT_CULL, 2)
        for limit in (4096, 10_000_000):
            t, st = sccd.ccd_mesh(mesh, 0.0, limit, 1e-6, True, want_stats=True)
            assert t == t_free and st["n_vf_culled"] + st["n_ee_culled"] > 0.3 * (st["n_vf_pairs"] + st["n_ee_pairs"])
            assert st["n_vf_checks"] + st["n_ee_checks"] < free_checks
        assert sccd.ipc_ccd_strategy(V0, V1, E, F, 0.0, 10_000_000, 1e-6, ctx=ctx) == t_ipc
    finally:
        ctx.set_option(sccd.OPT_CULL, 1)
    try:
        ctx.set_option(sccd.OPT_LIMIT_LEVEL_ORDER, 1)
        t, st = sccd.ccd_mesh(mesh, 0.0, 10_000_000, 1e-6, True, want_stats=True)
        assert t == t_free
        level_checks = st["n_vf_checks"] + st["n_ee_checks"]
    finally:
        ctx.set_option(sccd.OPT_LIMIT_LEVEL_ORDER, 0)
    assert level_checks != free_checks  # (another traversal: it did run in level order)
    want, _, _ = orc.ccd(V0, V1, E, F, 0.0, 3, 1e-6, True)
    got = sccd.ccd_mesh(mesh, 0.0, 3, 1e-6, True)
    assert got >= t_free  # truncation can only lose collisions
    assert got == want
    try:
        ctx.set_option(sccd.OPT_CULL, 2)
        assert sccd.ccd_mesh(mesh, 0.0, 3, 1e-6, True) == want  # (level order on the kept list: the same truncated answer)
    finally:
        ctx.set_option(sccd.OPT_CULL, 1)


@pytest.mark.parametrize("arith", [0, 1])
def test_check_limits_where_the_certificate_fails_fall_back_to_level_order(sccd, ctx, orc, arith):
    """Limits >= 4096 start on the fast kernel.  The heaviest vertex-face queries of this soup are popped 13,424, 12,657 and
    8,883 times in the reference's level order; handed over without the rest of the scene they also hold the earliest
    impact, so limits of 4,096 ... 10,000 DO cut it off (TOI 1 instead of 0.125): the host-side proof fails and the call is
    redone in level order.  The oracle's answer, bit for bit, whichever way each limit went; different limits, different
    answers."""
    V0, V1, E, F = _scene("soup_dense")
    vb, eb, fb = orc.build_boxes(V0, V1, E, F, 0.0)
    pv = orc.sort_and_sweep(vb, fb)[0]
    _, pq, _ = orc.narrow_phase(V0, V1, E, F, pv, True, per_query=True, arith=arith)
    hits = np.nonzero(pq < 1)[0]
    counts = [(orc.narrow_phase(V0, V1, E, F, pv[i:i + 1], True, arith=arith)[2]["max_checks_per_query"], int(i)) for i in hits]
    heavy = [i for _, i in sorted(counts, reverse=True)[:3]]
    assert sorted(counts, reverse=True)[2][0] > 8000
    misses = np.nonzero(pq >= 1)[0][:300]  # (queries without an impact ride along)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    try:
        ctx.set_option(sccd.OPT_ARITH, arith)
        for sel in ([heavy[0]], heavy[:2], heavy):
            pairs = np.ascontiguousarray(np.concatenate([pv[sel], pv[misses]]))
            seen = set()
            for k in (4096, 6000, 10_000, 13_000, 20_000):
                want = orc.narrow_phase(V0, V1, E, F, pairs, True, 0.0, k, 1e-6, True, arith=arith)[0]
                got = sccd.narrow_phase(mesh, pairs, True, k, 1e-6, 0.0, True)
                assert got == want, (sel, k)
                ctx.set_option(sccd.OPT_LIMIT_LEVEL_ORDER, 1)
                assert sccd.narrow_phase(mesh, pairs, True, k, 1e-6, 0.0, True) == want, (sel, k)
                ctx.set_option(sccd.OPT_LIMIT_LEVEL_ORDER, 0)
                seen.add(want)
            assert len(seen) >= 2, sel
    finally:
        ctx.set_option(sccd.OPT_ARITH, sccd.ARITH_DEFAULT)
        ctx.set_option(sccd.OPT_LIMIT_LEVEL_ORDER, 0)


@pytest.mark.parametrize("arith", [0, 1])
def test_check_limits_follow_the_reference_level_order(sccd, ctx, orc, arith):
    """max_iter >= 0 (root_finder.cu:287-305): the reference counts the domains of a query as its breadth-first
    launches pop them and drops the query's domains once the count has passed the limit.  On this contact-rich soup a
    vertex-face query is popped 11,701 times in level order but needs only 3,384 depth-first checks: limits of 50,
    500 and 5,000 all truncate queries in the reference, and only the level order reproduces WHICH domains are lost.
    Bit-equal to the oracle (level-snapshot serialisation on both sides) for the TOI and for every per-query TOI."""
    V0, V1, E, F = _scene("soup_dense")
    vb, eb, fb = orc.build_boxes(V0, V1, E, F, 0.0)
    pv = orc.sort_and_sweep(vb, fb)[0]
    stats = orc.narrow_phase(V0, V1, E, F, pv, True, arith=arith)[2]
    assert stats["max_checks_per_query"] > 5000  # the limits below do bite
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    try:
        ctx.set_option(sccd.OPT_ARITH, arith)
        seen = set()
        for k in (0, 7, 50, 500, 5000, 20000):
            want = orc.ccd(V0, V1, E, F, 0.0, k, 1e-6, True, arith=arith)[0]
            for cull in (2, 0, 1):  # (forced on: the limited level order runs on the kept list -- the same answer; off; the default last)
                ctx.set_option(sccd.OPT_CULL, cull)
                assert sccd.ccd_mesh(mesh, 0.0, k, 1e-6, True) == want, (k, cull)
            seen.add(want)
            want_t, want_pq, _ = orc.narrow_phase(V0, V1, E, F, pv, True, 0.0, k, 1e-6, True, arith=arith, per_query=True)
            # per-query output with a limit: the fast kernel without the limit, then ONLY the queries that reported an impact
            # redone in level order with it (narrow.hip) -- and, the cross-check, the whole call in level order
            for level_order in (0, 1):
                ctx.set_option(sccd.OPT_LIMIT_LEVEL_ORDER, level_order)
                got_t, col = sccd.narrow_phase(mesh, pv, True, k, 1e-6, 0.0, True, want_collisions=True)
                ctx.set_option(sccd.OPT_LIMIT_LEVEL_ORDER, 0)
                assert got_t == want_t, (k, level_order)
                hit = want_pq < 1
                assert len(col) == int(hit.sum()), (k, level_order)
                assert np.array_equal(col["toi"], want_pq[hit]), (k, level_order)
        assert len(seen) >= 2  # different limits, different answers: the test can tell them apart
    finally:
        ctx.set_option(sccd.OPT_ARITH, sccd.ARITH_DEFAULT)


def test_sort_is_a_stable_permutation(sccd, ctx):
    import torch

    for n in (1, 63, 4097, 300_000):
        g = torch.Generator().manual_seed(n)
        keys = torch.randint(0, 2**31 - 1, (n,), generator=g, dtype=torch.int64)
        keys[: n // 3] &= 0xFF00  # many ties
        k = keys.to(torch.int32).cuda()
        v = torch.arange(n, dtype=torch.int32).cuda()
        torch.cuda.synchronize()
        ctx.sort_pairs_u32(k.data_ptr(), v.data_ptr(), n)
        ctx.synchronize()
        want_k, want_v = torch.sort(keys, stable=True)
        assert torch.equal(k.cpu().to(torch.int64), want_k)
        assert torch.equal(v.cpu().to(torch.int64), want_v)
    # digit distributions that stress the ranking: one key only (every lane of a row on the same counter), two keys,
    # keys already sorted, keys that differ in one digit only, and a multi-tile list with a ragged end
    n = 3_000_017
    g = torch.Generator().manual_seed(5)
    cases = {
        "one key": torch.full((n,), 0x12345678, dtype=torch.int64),
        "two keys": torch.randint(0, 2, (n,), generator=g, dtype=torch.int64) * 0x01010101,
        "sorted": torch.arange(n, dtype=torch.int64) * 5,
        "top digit only": torch.randint(0, 128, (n,), generator=g, dtype=torch.int64) << 24,
        "few keys in the middle digit": torch.randint(0, 3, (n,), generator=g, dtype=torch.int64) << 8,
    }
    for name, keys in cases.items():
        k = keys.to(torch.int32).cuda()
        v = torch.arange(n, dtype=torch.int32).cuda()
        torch.cuda.synchronize()
        ctx.sort_pairs_u32(k.data_ptr(), v.data_ptr(), n)
        ctx.synchronize()
        want_k, want_v = torch.sort(keys, stable=True)
        assert torch.equal(k.cpu().to(torch.int64), want_k), name
        assert torch.equal(v.cpu().to(torch.int64), want_v), name


# ---- BASELINE.json full size (configs[3]/[4]): 708 x 708 folded cloth, 999,698 triangles ---------
@pytest.fixture(scope="module")
def cloth1m():
    return scenes.folded_cloth(708)


def test_full_size_random_1m_boxes(sccd, ctx):
    """BASELINE configs[2] at full size: 1M random boxes, one list.  The sorted pair list must hash to
    the oracle's (tests/golden), and hold the size-independent properties: no pair twice, a < b,
    every reported pair really overlaps, sweeping along y or z reports the same set."""
    G = json.load(open(GOLDEN))["random_1m"]
    b = scenes.random_boxes(1_000_000, seed=42, max_extent=0.027)
    bp = sccd.BroadPhase(ctx)
    bp.build(sccd.DeviceAABBs(b, ctx))
    got = bp.detect_overlaps().reshape(-1, 2)
    assert len(got) == G["n"]
    srt = _sorted(got)
    assert hashlib.sha256(srt.tobytes()).hexdigest() == G["sha256"]
    assert (srt[:, 0] < srt[:, 1]).all()
    key = srt[:, 0].astype(np.int64) * len(b) + srt[:, 1]
    assert (np.diff(key) > 0).all()  # sorted and unique
    i, j = srt[::97, 0], srt[::97, 1]
    assert ((b["min"][i] <= b["max"][j]) & (b["min"][j] <= b["max"][i])).all()
    try:
        ctx.set_option(sccd.OPT_SORT_AXIS, 2)
        bp.build(sccd.DeviceAABBs(b, ctx))
        other = bp.detect_overlaps().reshape(-1, 2)
    finally:
        ctx.set_option(sccd.OPT_SORT_AXIS, 0)
    assert hashlib.sha256(_sorted(other).tobytes()).hexdigest() == G["sha256"]
    got2, next_axis = sccd.sort_and_sweep(b, sort_axis=0, ctx=ctx)
    assert len(got2) == G["n"] and next_axis == G["next_axis"]


def test_full_size_pair_sets_match_golden_hashes(sccd, ctx, cloth1m):
    """1.49 M VF + 5.06 M EE pairs: identical to the CPU restatement (count + SHA-256 of the
    sorted list), plus size-independent properties: no duplicates, valid ids, no shared vertex."""
    G = json.load(open(GOLDEN))["folded_cloth_708"]
    V0, V1, E, F = cloth1m
    assert (len(V0), len(E), len(F)) == (G["nV"], G["nE"], G["nF"]) == (501264, 1500961, 999698)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    vb, eb, fb = sccd.DeviceAABBs.from_mesh(mesh, 0.0)
    assert hashlib.sha256(vb.download().tobytes()).hexdigest() == G["sha_vertex_boxes"]
    bp = sccd.BroadPhase(ctx)
    bp.build(vb, fb)
    vf = _sorted(bp.detect_overlaps())
    bp.build(eb)
    ee = _sorted(bp.detect_overlaps())
    assert len(vf) == G["n_vf"] and hashlib.sha256(vf.tobytes()).hexdigest() == G["sha_vf"]
    assert len(ee) == G["n_ee"] and hashlib.sha256(ee.tobytes()).hexdigest() == G["sha_ee"]
    # properties that hold for any input
    assert np.all((vf[1:] != vf[:-1]).any(axis=1)) and np.all((ee[1:] != ee[:-1]).any(axis=1))  # no duplicate
    assert vf[:, 0].max() < len(V0) and vf[:, 1].max() < len(F) and ee.max() < len(E) and np.all(ee[:, 0] < ee[:, 1])
    assert not np.any(F[vf[:, 1]] == vf[:, [0]])  # the vertex is never a corner of its face
    ea, eb_ = E[ee[:, 0]], E[ee[:, 1]]
    assert not np.any((ea[:, [0]] == eb_) | (ea[:, [1]] == eb_))  # the two edges share no vertex


@pytest.mark.parametrize("arith", [0, 1])
def test_full_size_ccd_matches_golden_toi(sccd, ctx, cloth1m, arith):
    G = json.load(open(GOLDEN))["folded_cloth_708"]
    V0, V1, E, F = cloth1m
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    ctx.set_option(sccd.OPT_ARITH, arith)
    try:
        toi, st = sccd.ccd_mesh(mesh, want_stats=True)
        # idempotence: a second call on the same resident mesh returns the same bits
        toi2 = sccd.ccd_mesh(mesh)
    finally:
        ctx.set_option(sccd.OPT_ARITH, sccd.ARITH_DEFAULT)
    assert toi == toi2 == float.fromhex(G["toi_fma" if arith else "toi_strict"])
    assert st["n_vf_pairs"] == G["n_vf"] and st["n_ee_pairs"] == G["n_ee"]
    # the level-synchronous kernel (reference scheme) agrees on the VF pass at full size
    ctx.set_option(sccd.OPT_NARROW_ALGO, 1)
    ctx.set_option(sccd.OPT_ARITH, arith)
    try:
        sccd.ccd_mesh_prepare(mesh, 0.0)
        t_vf, _ = sccd.ccd_mesh_pass(mesh, True, 1.0)
    finally:
        ctx.set_option(sccd.OPT_NARROW_ALGO, 0)
        ctx.set_option(sccd.OPT_ARITH, sccd.ARITH_DEFAULT)
    assert t_vf == float.fromhex(G["toi_vf_fma" if arith else "toi_vf_strict"])


def test_ccd_from_a_callers_bound(sccd, ctx, orc):
    """sccd_ccd_mesh_from: min(bound, earliest impact below it).  Below the bound: the oracle's TOI, bit for bit; a bound at or below the
    earliest impact comes back itself; the context's own history is not touched (the next plain call returns the same as ever); and the
    job-wide protocol built on it (sccd.dist.GlobalPrior, here with one rank) walks through hits and a broken bound."""
    from sccd import dist as sdist

    V0, V1, E, F = _scene("cloth_ball_small")
    want = orc.ccd(V0, V1, E, F, 0.0, -1, 1e-6, True, nthreads=8)[0]
    assert 0.0 < want < 1.0
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    for cull, halves in ((2, 2), (0, 0), (1, 1)):
        ctx.set_option(sccd.OPT_CULL, cull)
        ctx.set_option(sccd.OPT_TWO_HALVES, halves)
        for bound in (1.0, 0.9, min(1.0, 1.125 * want), float(np.nextafter(want, 1.0))):
            assert sccd.ccd_mesh_from(mesh, bound) == want, (cull, halves, bound)
        for bound in (want, 0.5 * want, 1e-9):
            assert sccd.ccd_mesh_from(mesh, bound) == bound, (cull, halves, bound)
        assert sccd.ccd_mesh(mesh) == want
    ctx.set_option(sccd.OPT_CULL, 1)
    ctx.set_option(sccd.OPT_TWO_HALVES, 1)
    with pytest.raises(Exception):
        sccd.ccd_mesh_from(mesh, 0.0)
    gp = sdist.GlobalPrior()
    run = lambda b: sccd.ccd_mesh_from(mesh, b, want_stats=True)  # noqa: E731
    assert gp.step(run)[0] == want and gp.bound == min(1.0, 1.125 * want)
    assert gp.step(run)[0] == want and (gp.hits, gp.misses) == (1, 0)
    mesh.update_vertices(V0, V0 + 0.5 * (V1 - V0))  # the impact moves to twice the time: beyond the bound
    later = orc.ccd(V0, V0 + 0.5 * (V1 - V0), E, F, 0.0, -1, 1e-6, True, nthreads=8)[0]
    assert later > gp.bound
    assert gp.step(run)[0] == later and (gp.hits, gp.misses) == (1, 1)
    assert gp.step(run)[0] == later and (gp.hits, gp.misses) == (2, 1)
    mesh.close()


def test_full_size_strategies_agree_where_the_default_rules_pick_them(sccd, orc, cloth1m):
    """The 1M-triangle cloth is large enough for the DEFAULT settings (SCCD_OPT_CULL = SCCD_OPT_TWO_HALVES = 1) to use the projection
    cull and the two halves of time, and with history on (SCCD_OPT_TOI_GUESS, the default) the last call on the mesh decides between one
    narrow launch per pass and two and lends the next call its bound.  The step as it is (impact at 0.408), cut to 0.6 (impact at 0.68:
    the bet of the two halves is lost) and to 0.3 (no impact): every strategy returns the same bits -- the defaults called three times in
    a row (first call, history settled, bound tried), no history, one launch per pass, no cull -- and the full step's are the golden TOI."""
    G = json.load(open(GOLDEN))["folded_cloth_708"]
    V0, V1, E, F = cloth1m
    c = sccd.Context(0)
    try:
        mesh = sccd.Mesh(V0, V1, E, F, ctx=c)
        seen = []
        for s in (1.0, 0.6, 0.3):
            mesh.update_vertices(V0, V0 + s * (V1 - V0))
            got = {}
            c.set_option(sccd.OPT_TOI_GUESS, 1)
            got["defaults"] = [sccd.ccd_mesh(mesh, want_stats=True) for _ in range(3)]
            st = got["defaults"][0][1]
            assert st["n_vf_culled"] + st["n_ee_culled"] > 0.5 * (st["n_vf_pairs"] + st["n_ee_pairs"]), st  # (the size rule let the cull run)
            ref = got["defaults"][0][0]
            assert all(t == ref for t, _ in got["defaults"]), (s, [t for t, _ in got["defaults"]])
            c.set_option(sccd.OPT_TOI_GUESS, 0)
            for name, cull, halves in (("no_history", 1, 1), ("one_launch", 1, 0), ("no_cull", 0, 1), ("neither", 0, 0)):
                c.set_option(sccd.OPT_CULL, cull)
                c.set_option(sccd.OPT_TWO_HALVES, halves)
                t, st2 = sccd.ccd_mesh(mesh, want_stats=True)
                assert t == ref, (s, name, t, ref)
                assert (st2["n_vf_pairs"], st2["n_ee_pairs"]) == (st["n_vf_pairs"], st["n_ee_pairs"])
            c.set_option(sccd.OPT_CULL, 1)
            c.set_option(sccd.OPT_TWO_HALVES, 1)
            seen.append(ref)
        assert seen[0] == float.fromhex(G["toi_fma"]) and 0.5 < seen[1] < 1.0 and seen[2] == 1.0, seen
        assert seen[1] == orc.ccd(V0, V0 + 0.6 * (V1 - V0), E, F, 0.0, -1, 1e-6, True, nthreads=64)[0]  # (the late impact: the oracle's)
    finally:
        c.close()


def test_full_size_sharded_passes_reduce_to_the_same_toi(sccd, ctx, cloth1m):
    """4 ranks emulated on one GPU: the shards partition the queries and the min over the ranks'
    TOIs, threaded VF -> EE like dist.ccd_sharded does, is the single-GPU TOI."""
    G = json.load(open(GOLDEN))["folded_cloth_708"]
    V0, V1, E, F = cloth1m
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    world = 4
    try:
        sccd.ccd_mesh_prepare(mesh, 0.0)
        toi, n_pairs = 1.0, 0
        for is_vf in (True, False):
            parts = []
            for r in range(world):
                ctx.set_option(sccd.OPT_SHARD_COUNT, world)
                ctx.set_option(sccd.OPT_SHARD_RANK, r)
                t, st = sccd.ccd_mesh_pass(mesh, is_vf, toi)
                parts.append(t)
                n_pairs += st["n_vf_pairs"] + st["n_ee_pairs"]
                assert st["n_vf_pairs"] + st["n_ee_pairs"] > 0.15 * (G["n_vf"] if is_vf else G["n_ee"])  # balanced
            toi = min(parts)  # the all-reduce(min)
    finally:
        ctx.set_option(sccd.OPT_SHARD_COUNT, 1)
        ctx.set_option(sccd.OPT_SHARD_RANK, 0)
    assert n_pairs == G["n_vf"] + G["n_ee"]
    assert toi == float.fromhex(G["toi_strict"])


@pytest.mark.parametrize("world", [1, 3])
def test_repeated_steps_on_one_mesh_build_speculatively_and_agree(sccd, ctx, orc, world):
    """A simulation calls ccd() step after step on one mesh: from the second step on the broad phase is enqueued on the previous
    step's entry counts (csrc/api.hip bp_build, the speculative build) -- also a rank's cell window of a multi-GPU job.  Moving
    the vertices a little keeps the guess, moving them a lot breaks it; every step's TOI and pair count are the oracle's."""
    V0, V1, E, F = scenes.folded_cloth(60, seed=3)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    rng = np.random.default_rng(11)
    try:
        for step, amp in enumerate([0.0, 1e-5, 1e-5, 3e-2, 1e-5, 0.0, 2e-3, 2e-3, 8e-2, 1e-4, 1e-4, 0.0]):
            W0 = V0 + rng.uniform(-amp, amp, V0.shape)
            W1 = V1 + rng.uniform(-amp, amp, V1.shape)
            mesh.update_vertices(W0, W1)
            want, nvf, nee = orc.ccd(W0, W1, E, F, nthreads=8)
            tois, pairs = [], 0
            for r in range(world):
                ctx.set_option(sccd.OPT_SHARD_COUNT, world)
                ctx.set_option(sccd.OPT_SHARD_RANK, r)
                t, st = sccd.ccd_mesh(mesh, want_stats=True)
                tois.append(t)
                pairs += st["n_vf_pairs"] + st["n_ee_pairs"]
            assert min(tois) == want, (step, amp, tois, want)
            assert pairs == nvf + nee, (step, amp, pairs, nvf + nee)
    finally:
        ctx.set_option(sccd.OPT_SHARD_COUNT, 1)
        ctx.set_option(sccd.OPT_SHARD_RANK, 0)


def test_ccd_with_collisions(sccd, ctx, orc):
    """ccd() of a TOI_PER_QUERY build (ccd.cu:14-78 with `collisions`): vertex-face records, then
    edge-edge; the running minimum of the list is the returned TOI, bit for bit."""
    V0, V1, E, F = scenes.cloth_ball(20, 1, seed=3)
    toi, col = sccd.ccd(V0, V1, E, F, 0.0, -1, 1e-6, True, ctx=ctx, want_collisions=True)
    assert toi == sccd.ccd(V0, V1, E, F, 0.0, -1, 1e-6, True, ctx=ctx)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    want = []
    for is_vf, sweep in ((True, orc.sort_and_sweep(vb, fb)), (False, orc.sort_and_sweep(eb))):
        pairs = sweep[0]
        _, pq, _ = orc.narrow_phase(V0, V1, E, F, pairs, is_vf, per_query=True)
        hit = pq < 1.0
        want.append(sorted((int(a), int(b), float(t)) for (a, b), t in zip(np.asarray(pairs)[hit], pq[hit])))
    n_vf = len(want[0])
    got_vf = sorted((int(r["aid"]), int(r["bid"]), float(r["toi"])) for r in col[:n_vf])
    got_ee = sorted((int(r["aid"]), int(r["bid"]), float(r["toi"])) for r in col[n_vf:])
    assert got_vf == want[0] and got_ee == want[1]
    assert len(col) and min(float(r["toi"]) for r in col) == toi
    # (round 6) the projection cull in front of the per-query narrow phase: a culled pair has no impact, hence no record -- the same
    # list, forced on this small mesh, with and without a check limit that no query reaches
    try:
        ctx.set_option(sccd.OPT_CULL, 2)
        for limit in (-1, 10_000_000):
            toi2, col2 = sccd.ccd(V0, V1, E, F, 0.0, limit, 1e-6, True, ctx=ctx, want_collisions=True)
            assert toi2 == toi and len(col2) == len(col)
            assert sorted((int(r["aid"]), int(r["bid"]), float(r["toi"])) for r in col2[:n_vf]) == want[0]
            assert sorted((int(r["aid"]), int(r["bid"]), float(r["toi"])) for r in col2[n_vf:]) == want[1]
    finally:
        ctx.set_option(sccd.OPT_CULL, 1)


@pytest.mark.parametrize("tol", [1e-9, 1e-11, 1e-13])
def test_bisection_deeper_than_the_compressed_entries(sccd, ctx, orc, tol):
    """The work-queue kernel holds an interval as (numerator, level <= 31); a tolerance that needs deeper
    bisection makes it hand the call to the level-synchronous kernel (NQ_OVF_INTERVAL) -- same bits either way."""
    V0 = np.array([[0.3, 0.3, 0.7], [0, 0, 0], [1, 0, 0], [0, 1, 0], [0.1, 0.2, 0.9], [0.9, 0.2, 0.8]], float)
    V1 = V0.copy()
    V1[0, 2] = -0.45
    V1[4] = [0.15, 0.25, -0.3]
    V1[5] = [0.85, 0.15, -0.2]
    E = np.array([[1, 2], [2, 3], [1, 3], [4, 5]], np.int32)
    F = np.array([[1, 2, 3]], np.int32)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    for is_vf, pairs in ((True, [[0, 0], [4, 0], [5, 0]]), (False, [[0, 3], [1, 3], [2, 3]])):
        want, want_pq, _ = orc.narrow_phase(V0, V1, E, F, pairs, is_vf, tol=tol, per_query=True)
        got = sccd.narrow_phase(mesh, pairs, is_vf, tol=tol)
        assert got == want and got < 1.0
        # the per-query output takes the same route: the queries concerned are listed by the kernel and redone in level order
        got_t, col = sccd.narrow_phase(mesh, pairs, is_vf, tol=tol, want_collisions=True)
        hit = want_pq < 1
        assert got_t == want and len(col) == int(hit.sum()) and np.array_equal(col["toi"], want_pq[hit])


def test_only_the_queries_beyond_level_31_are_redone_in_level_order(sccd, ctx, orc):
    """A scene measured in millimetres: tolerance / (3 x extent) drops below 2^-31 for the queries in contact, which the
    work-queue kernel cannot hold as (numerator, level).  It lists THOSE queries and they alone are redone by the
    level-synchronous kernels (round 1 redid the whole call: several times slower, and out of memory on contact-rich
    scenes) -- the time of impact is the oracle's, and the check count stays far below a level-order pass over all."""
    V0, V1, E, F = scenes.cloth_ball(24, 1, seed=3)
    scale = 2000.0
    V0, V1 = V0 * scale, V1 * scale
    want, n_vf, n_ee = orc.ccd(V0, V1, E, F, 0.0, -1, 1e-6, True, nthreads=8)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    got, st = sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True, want_stats=True)
    assert got == want and got < 1.0
    try:
        ctx.set_option(sccd.OPT_NARROW_ALGO, 1)  # everything in level order
        lvl, st_lvl = sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True, want_stats=True)
    finally:
        ctx.set_option(sccd.OPT_NARROW_ALGO, 0)
    assert lvl == want
    assert st["n_vf_checks"] > 0 and st_lvl["n_vf_checks"] > 0


@pytest.mark.parametrize("s", [1.0, 0.41, 0.3, 0.2])
def test_queries_beyond_level_31_in_the_second_half_of_time(sccd, orc, s):
    """ADVICE r05 (high).  Two halves of time with a cull per slab: the second launch walks a list of its OWN (the pairs kept for
    [0.5, b]) in plain mode, and a query it cannot hold as (numerator, level <= 31) only raises a flag.  The fallback used to redo the
    FIRST half's list alone -- a query that lives in the second list only (no approach before 0.5, an impact after it) was never
    bisected to the end and its impact was lost.  The millimetre-scale scene of the test above (every contact query overflows), its
    motion cut so that the earliest impact falls before 0.5, after it (0.62, 0.85), or nowhere: the oracle's TOI each time, with the
    halves and the cull forced, and the same with both off."""
    V0, V1, E, F = scenes.cloth_ball(24, 1, seed=3)
    V0, V1 = V0 * 2000.0, V1 * 2000.0
    W1 = V0 + s * (V1 - V0)
    want = orc.ccd(V0, W1, E, F, 0.0, -1, 1e-6, True, nthreads=8)[0]
    assert (want < 0.5) if s == 1.0 else (0.5 < want < 1.0) if s > 0.25 else want == 1.0
    c = sccd.Context(0)
    try:
        c.set_option(sccd.OPT_TOI_GUESS, 0)
        mesh = sccd.Mesh(V0, W1, E, F, ctx=c)
        for halves, cull in ((2, 2), (2, 0), (0, 2), (0, 0)):
            c.set_option(sccd.OPT_TWO_HALVES, halves)
            c.set_option(sccd.OPT_CULL, cull)
            got = sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True)
            assert got == want, (s, halves, cull, got, want)
        mesh.close()
    finally:
        c.close()


@pytest.mark.parametrize("arith", [0, 1])
def test_float_build_matches_the_oracle_twin(sccd, ctx, orc, arith):
    """SCCD_OPT_SCALAR = 1 = the reference with SCALABLE_CCD_USE_DOUBLE=OFF: vertices cast to float first, float boxes
    (stored widened), float Tight-Inclusion on the level-synchronous kernels.  Boxes, pair sets and the TOI must be the
    oracle twin's (orc_*_f32), bit for bit.  (Added after gpurun closed in round 1: the arithmetic header is checked on
    the host, tests/test_ti_f32_host.py; this test has not run on a GPU yet.)"""
    for (V0, V1, E, F), ms in ((scenes.cloth_ball(20, 1, seed=3), 0.0), (scenes.triangle_soup(150, seed=9, size=0.12, motion=0.3), 1e-3)):
        want_vb, want_eb, want_fb = orc.build_boxes(V0, V1, E, F, ms, scalar="f32")
        want_vf, _, _ = orc.sort_and_sweep(want_vb, want_fb)
        want_ee, _, _ = orc.sort_and_sweep(want_eb)
        want_toi, _, _ = orc.ccd(V0, V1, E, F, ms, -1, 1e-6, True, arith=arith, nthreads=4, scalar="f32")
        ctx.set_option(sccd.OPT_SCALAR, 1)
        ctx.set_option(sccd.OPT_ARITH, arith)
        try:
            vb = sccd.build_vertex_boxes(V0, V1, ms, ctx=ctx)
            for f in ("min", "max", "vertex_ids", "element_id"):
                assert np.array_equal(vb[f], want_vb[f])
            eb, fb = sccd.build_edge_boxes(vb, E, ctx=ctx), sccd.build_face_boxes(vb, F, ctx=ctx)
            bp = sccd.BroadPhase(ctx)
            bp.build(sccd.DeviceAABBs(vb, ctx), sccd.DeviceAABBs(fb, ctx))
            assert np.array_equal(_sorted(bp.detect_overlaps()), want_vf)
            bp.build(sccd.DeviceAABBs(eb, ctx))
            assert np.array_equal(_sorted(bp.detect_overlaps()), want_ee)
            toi = sccd.ccd(V0, V1, E, F, ms, -1, 1e-6, True, ctx=ctx)  # the float build's depth-first kernel (np_walk_f32_k)
            assert toi == want_toi and toi == float(np.float32(toi))
            ctx.set_option(sccd.OPT_NARROW_ALGO, 1)  # ... and the level-synchronous kernels
            assert sccd.ccd(V0, V1, E, F, ms, -1, 1e-6, True, ctx=ctx) == want_toi
            ctx.set_option(sccd.OPT_NARROW_ALGO, 0)
            # per-query output in float: every reported impact is the oracle twin's, both ways
            mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
            want_t, want_pq, _ = orc.narrow_phase(V0, V1, E, F, want_vf, True, ms=ms, arith=arith, per_query=True, scalar="f32")
            for algo in (0, 1):
                ctx.set_option(sccd.OPT_NARROW_ALGO, algo)
                got_t, col = sccd.narrow_phase(mesh, want_vf, True, -1, 1e-6, ms, True, want_collisions=True)
                ctx.set_option(sccd.OPT_NARROW_ALGO, 0)
                hit = want_pq < 1
                assert got_t == want_t and len(col) == int(hit.sum()), algo
                assert np.array_equal(col["toi"], want_pq[hit].astype(np.float64)), algo
            # a tolerance that drives the bisection past level 23 (floats of the form k 2^-d end there): those queries are
            # listed by the kernel and redone in level order -- the same TOI as level order throughout
            tiny = orc.ccd(V0, V1, E, F, ms, -1, 1e-9, True, arith=arith, nthreads=4, scalar="f32")[0]
            assert sccd.ccd(V0, V1, E, F, ms, -1, 1e-9, True, ctx=ctx) == tiny
            mesh.close()
        finally:
            ctx.set_option(sccd.OPT_NARROW_ALGO, 0)
            ctx.set_option(sccd.OPT_SCALAR, 0)
            ctx.set_option(sccd.OPT_ARITH, sccd.ARITH_DEFAULT)


@pytest.mark.parametrize("n_active", [64, 24, 1, 0])
def test_lds_direct_gather_idiom(sccd, ctx, n_active):
    """The narrow-phase kernel issues its LDS-direct gathers as inline assembly that the compiler's wait-count
    bookkeeping does not see (narrow_walk.inc: nq_glds16, waited for by hand at the hand-over).  This pins the idiom
    against a compiler update: landed layout (piece p of lane l at base + (p * 64 + l) * 16), inactive lanes
    untouched, a late hand-placed wait behind other LDS traffic, per-wave M0 bases with two waves per block."""
    assert ctx.selftest_lds_gather(n_waves=512, n_active=n_active) == 0


def test_walk_helpers_on_the_device_equal_the_host():
    """ti_math.hpp's stackless-walk helpers (nq_descend / nq_backtrack / nq_donate: 96-bit path arithmetic) give the
    same domains and path bits on the device as compiled for the host, over 1.6 million random states up to depth 93
    (tests/gpu_probe/walk_probe.hip; the host side is what tests/cpp/test_ti_host.cpp checks against the oracle)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "gpu_probe", "walk_probe")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", root, exe[len(root) + 1:]])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "WALK PROBE: 0 mismatches" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("seed", [9143, 9220, 9040, 9110, 9124, 9348, 9392])
def test_soak_seeds_that_once_failed(sccd, ctx, orc, seed):
    """Scenes of tools/soak.py on which a build once returned a later time of impact than the oracle: the wave-local
    TOI copy was assembled from two `readfirstlane` results (int) without going through `unsigned`, so a low word
    with its top bit set smeared over the high word and the wave pruned with garbage.  The 1M-triangle cloth and the
    rest of this suite happened to have TOIs whose low word is positive."""
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("soak_tool", os.path.join(root, "tools", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    V0, V1, E, F, kind, scale, shift, ms, allow_zero, arith, world, sweep_algo, scan_build, narrow_algo = soak.scene_of(seed)
    want = orc.ccd(V0, V1, E, F, ms, -1, 1e-6, allow_zero, arith=arith, nthreads=8)[0]
    try:
        ctx.set_option(sccd.OPT_ARITH, arith)
        ctx.set_option(sccd.OPT_SWEEP_ALGO, sweep_algo)
        mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
        for _ in range(3):  # (the failure depended on timing)
            assert sccd.ccd_mesh(mesh, ms, -1, 1e-6, allow_zero) == want
    finally:
        ctx.set_option(sccd.OPT_ARITH, sccd.ARITH_DEFAULT)
        ctx.set_option(sccd.OPT_SWEEP_ALGO, 0)


@pytest.mark.parametrize("seed", list(range(12_000, 12_048)))
def test_soak_scenes(sccd, ctx, orc, seed):
    """48 scenes of tools/soak.py (cloth-ball, folded cloth, triangle soups; scales 1e-3 .. 1e3, shifts up to 1e4, minimum
    separations, both zero-TOI policies and arithmetic contracts, 1 / 2 / 3 / 8 shards, the three sweep algorithms): pair
    sets and the time of impact against the oracle.  (The tool itself runs thousands of them; these travel with the suite.)"""
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("soak_tool", os.path.join(root, "tools", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    V0, V1, E, F, kind, scale, shift, ms, allow_zero, arith, world, sweep_algo, scan_build, narrow_algo = soak.scene_of(seed)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F, ms)
    want_vf = orc.sort_and_sweep(vb, fb, nthreads=8)[0]
    want_ee = orc.sort_and_sweep(eb, nthreads=8)[0]
    want = orc.ccd(V0, V1, E, F, ms, -1, 1e-6, allow_zero, arith=arith, nthreads=8)[0]
    try:
        ctx.set_option(sccd.OPT_ARITH, arith)
        ctx.set_option(sccd.OPT_SWEEP_ALGO, sweep_algo)
        mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
        dv, de, df = sccd.DeviceAABBs.from_mesh(mesh, ms)
        bp = sccd.BroadPhase(ctx)
        tois, got_vf, got_ee = [], [], []
        for r in range(world):
            ctx.set_option(sccd.OPT_SHARD_COUNT, world)
            ctx.set_option(sccd.OPT_SHARD_RANK, r)
            bp.build(dv, df)
            got_vf.append(bp.detect_overlaps().reshape(-1, 2))
            bp.build(de)
            got_ee.append(bp.detect_overlaps().reshape(-1, 2))
            tois.append(sccd.ccd_mesh(mesh, ms, -1, 1e-6, allow_zero))
        assert np.array_equal(_sorted(np.concatenate(got_vf)), want_vf)  # every pair from exactly one shard
        assert np.array_equal(_sorted(np.concatenate(got_ee)), want_ee)
        assert min(tois) == want
    finally:
        ctx.set_option(sccd.OPT_SHARD_COUNT, 1)
        ctx.set_option(sccd.OPT_SHARD_RANK, 0)
        ctx.set_option(sccd.OPT_ARITH, sccd.ARITH_DEFAULT)
        ctx.set_option(sccd.OPT_SWEEP_ALGO, 0)


def test_option_defaults_and_the_retired_id(sccd):
    """A fresh context computes in the reference's production contract (nvcc --use_fast_math => fused, CMakeLists.txt:219-225);
    option id 12 (SCCD_OPT_MAX_ITER_FAST of 0.1, opposite sense) is refused instead of silently selecting the slow path."""
    c = sccd.Context(0)
    try:
        assert c.get_option(sccd.OPT_ARITH) == sccd.ARITH_DEFAULT == sccd.ARITH_FMA
        assert sccd.lib().sccd_set_option(c._h, 12, 1) != 0
        assert c.get_option(sccd.OPT_LIMIT_LEVEL_ORDER) == 0
        c.set_option(sccd.OPT_LIMIT_LEVEL_ORDER, 1)
        assert c.get_option(sccd.OPT_LIMIT_LEVEL_ORDER) == 1
    finally:
        c.close()


def test_ccd_with_many_bad_indices_fails_fast_and_leaves_toi_alone(sccd, ctx):
    """sccd_ccd() from host matrices runs the step on clamped indices until the verdict is in (every bad element hangs on
    vertex 0): the call must still come back quickly with the error, and *toi must not be written (sccd.h)."""
    import ctypes as C
    import time

    V0, V1, E, F = _scene("cloth_ball_small")
    badF = np.array(F, np.int32, copy=True)
    badF[::3, 1] = len(V0) + 7  # a third of the faces
    badE = np.array(E, np.int32, copy=True)
    badE[::2, 0] = -5
    V0c, V1c = np.asfortranarray(V0, dtype=np.float64), np.asfortranarray(V1, dtype=np.float64)
    Ec, Fc = np.asfortranarray(badE, dtype=np.int32), np.asfortranarray(badF, dtype=np.int32)
    t = C.c_double(0.75)
    t0 = time.perf_counter()
    rc = sccd.lib().sccd_ccd(ctx._h, V0c.ctypes.data_as(C.c_void_p), V1c.ctypes.data_as(C.c_void_p), C.c_int(len(V0)),
                             Ec.ctypes.data_as(C.c_void_p), C.c_int(len(E)), Fc.ctypes.data_as(C.c_void_p), C.c_int(len(F)),
                             C.c_double(0.0), C.c_int(-1), C.c_double(1e-6), C.c_int(1), C.c_int(0), C.byref(t))
    dt = time.perf_counter() - t0
    assert rc != 0 and t.value == 0.75
    assert dt < 20.0, dt
    assert sccd.ccd(V0, V1, E, F, ctx=ctx) == sccd.ccd(V0, V1, E, F, ctx=ctx)  # the context is usable afterwards


def test_ccd_mesh_dev_leaves_the_toi_in_device_memory(sccd, ctx):
    """sccd_ccd_mesh_dev: the step's TOI also lands in a caller-owned device word, by a copy on the context's stream (what a
    multi-GPU caller all-reduces in place: sccd/dist.py DeviceMin)."""
    import torch

    V0, V1, E, F = _scene("cloth_ball_small")
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    word = torch.full((1,), 7.0, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    t, st = sccd.ccd_mesh_dev(mesh, word.data_ptr(), want_stats=True)
    ctx.synchronize()
    assert word.item() == t == sccd.ccd_mesh(mesh) and st["n_vf_pairs"] > 0
    assert ctx.stream_ptr() != 0
    from sccd import dist as sdist

    dm = sdist.DeviceMin(ctx, torch.device("cuda", 0))  # (no process group: reduce() is a no-op, value() reads the word)
    sccd.ccd_mesh_dev(mesh, dm.ptr())
    dm.reduce()
    assert dm.value() == t
    mesh.close()


def test_speculative_toi_bound_is_exact_through_hits_and_misses(sccd, orc):
    """ccd() on a mesh whose previous call found an impact at T starts from the bound 1.125 T (drivers.hip ccd_on_mesh): a result
    below the bound is exact, a result AT the bound proves nothing and the step is redone from 1.  The mesh is stepped through
    motions that keep the TOI (bound holds), push it beyond the bound (miss), remove every impact (miss, result 1) and bring it
    back: every result is the oracle's, and both the hits and the misses really happened."""
    if os.environ.get("SCCD_SPECULATE") == "0":
        pytest.skip("SCCD_SPECULATE=0 switches the speculative bound off with the speculative build")
    c = sccd.Context(0)
    try:
        V0, V1, E, F = _scene("cloth_ball_small")
        mesh = sccd.Mesh(V0, V1, E, F, ctx=c)
        # (after a miss the library rests for a few steps before it tries a bound again: the sequence gives it the time)
        scales = [1.0, 1.0, 0.999, 0.55, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.05, 0.05, 1.0, 1.0, 0.3, 0.3, 0.3, 0.3, 0.3, 1.0]
        seen = []
        for s in scales:
            W1 = V0 + s * (V1 - V0)
            mesh.update_vertices(V0, W1)
            want = orc.ccd(V0, W1, E, F, 0.0, -1, 1e-6, True, nthreads=8)[0]
            got = sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True)
            assert got == want, (s, got, want)
            seen.append(want)
        hits, misses = c.get_option(sccd.OPT_TOI_GUESS_HITS), c.get_option(sccd.OPT_TOI_GUESS_MISSES)
        assert hits >= 2 and misses >= 2, (hits, misses, seen)
        assert min(seen) < 1.0 and max(seen) == 1.0, seen  # the sequence had impacts and a step without any
        c.set_option(sccd.OPT_TOI_GUESS, 0)  # ... and switched off, every call starts from 1
        c.set_option(sccd.OPT_TOI_GUESS_HITS, 0)
        for s in (1.0, 1.0, 1.0):
            mesh.update_vertices(V0, V0 + s * (V1 - V0))
            assert sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True) == seen[0]
        assert c.get_option(sccd.OPT_TOI_GUESS_HITS) == 0 and c.get_option(sccd.OPT_TOI_GUESS_MISSES) == 0
        mesh.close()
    finally:
        c.close()


@pytest.mark.parametrize("path", ["default", "passes_apart", "narrow_algo_1", "cutoff"])
def test_speculative_toi_bound_is_exact_in_the_float_build(sccd, orc, path):
    """the bound is 1.125 x a float TOI -- not a float -- and the float build's run_narrow() paths (passes apart, the
    level-order algorithm, chunked sweeps) round the TOI they start from to float: the bound has to BE that rounded value, or a call
    in which nothing is accepted below it returns the rounded bound as if it were a hit.  Stepped through holds, misses and
    no-impact steps on every such path, against the oracle's float twin."""
    if os.environ.get("SCCD_SPECULATE") == "0":
        pytest.skip("SCCD_SPECULATE=0 switches the speculative bound off with the speculative build")
    c = sccd.Context(0)
    try:
        c.set_option(sccd.OPT_SCALAR, 1)
        if path == "passes_apart":
            c.set_option(sccd.OPT_PASSES_APART, 1)
        elif path == "narrow_algo_1":
            c.set_option(sccd.OPT_NARROW_ALGO, 1)
        elif path == "cutoff":
            c.set_option(sccd.OPT_MAX_OVERLAP_CUTOFF, 700)
        V0, V1, E, F = _scene("cloth_ball_small")
        mesh = sccd.Mesh(V0, V1, E, F, ctx=c)
        # (scales chosen so that 1.125 x TOI is not float-representable; 0.55 / 0.05 push the impact beyond the bound / away)
        scales = [1.0, 1.0, 0.9, 0.55, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.05, 0.05, 0.77, 0.77, 0.3, 0.3, 0.3, 0.3, 0.3, 0.77]
        seen = []
        for s in scales:
            W1 = V0 + s * (V1 - V0)
            mesh.update_vertices(V0, W1)
            want = orc.ccd(V0, W1, E, F, 0.0, -1, 1e-6, True, nthreads=8, scalar="f32")[0]
            got = sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True)
            assert got == want, (path, s, got, want)
            seen.append(want)
        hits, misses = c.get_option(sccd.OPT_TOI_GUESS_HITS), c.get_option(sccd.OPT_TOI_GUESS_MISSES)
        assert hits >= 2 and misses >= 1, (hits, misses, seen)
        mesh.close()
    finally:
        c.close()


# ---- the step without a host in it (csrc/drivers.hip ccd_on_mesh_from) -------------------------------------------------------
@pytest.mark.parametrize("halves", [0, 2])
@pytest.mark.parametrize("scene", ["cloth_ball_small", "cloth_ball_10k", "folded_120"])
def test_a_warm_default_step_reads_nothing_back(sccd, orc, scene, halves):
    """One thread enqueues both passes' chains and then waits for two VERDICTS (np_verdict_k: the walk launch's counters, the sweep's
    counters, the grid and entry counts of the speculative build, in pinned memory): a step on a mesh the context has stepped before
    launches NO read-back kernel and waits exactly twice -- unless a pass runs the second half of its time (its first launch accepted
    nothing before 0.5): that pass's final counters then come by one read-back.  (task 1: "a test counts ReadBack::sync
    calls per step".)  The result is the oracle's on every step, warm or not."""
    V0, V1, E, F = _scene(scene)
    want = orc.ccd(V0, V1, E, F, 0.0, -1, 1e-6, True, nthreads=8)[0]
    c = sccd.Context(0)
    try:
        c.set_option(sccd.OPT_TOI_GUESS, 0)
        c.set_option(sccd.OPT_TWO_HALVES, halves)
        c.set_option(sccd.OPT_CULL, 2)
        mesh = sccd.Mesh(V0, V1, E, F, ctx=c)
        for _ in range(3):  # (the first call builds the slow way and allocates; the second meets buffers sized by the first)
            assert sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True) == want
        # (laboratory switches that make builds wait for their counts, or break every third guess on purpose: results only)
        counted = os.environ.get("SCCD_SPECULATE") != "0" and not os.environ.get("SCCD_SPEC_BREAK")
        for _ in range(4):
            rb0, w0, sp0 = c.get_option(sccd.OPT_READ_BACKS), c.get_option(sccd.OPT_HOST_WAITS), c.get_option(sccd.OPT_SPEC_MISSES)
            assert sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True) == want
            if not counted:
                continue
            rb, w = c.get_option(sccd.OPT_READ_BACKS) - rb0, c.get_option(sccd.OPT_HOST_WAITS) - w0
            assert c.get_option(sccd.OPT_SPEC_MISSES) == sp0
            second_half = halves == 2 and want >= 0.5  # (both passes start from 1: each runs its second half iff nothing lies before 0.5)
            assert (rb, w) == ((2, 4) if second_half else (0, 2)), (scene, halves, want, rb, w)
            assert c.get_option(sccd.OPT_DEVICE_SPAN_NS) > 0
        mesh.close()
    finally:
        c.close()


# ---- two halves of time (csrc/narrow_walk.inc) ----------------------------------------------------------------------------
@pytest.mark.parametrize("scene", ["cloth_ball_small", "soup_dense", "folded_120"])
def test_two_halves_of_time_change_no_result(sccd, orc, scene):
    """A plain narrow launch from a TOI above 0.5 runs as two launches (the first from the bound 0.5, the second -- only if the first
    accepted nothing -- over what lies at or beyond 0.5).  The motion is scaled so that the earliest impact falls before 0.5 (the first
    launch decides), between 0.5 and 1 (the second one does), nowhere (both run, result 1), with and without the cull, both zero-TOI
    policies: every result is the oracle's, and the option switched off gives the same."""
    V0, V1, E, F = _scene(scene)
    c = sccd.Context(0)
    try:
        c.set_option(sccd.OPT_TOI_GUESS, 0)
        mesh = sccd.Mesh(V0, V1, E, F, ctx=c)
        seen = []
        for s in (3.0, 1.0, 0.8, 0.62, 0.5, 0.41, 0.3, 0.2, 0.1, 0.02):
            W1 = V0 + s * (V1 - V0)
            mesh.update_vertices(V0, W1)
            for allow_zero in (True, False):
                want = orc.ccd(V0, W1, E, F, 0.0, -1, 1e-6, allow_zero, nthreads=8)[0]
                for cull in (2, 0):  # (2: forced -- under the defaults, 1, meshes this small run neither the cull nor the two halves)
                    c.set_option(sccd.OPT_CULL, cull)
                    c.set_option(sccd.OPT_TWO_HALVES, 2)
                    two = sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, allow_zero)
                    c.set_option(sccd.OPT_TWO_HALVES, 0)
                    one = sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, allow_zero)
                    assert two == want and one == want, (scene, s, allow_zero, cull, two, one, want)
            seen.append(want)
        if scene == "cloth_ball_small":  # (the soup collides at once whatever the scale, the small folded cloth never: covered, not counted)
            assert min(seen) < 0.5 and any(0.5 <= t < 1.0 for t in seen) and max(seen) == 1.0, seen  # either launch got to decide, and neither
        # the pass-by-pass API (a rank of a multi-GPU job) and a start from a caller's TOI between 0.5 and 1
        c.set_option(sccd.OPT_CULL, 2)
        c.set_option(sccd.OPT_TWO_HALVES, 2)
        mesh.update_vertices(V0, V1)
        sccd.ccd_mesh_prepare(mesh, 0.0)
        want = orc.ccd(V0, V1, E, F, 0.0, -1, 1e-6, True, nthreads=8)[0]
        for t0 in (1.0, 0.75, 0.5, 0.3):
            t = t0
            for is_vf in (True, False):
                t, _ = sccd.ccd_mesh_pass(mesh, is_vf, t)
            assert t == min(t0, want), (scene, t0, t, want)
        mesh.close()
    finally:
        c.close()


# ---- the projection cull (csrc/narrow_cull.inc) -----------------------------------------------------------------------------
def _cull_scenes():
    """(name, V0, V1, E, F, ms): ordinary scenes and the regimes the cull's bound has to get right -- slow and static pairs (the
    reference's edge-edge tolerances come from the wrong pairing there: Condition 1 accepts domains with large images), large and
    shifted coordinates (the numerical-error bound grows with the cube of the magnitude), minimum separations, resting contact."""
    out = []
    V0, V1, E, F = scenes.cloth_ball(24, 1, seed=3)
    out.append(("cloth_ball", V0, V1, E, F, 0.0))
    out.append(("cloth_ball_ms", V0, V1, E, F, 2e-3))
    out.append(("slow", V0, V0 + 1e-7 * (V1 - V0), E, F, 0.0))
    out.append(("very_slow_ms", V0, V0 + 1e-9 * (V1 - V0), E, F, 1e-3))
    out.append(("static", V0, V0.copy(), E, F, 0.0))
    out.append(("scaled_1e3", 1e3 * V0 + 500.0, 1e3 * V1 + 500.0, E, F, 0.0))
    out.append(("shifted_1e5", V0 + 1e5, V1 + 1e5, E, F, 0.0))
    Vs0, Vs1, Es, Fs = scenes.triangle_soup(700, seed=11, size=0.1, motion=0.3)
    out.append(("soup", Vs0, Vs1, Es, Fs, 1e-4))
    Vf0, Vf1, Ef, Ff = scenes.folded_cloth(60)
    out.append(("folded", Vf0, Vf1, Ef, Ff, 0.0))
    # a mesh pressed flat onto itself: every nearby pair is in (resting) contact
    Vc0, Vc1, Ec, Fc = scenes.folded_cloth(30)
    Vc1 = Vc0.copy()
    Vc1[:, 2] *= 0.0
    out.append(("flattened", Vc0, Vc1, Ec, Fc, 0.0))
    return out


# THE CULL'S BOUND DEPENDS ON THE TOLERANCE AND ON THE SCENE'S SCALE (narrow_cull.inc: rho from co_domain_tolerance / 3 and the lengths
# L_k, Condition 4 from 2^-52 L, the numerical error from the cube of the coordinates): every claim about it is tested over this grid
# (until round 6 only at 1e-6 and the scenes' own scale).  _scaled: the scene and its minimum separation times `scale`.
_CULL_GRID = [(tol, scale) for scale in (1.0, 1e3) for tol in (1e-3, 1e-6, 1e-9, 1e-12)]
# (whole-step calls on these do not finish in the ORACLE within a test's patience -- a minimum-separation shell or a slow scene's
# resting contacts under a tolerance many orders below it: hours of bisection for any traversal, the reference's included; the
# cull's own claim is still tested on them below, pair by pair)
# (name, tolerance, scale): orc.ccd with 8 threads does not return within 40 s on 8 cores (tools: the grid was timed once, round 6)
_CCD_TOO_DEEP = {
    ("cloth_ball_ms", 1e-9, 1.0), ("cloth_ball_ms", 1e-12, 1.0), ("very_slow_ms", 1e-12, 1.0), ("scaled_1e3", 1e-12, 1.0),
    ("cloth_ball", 1e-12, 1e3), ("cloth_ball_ms", 1e-6, 1e3), ("cloth_ball_ms", 1e-9, 1e3), ("cloth_ball_ms", 1e-12, 1e3),
    ("very_slow_ms", 1e-9, 1e3), ("very_slow_ms", 1e-12, 1e3), ("scaled_1e3", 1e-6, 1e3), ("scaled_1e3", 1e-9, 1e3), ("scaled_1e3", 1e-12, 1e3),
}


class _oracle_budget:
    """with _oracle_budget(orc, domains): the oracle's level order gives up (MemoryError) at that many live domains instead of 1.9 GB worth
    -- the grid's far corners hold pair sets it cannot decide, and it should say so in milliseconds, not after filling memory."""

    def __init__(self, orc, domains):
        self.orc, self.domains = orc, domains

    def __enter__(self):
        self.orc.lib().orc_set_level_budget(C.c_int64(self.domains))

    def __exit__(self, *a):
        self.orc.lib().orc_set_level_budget(C.c_int64(0))


def _scaled(case, scale):
    name, V0, V1, E, F, ms = _cull_scenes()[case]
    return name, V0 * scale, V1 * scale, E, F, ms * scale


@pytest.mark.parametrize("tol,scale", _CULL_GRID)
@pytest.mark.parametrize("case", range(10))
def test_projection_cull_changes_no_result(sccd, orc, case, tol, scale):
    """ccd() with the cull (the default) and without it, against the oracle: the same TOI bit for bit, on every scene, both zero-TOI
    policies, tolerances from 1e-3 to 1e-12, the scene's own scale and a thousand times that -- and on the ordinary scenes the cull
    really removes pairs."""
    name, V0, V1, E, F, ms = _scaled(case, scale)
    if (name, tol, scale) in _CCD_TOO_DEEP:
        pytest.skip("the oracle's own bisection of this scene does not end at this tolerance (see _CCD_TOO_DEEP)")
    c = sccd.Context(0)
    try:
        c.set_option(sccd.OPT_TOI_GUESS, 0)
        mesh = sccd.Mesh(V0, V1, E, F, ctx=c)
        for allow_zero in (True, False):
            want = orc.ccd(V0, V1, E, F, ms, -1, tol, allow_zero, nthreads=8)[0]
            c.set_option(sccd.OPT_CULL, 2)  # (forced: the default, 1, leaves meshes this small alone)
            got, st = sccd.ccd_mesh(mesh, ms, -1, tol, allow_zero, want_stats=True)
            c.set_option(sccd.OPT_CULL, 0)
            plain, st0 = sccd.ccd_mesh(mesh, ms, -1, tol, allow_zero, want_stats=True)
            assert got == want and plain == want, (name, tol, scale, allow_zero, got, plain, want)
            assert st0["n_vf_culled"] == 0 and st0["n_ee_culled"] == 0
            assert (st["n_vf_pairs"], st["n_ee_pairs"]) == (st0["n_vf_pairs"], st0["n_ee_pairs"])  # every overlap still counts as a query
            if name in ("cloth_ball", "soup", "folded") and tol <= 1e-6:
                assert st["n_vf_culled"] + st["n_ee_culled"] > 0.3 * (st["n_vf_pairs"] + st["n_ee_pairs"]), (name, tol, scale, st)
        mesh.close()
    finally:
        c.close()


@pytest.mark.parametrize("tol,scale", _CULL_GRID)
@pytest.mark.parametrize("case", range(10))
def test_culled_queries_have_no_impact_in_the_oracle(sccd, ctx, orc, case, tol, scale):
    """The claim itself: a pair the cull drops has NO accepted domain in the reference's bisection -- the oracle's per-query output
    (every query bisected on its own, pruned by nothing but its own earliest impact, root_finder.cu:297) reports no impact for it,
    under either zero-TOI policy, at every tolerance and scale of the grid.  And the kept pairs are a subset of the list, each once."""
    name, V0, V1, E, F, ms = _scaled(case, scale)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F, ms)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    undecided, decided = [], 0
    for is_vf, pairs in ((True, orc.sort_and_sweep(vb, fb, nthreads=8)[0]), (False, orc.sort_and_sweep(eb, nthreads=8)[0])):
        pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
        if len(pairs) == 0:
            continue
        kept = sccd.query_cull(mesh, pairs, is_vf, ms, tol)
        key = lambda p: p[:, 0].astype(np.int64) << 32 | p[:, 1].astype(np.int64)  # noqa: E731
        kk, ka = np.sort(key(kept)), np.sort(key(pairs))
        assert len(np.unique(kk)) == len(kk) and np.isin(kk, ka).all(), name
        culled = pairs[~np.isin(key(pairs), kk)]
        if len(culled) == 0:
            continue
        default_point = (tol, scale) == (1e-6, 1.0)
        for allow_zero, arith in ((True, 1), (False, 1), (True, 0)) if default_point else ((True, 1), (False, 0)):
            try:
                with _oracle_budget(orc, 0 if default_point else 1 << 21):
                    _, per_query, _ = orc.narrow_phase(V0, V1, E, F, culled, is_vf, ms=ms, tol=tol, allow_zero_toi=allow_zero, per_query=True, arith=arith)
            except MemoryError:  # (the ORACLE's level order outgrows its budget: pairs just beyond a minimum-separation shell under a tolerance far below it)
                undecided.append((is_vf, allow_zero, arith))
                continue
            decided += 1
            assert np.all(np.isinf(per_query)), (name, tol, scale, is_vf, allow_zero, arith, int(np.isfinite(per_query).sum()), len(culled))
    mesh.close()
    assert not (undecided and tol >= 1e-6 and scale == 1.0), (name, undecided)  # (at ordinary tolerances the oracle decides every scene)
    if undecided and not decided:
        pytest.skip("the oracle's level-order bisection of the culled pairs outgrows its memory budget at this tolerance and scale")


# the float build's cull (round 6): float filter constants, Condition 4 one FLOAT ulp wide, vertices rounded to float first
_CULL_GRID_F32 = [(1e-6, 1.0), (1e-3, 1.0), (1e-5, 1.0), (1e-4, 1e3)]


@pytest.mark.parametrize("tol,scale", _CULL_GRID_F32)
@pytest.mark.parametrize("case", range(10))
def test_float_build_cull_changes_no_result_and_drops_no_impact(sccd, orc, case, tol, scale):
    """SCCD_OPT_SCALAR = 1 with the projection cull forced and off, against the oracle's float twin: the same TOI bit for bit under both
    zero-TOI policies; and the claim itself -- no pair the cull drops (whole step, and the slabs [0, 0.5] / [0.5, 1]) has an impact
    in the float twin's per-query output inside the slab."""
    name, V0, V1, E, F, ms = _scaled(case, scale)
    c = sccd.Context(0)
    try:
        c.set_option(sccd.OPT_TOI_GUESS, 0)
        c.set_option(sccd.OPT_SCALAR, 1)
        mesh = sccd.Mesh(V0, V1, E, F, ctx=c)
        culled_total = 0
        for allow_zero in (True, False):
            want = orc.ccd(V0, V1, E, F, ms, -1, tol, allow_zero, nthreads=8, scalar="f32")[0]
            c.set_option(sccd.OPT_CULL, 2)
            got, st = sccd.ccd_mesh(mesh, ms, -1, tol, allow_zero, want_stats=True)
            c.set_option(sccd.OPT_CULL, 0)
            plain, st0 = sccd.ccd_mesh(mesh, ms, -1, tol, allow_zero, want_stats=True)
            assert got == want and plain == want, (name, tol, scale, allow_zero, got, plain, want)
            assert (st["n_vf_pairs"], st["n_ee_pairs"]) == (st0["n_vf_pairs"], st0["n_ee_pairs"]) and st0["n_vf_culled"] + st0["n_ee_culled"] == 0
            culled_total += st["n_vf_culled"] + st["n_ee_culled"]
        if name in ("cloth_ball", "folded") and scale == 1.0:
            assert culled_total > 0, name  # (the float build's cull does remove pairs on ordinary scenes)
        # pair by pair (float boxes: the float build's broad phase)
        vb, eb, fb = orc.build_boxes(V0, V1, E, F, ms, scalar="f32")
        key = lambda p: p[:, 0].astype(np.int64) << 32 | p[:, 1].astype(np.int64)  # noqa: E731
        for is_vf, pairs in ((True, orc.sort_and_sweep(vb, fb, nthreads=8)[0]), (False, orc.sort_and_sweep(eb, nthreads=8)[0])):
            pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
            if len(pairs) == 0:
                continue
            ka = key(pairs)
            slabs = ((0.0, 1.0), (0.0, 0.5), (0.5, 1.0))
            gone = [~np.isin(ka, key(sccd.query_cull_slab(mesh, pairs, is_vf, ms, tol, t_lo, t_hi))) for t_lo, t_hi in slabs]
            some = np.logical_or.reduce(gone)
            if not some.any():
                continue
            try:
                with _oracle_budget(orc, 1 << 21):
                    pq = np.full(len(pairs), np.inf)
                    pq[some] = orc.narrow_phase(V0, V1, E, F, pairs[some], is_vf, ms=ms, tol=tol, allow_zero_toi=True, per_query=True, scalar="f32")[1]
            except MemoryError:
                continue
            for (t_lo, t_hi), g in zip(slabs, gone):
                bad = g & (pq >= t_lo) & (pq < t_hi)
                assert not bad.any(), (name, tol, scale, is_vf, t_lo, t_hi, int(bad.sum()), pq[bad][:4])
        mesh.close()
    finally:
        c.close()


@pytest.mark.parametrize("tol,scale", _CULL_GRID)
@pytest.mark.parametrize("case", range(10))
def test_slab_culls_drop_no_query_with_an_impact_in_their_slab(sccd, ctx, orc, case, tol, scale):
    """The cull per slab of time (narrow_cull.inc, "slabs of time"; sccd_query_cull_slab): a pair that is dropped for [t_lo, t_hi] has
    no earliest impact inside [t_lo, t_hi) in the oracle's per-query output -- the slabs ccd() uses ([0, 0.5] and [0.5, 1] for the
    two launches of a start from 1, [0, b] and [0.5, b] for a start from a bound) and a few others.  And the whole step is the
    slab (0, 1): the same list as sccd_query_cull's."""
    name, V0, V1, E, F, ms = _scaled(case, scale)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F, ms)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    key = lambda p: p[:, 0].astype(np.int64) << 32 | p[:, 1].astype(np.int64)  # noqa: E731
    dropped_somewhere = 0
    undecided, decided = [], 0
    for is_vf, pairs in ((True, orc.sort_and_sweep(vb, fb, nthreads=8)[0]), (False, orc.sort_and_sweep(eb, nthreads=8)[0])):
        pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
        if len(pairs) == 0:
            continue
        ka = key(pairs)
        whole = np.sort(key(sccd.query_cull(mesh, pairs, is_vf, ms, tol)))
        assert np.array_equal(np.sort(key(sccd.query_cull_slab(mesh, pairs, is_vf, ms, tol, 0.0, 1.0))), whole), name
        default_point = (tol, scale) == (1e-6, 1.0)
        slabs = ((0.0, 0.5), (0.5, 1.0), (0.0, 0.3), (0.5, 0.77), (0.25, 0.75), (0.0, 1e-3), (0.999, 1.0)) if default_point else ((0.0, 0.5), (0.5, 1.0), (0.0, 0.3), (0.5, 0.77))
        gone = []
        for t_lo, t_hi in slabs:
            kk = np.sort(key(sccd.query_cull_slab(mesh, pairs, is_vf, ms, tol, t_lo, t_hi)))
            assert len(np.unique(kk)) == len(kk) and np.isin(kk, ka).all(), (name, t_lo, t_hi)
            gone.append(~np.isin(ka, kk))
        # (the oracle bisects the pairs that were dropped for some slab: the ones kept everywhere include the resting contacts, whose
        # level-order bisection outgrows its memory budget on the shifted scene)
        some = np.logical_or.reduce(gone)
        dropped_somewhere += int(some.sum())
        if not some.any():
            continue
        for az, ar in ((True, 1), (False, 1), (True, 0)) if default_point else ((True, 1), (False, 0)):
            pq = np.full(len(pairs), np.inf)
            try:
                with _oracle_budget(orc, 0 if default_point else 1 << 21):
                    pq[some] = orc.narrow_phase(V0, V1, E, F, pairs[some], is_vf, ms=ms, tol=tol, allow_zero_toi=az, per_query=True, arith=ar)[1]
            except MemoryError:  # (the ORACLE's level order outgrows its budget: resting contacts among the pairs some slab dropped, under a tiny tolerance)
                undecided.append((is_vf, az, ar))
                continue
            decided += 1
            assert not np.isnan(pq[some]).any(), (name, tol, scale)  # (NaN: the oracle's level order gave up -- nothing would have been compared)
            for (t_lo, t_hi), g in zip(slabs, gone):
                bad = g & (pq >= t_lo) & (pq < t_hi)
                assert not bad.any(), (name, tol, scale, is_vf, az, ar, t_lo, t_hi, int(bad.sum()), pq[bad][:4])
    mesh.close()
    if name in ("cloth_ball", "folded", "soup") and tol <= 1e-6:
        assert dropped_somewhere > 0, (name, tol, scale)
    assert not (undecided and tol >= 1e-6 and scale == 1.0), (name, undecided)  # (at ordinary tolerances the oracle decides every scene)
    if undecided and not decided:
        pytest.skip("the oracle's level-order bisection of the dropped pairs outgrows its memory budget at this tolerance and scale")


def _condition4_pair(delta=5e-6):
    """A long, nearly static pair of edges under a tolerance far below double resolution: edge a spans 800 units along the diagonal of
    the plane z = 0, edge b (length 1) stands on that plane, upright, `delta` beside a's line, drifting 1e-6 along a's direction (the
    common normal of the two edges is the stand-off's direction: one of the cull's four).  With the
    reference's edge-edge tolerances (root_finder.cu:82-87: tol_u REUSES tol_t, tol_w comes from L_u) and co_domain_tolerance = 1e-15,
    tol_w = 4.2e-19 lies far below one ulp of w: Condition 1 is out of reach and the bisection ends in Condition 4, on domains whose
    u interval is still 2^-24 wide -- an image box 4.8e-5 wide, a hundred times Condition 1's bound."""
    n = np.array([1.0, -1.0, 0.0]) / np.sqrt(2.0)
    a0, a1 = np.array([-400.0, -400.0, 0.0]), np.array([400.0, 400.0, 0.0])
    P = np.array([0.37, 0.37, 0.0])
    b1 = P + delta * n  # (w = 1 is the end on the plane: one ulp of w is 2^-53 there)
    b0 = b1 + np.array([0.0, 0.0, 1.0])
    V0 = np.array([a0, a1, b0, b1])
    V1 = V0.copy()
    V1[2:] += 1e-6 * np.array([1.0, 1.0, 0.0])
    return V0, V1, np.array([[0, 1], [2, 3]], np.int32), np.zeros((0, 3), np.int32)


def test_cull_keeps_a_query_the_reference_accepts_by_condition_4(sccd, ctx, orc):
    """VERDICT r05's constructed case.  The domain [0, 2^-24] x [k 2^-24, (k + 1) 2^-24] x [1 - 2^-53, 1] around the closest approach
    passes the reference's inclusion test (the oracle's, root_finder.cu:157-198, evaluated on exactly that domain), is no Condition-1,
    -2 or -3 domain, its split dimension is w (the largest width / tolerance, :200-211) and w cannot be halved: Condition 4 (:222-225,
    :362) ACCEPTS it -- at a stand-off of 5e-6, three times beyond the reach the cull allowed for until round 6 (2 e_max + tol_u L_u +
    slack = 1.07e-6).  The cull must keep the pair -- round 5's bound dropped it (gpurun_out of round 6: the same test on a library built
    with -DCULL_NO_CONDITION4 fails at the first query_cull).  (Nobody bisects this query to the end: the quirk's tolerances make it
    2^60 nodes for any traversal -- the single domain is the evidence.)"""
    co = 1e-15
    V0, V1, E, F = _condition4_pair()
    v24 = np.concatenate([V0.ravel(), V1.ravel()])
    tol3, err3 = orc.query_constants(v24, False, False, co)
    assert tol3[2] < 2.0 ** -53 and tol3[0] == tol3[1] > 2.0 ** -40
    k = int(np.floor((0.37 + 400.0) / 800.0 * 2 ** 24))
    dom = np.array([0.0, 2.0 ** -24, k / 2 ** 24, (k + 1) / 2 ** 24, 1.0 - 2.0 ** -53, 1.0])
    inside, true_tol, box_in = orc.inclusion(v24, dom, err3, 0.0, False)
    assert inside and not box_in and true_tol > 1e-5 > co  # (not rejected; neither Condition 2 nor 3)
    width = dom[1::2] - dom[0::2]
    assert (width > tol3).all()  # (nor Condition 1)
    ratio = width / tol3
    assert ratio[2] > ratio[0] and ratio[2] > ratio[1]  # split_dimension = w ...
    mid = (dom[4] + dom[5]) / 2
    assert mid <= dom[4] or mid >= dom[5]  # ... which cannot be halved: Condition 4 accepts
    assert 5e-6 > 3 * (2 * err3[0] + tol3[1] * 800.0 + 1e-13 * 400.0)  # beyond Condition 1's reach, comfortably
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    try:
        pair = np.array([[0, 1]], np.int32)
        assert len(sccd.query_cull(mesh, pair, False, 0.0, co)) == 1
        for t_lo, t_hi in ((0.0, 0.5), (0.0, 1e-3), (0.0, 1.0)):
            assert len(sccd.query_cull_slab(mesh, pair, False, 0.0, co, t_lo, t_hi)) == 1
        assert len(sccd.query_cull(mesh, pair, False, 0.0, 1e-6)) == 1  # (1e-6: tol_u = 1/3, Condition 1 alone reaches 267 units -- kept)
        far = _condition4_pair(delta=1e-2)
        mesh.update_vertices(far[0], far[1])
        assert len(sccd.query_cull(mesh, pair, False, 0.0, co)) == 0  # (and the bound is no blanket keep: 1e-2 away is culled)
    finally:
        mesh.close()


def test_a_list_of_queries_survives_the_fallback_that_reads_it(sccd, orc):
    """Soak seed 60120 (round 6): the float build's walk kernel LISTS the queries it cannot finish, in a buffer sized for this call;
    the fallback that redoes them asked for a larger buffer before it had read the list -- a grow-only buffer does not keep its
    contents -- and redid whatever the new allocation held: the first float call on a context whose earlier calls had left the
    buffer small returned a later TOI (or faulted), the next one, buffer grown, was right.  A fresh context, one double call, then
    the float build at a tolerance of 1e-6 of the scene: the oracle's float twin on the first call."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import soak

    V0, V1, E, F, kind, scale, shift, ms, allow_zero, arith, *_ = soak.scene_of(60120)
    tol = 1e-6 * scale
    want = orc.ccd(V0, V1, E, F, ms, -1, tol, allow_zero, arith=arith, nthreads=8, scalar="f32")[0]
    for apart in (0, 1):
        c = sccd.Context(0)
        try:
            c.set_option(sccd.OPT_ARITH, arith)
            c.set_option(sccd.OPT_PASSES_APART, apart)
            mesh = sccd.Mesh(V0, V1, E, F, ctx=c)
            assert sccd.ccd_mesh(mesh, ms, -1, 1e-6, allow_zero) == orc.ccd(V0, V1, E, F, ms, -1, 1e-6, allow_zero, arith=arith, nthreads=8)[0]
            c.set_option(sccd.OPT_SCALAR, 1)
            for _ in range(2):
                assert sccd.ccd_mesh(mesh, ms, -1, tol, allow_zero) == want
            mesh.close()
        finally:
            c.close()


def test_float_build_is_bounded_on_a_query_that_explodes(sccd, ctx):
    """Soak seed 500388 (a small cloth on a ball, scaled by 104, minimum separation 0.31): in float Condition 1 is out of reach, so
    the queries in resting contact are bisected down to single ulps -- 182 s for the oracle's float twin on 8 cores, minutes for ONE
    lane of the float build's depth-first kernels before they had check budgets.  With the budgets the lanes hand such queries on
    and level order finishes the call: the oracle twin's value (recorded from that 182 s run), within seconds."""
    import sys
    import time

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import soak

    V0, V1, E, F, kind, scale, shift, ms, allow_zero, arith, *_ = soak.scene_of(500388)
    assert (len(F), arith, allow_zero) == (370, 0, True)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    try:
        ctx.set_option(sccd.OPT_ARITH, arith)
        assert sccd.ccd_mesh(mesh, ms, -1, 1e-6, allow_zero) == float.fromhex("0x1.e125ec0000000p-3")  # the double build
        ctx.set_option(sccd.OPT_SCALAR, 1)
        t0 = time.perf_counter()
        got = sccd.ccd_mesh(mesh, ms, -1, 1e-6, allow_zero)
        dt = time.perf_counter() - t0
        assert got == float.fromhex("0x1.bf121a0000000p-3"), got
        assert dt < 10.0, dt
    finally:
        ctx.set_option(sccd.OPT_SCALAR, 0)
        ctx.set_option(sccd.OPT_ARITH, sccd.ARITH_DEFAULT)
        mesh.close()
