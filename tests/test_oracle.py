"""The CPU oracle against independent checks: brute force pair sets, analytic times of impact,
libm nextafter semantics, and the committed golden vectors (tests/golden/, made by
tests/golden/make_golden.py from the oracle itself -- parity vs the reference is UNPINNED, see
oracle/sccd_oracle.h)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from sccd import scenes

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_vertex_box_is_conservative_one_ulp(orc):
    # aabb.cpp:31-36: min = nextafter_down(p) - nextafter_up(r), max = nextafter_up(p) + nextafter_up(r)
    V0 = np.array([[0.0, 1.0, -2.5], [1e-300, -1e300, 3.0]])
    V1 = np.array([[0.5, 1.0, -3.0], [0.0, 1e300, 3.0]])
    E = np.zeros((0, 2), np.int32)
    F = np.zeros((0, 3), np.int32)
    vb, _, _ = orc.build_boxes(V0, V1, E, F, 0.0)
    tiny = 5e-324
    for i in range(2):
        for k in range(3):
            lo = min(np.nextafter(V0[i, k], -np.inf) - tiny, np.nextafter(V1[i, k], -np.inf) - tiny)
            hi = max(np.nextafter(V0[i, k], np.inf) + tiny, np.nextafter(V1[i, k], np.inf) + tiny)
            assert vb["min"][i, k] == lo and vb["max"][i, k] == hi
    assert list(vb["vertex_ids"][1]) == [1, -2, -2] and vb["element_id"][1] == 1
    vb2, _, _ = orc.build_boxes(V0, V1, E, F, 0.25)
    r = np.nextafter(0.25, np.inf)
    assert vb2["min"][0, 0] == np.nextafter(0.0, -np.inf) - r
    assert vb2["max"][0, 0] == np.nextafter(0.5, np.inf) + r


def test_element_box_ids(orc):
    V0, V1, E, F = scenes.cloth_ball(6, 0)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    assert np.array_equal(eb["vertex_ids"][:, 0], E[:, 0]) and np.array_equal(eb["vertex_ids"][:, 2], -E[:, 0] - 1)
    assert np.array_equal(fb["vertex_ids"], F)
    assert np.array_equal(fb["min"], np.minimum(np.minimum(vb["min"][F[:, 0]], vb["min"][F[:, 1]]), vb["min"][F[:, 2]]))


@pytest.mark.parametrize("scene", ["cloth_ball", "soup", "random"])
def test_sweep_equals_brute_force(orc, scene):
    if scene == "random":
        boxes = scenes.random_boxes(3000, seed=5, max_extent=0.08)
        sw, ax, tests = orc.sort_and_sweep(boxes)
        assert np.array_equal(sw, orc.brute_force(boxes)) and len(sw) > 1000
        # the pair set does not depend on the sort axis
        for axis in (1, 2):
            assert np.array_equal(orc.sort_and_sweep(boxes, sort_axis=axis)[0], sw)
        return
    V0, V1, E, F = scenes.cloth_ball(24, 1) if scene == "cloth_ball" else scenes.triangle_soup(300, seed=2)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    vf, _, _ = orc.sort_and_sweep(vb, fb)
    ee, _, _ = orc.sort_and_sweep(eb)
    assert np.array_equal(vf, orc.brute_force(vb, fb)) and len(vf) > 0
    assert np.array_equal(ee, orc.brute_force(eb)) and len(ee) > 0
    assert np.array_equal(orc.sort_and_sweep(vb, fb, nthreads=4)[0], vf)


def test_next_sort_axis_is_argmax_variance(orc):
    b = scenes.random_boxes(500, seed=9, z_scale=0.01)
    b["min"][:, 1] *= 3
    b["max"][:, 1] = b["min"][:, 1] + 0.01
    _, ax, _ = orc.sort_and_sweep(b)
    assert ax == 1  # sort_and_sweep.cpp:176-195


def test_empty_and_single(orc):
    e = np.zeros(0, scenes.AABB_DTYPE)
    one = scenes.random_boxes(1)
    assert len(orc.sort_and_sweep(e)[0]) == 0 and len(orc.sort_and_sweep(one)[0]) == 0
    assert len(orc.sort_and_sweep(e, one)[0]) == 0 and len(orc.sort_and_sweep(one, e)[0]) == 0


# ---- analytic known-answer times of impact --------------------------------------------------
def _vf_scene(z0, z1, tri_z=0.0, xy=(0.25, 0.25)):
    """vertex moving along z over a static unit right triangle"""
    V0 = np.array([[xy[0], xy[1], z0], [0, 0, tri_z], [1, 0, tri_z], [0, 1, tri_z]], float)
    V1 = V0.copy()
    V1[0, 2] = z1
    F = np.array([[1, 2, 3]], np.int32)
    E = np.array([[1, 2], [2, 3], [1, 3]], np.int32)
    return V0, V1, E, F


@pytest.mark.parametrize("arith", [0, 1])
def test_vertex_hits_static_triangle(orc, arith):
    V0, V1, E, F = _vf_scene(1.0, -1.0)  # crosses z = 0 at t* = 0.5
    toi, _, st = orc.narrow_phase(V0, V1, E, F, [[0, 0]], True, tol=1e-6, arith=arith)
    assert 0.5 - 1e-5 <= toi <= 0.5 and st["n_checks"] > 10
    V0, V1, E, F = _vf_scene(0.3, -0.9)  # t* = 0.25
    toi, _, _ = orc.narrow_phase(V0, V1, E, F, [[0, 0]], True, arith=arith)
    assert 0.25 - 1e-5 <= toi <= 0.25


def test_miss_returns_one(orc):
    V0, V1, E, F = _vf_scene(1.0, 0.5)  # never reaches the triangle
    assert orc.narrow_phase(V0, V1, E, F, [[0, 0]], True)[0] == 1.0
    V0, V1, E, F = _vf_scene(1.0, -1.0, xy=(2.0, 2.0))  # passes beside it
    assert orc.narrow_phase(V0, V1, E, F, [[0, 0]], True)[0] == 1.0


def _tilted_scene():
    """vertex dropping onto a TILTED triangle: with ms > 0 the inflated contact starts at a
    single (u,v) -- a face-on, axis-aligned contact would enter on a whole 2-D patch and make
    Tight-Inclusion (reference and restatement alike) refine ~1/tol^2 domains."""
    V0 = np.array([[0.25, 0.25, 1.0], [0, 0, 0], [1, 0, 0.3], [0, 1, 0.5]], float)
    V1 = V0.copy()
    V1[0, 2] = -1.0
    return V0, V1, np.array([[1, 2], [2, 3], [1, 3]], np.int32), np.array([[1, 2, 3]], np.int32)


def _sampled_entry_time(V0, V1, ms, nt=4001, nuv=301):
    """first t at which min_(u,v) |p(t) - T(u,v)|_inf <= ms, by dense sampling"""
    u, v = np.meshgrid(np.linspace(0, 1, nuv), np.linspace(0, 1, nuv))
    keep = (u + v) <= 1
    u, v = u[keep], v[keep]
    T = V0[1] + np.outer(u, V0[2] - V0[1]) + np.outer(v, V0[3] - V0[1])  # static triangle
    for t in np.linspace(0, 1, nt):
        p = V0[0] + t * (V1[0] - V0[0])
        if np.abs(p - T).max(axis=1).min() <= ms:
            return t
    return 1.0


def test_minimum_separation_hits_earlier(orc):
    V0, V1, E, F = _tilted_scene()
    t0 = orc.narrow_phase(V0, V1, E, F, [[0, 0]], True, ms=0.0)[0]
    t1 = orc.narrow_phase(V0, V1, E, F, [[0, 0]], True, ms=0.05)[0]
    assert 0.4 - 1e-5 <= t0 <= 0.4  # plane z = 0.3x + 0.5y is 0.2 under the vertex: 1 - 2t = 0.2
    assert abs(t1 - _sampled_entry_time(V0, V1, 0.05)) < 5e-3
    assert t1 < t0 - 0.01


def test_allow_zero_toi(orc):
    V0, V1, E, F = _vf_scene(0.0, -1.0)  # in contact at t = 0
    assert orc.narrow_phase(V0, V1, E, F, [[0, 0]], True, allow_zero_toi=True)[0] == 0.0
    t = orc.narrow_phase(V0, V1, E, F, [[0, 0]], True, allow_zero_toi=False)[0]
    assert 0.0 <= t <= 1e-5  # Condition 1 has no zero guard (root_finder.cu:322)


def test_edge_edge_crossing(orc):
    # edge a along x at height z(t) = 1 - 2t, edge b along y at z = 0: touch at t = 0.5
    V0 = np.array([[-1, 0, 1], [1, 0, 1], [0, -1, 0], [0, 1, 0]], float)
    V1 = V0.copy()
    V1[:2, 2] = -1
    E = np.array([[0, 1], [2, 3]], np.int32)
    F = np.zeros((0, 3), np.int32)
    toi, _, _ = orc.narrow_phase(V0, V1, E, F, [[0, 1]], False)
    assert 0.5 - 1e-5 <= toi <= 0.5


def test_static_query_has_infinite_tolerance(orc):
    v = np.zeros((8, 3))
    v[:4] = [[0.2, 0.2, 1], [0, 0, 0], [1, 0, 0], [0, 1, 0]]
    v[4:] = v[:4]
    tol, err = orc.query_constants(v, True, False, 1e-6)
    assert np.isinf(tol[0])  # nothing moves: co_domain_tol / (3 * 0) on the time axis
    assert np.allclose(tol[1:], 1e-6 / 3)  # unit edges
    assert np.allclose(err, 6.661338147750939e-15)  # max(1, |coords|)^3 * filter


def test_toi_in_is_an_upper_bound(orc):
    V0, V1, E, F = _vf_scene(1.0, -1.0)
    assert orc.narrow_phase(V0, V1, E, F, [[0, 0]], True, toi=0.25)[0] == 0.25
    assert orc.narrow_phase(V0, V1, E, F, [[0, 0]], True, toi=0.0)[0] == 0.0


def test_bfs_and_dfs_orders_agree(orc):
    V0, V1, E, F = scenes.triangle_soup(150, seed=4)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    vf, _, _ = orc.sort_and_sweep(vb, fb)
    ee, _, _ = orc.sort_and_sweep(eb)
    for arith in (0, 1):
        a, _, _ = orc.narrow_phase(V0, V1, E, F, vf, True, arith=arith)
        b, _ = orc.narrow_phase_mt(V0, V1, E, F, vf, True, arith=arith, nthreads=4)
        assert a == b and a < 1
        a2, _, _ = orc.narrow_phase(V0, V1, E, F, ee, False, toi=a, arith=arith)
        b2, _ = orc.narrow_phase_mt(V0, V1, E, F, ee, False, toi=b, arith=arith, nthreads=4)
        assert a2 == b2 <= a


def test_per_query_toi_min_is_global(orc):
    V0, V1, E, F = scenes.triangle_soup(80, seed=6)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    vf, _, _ = orc.sort_and_sweep(vb, fb)
    g, _, _ = orc.narrow_phase(V0, V1, E, F, vf, True)
    t, pq, _ = orc.narrow_phase(V0, V1, E, F, vf, True, per_query=True)
    assert t == g == min(1.0, pq.min())  # tests/test_narrow_phase.cu:60-62


def test_golden_vectors(orc):
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        G = json.load(f)
    from golden.make_golden import SLOW_CASES, compute_case

    for name, want in G.items():
        if name in SLOW_CASES or name == "contract_split":  # (its own test below: inputs and both answers are in the file)
            continue
        got = compute_case(orc, name)
        assert got == want, name


def test_level_order_gives_up_at_its_domain_budget(orc):
    """A vertex sliding inside the plane of a triangle with allow_zero_toi off: in level order one query keeps
    1.5 M live domains.  The restatement must report that it ran out of budget, not take the host's memory."""
    V0 = np.array([[0.2, 0.2, 0.0], [0, 0, 0], [1, 0, 0], [0, 1, 0]], float)
    V1 = V0.copy()
    V1[0] = [0.3, 0.25, 0.0]
    E = np.array([[1, 2], [2, 3], [1, 3]], np.int32)
    F = np.array([[1, 2, 3]], np.int32)
    orc.lib().orc_set_level_budget(C.c_int64(1 << 16))
    try:
        with pytest.raises(MemoryError):
            orc.narrow_phase(V0, V1, E, F, [[0, 0]], True, allow_zero_toi=False, per_query=True)
    finally:
        orc.lib().orc_set_level_budget(C.c_int64(0))
    toi, _, st = orc.narrow_phase(V0, V1, E, F, [[0, 0]], True, allow_zero_toi=False, per_query=True)
    assert toi == 0.0 and st["max_queue"] > 1 << 20


# ---- SCALABLE_CCD_USE_DOUBLE = OFF twin (Scalar = float) ----------------------------------------
def test_f32_boxes_are_float_valued_and_conservative(orc):
    """aabb.cpp:43-47: vertices are cast to float FIRST, then inflated with nextafterf -- the float box still
    contains the double box of the same vertex, and every coordinate is a float."""
    V0, V1, E, F = scenes.cloth_ball(12, 1, seed=4)
    for r in (0.0, 1e-3):
        v64, e64, f64 = orc.build_boxes(V0, V1, E, F, r)
        v32, e32, f32 = orc.build_boxes(V0, V1, E, F, r, scalar="f32")
        for b64, b32 in ((v64, v32), (e64, e32), (f64, f32)):
            assert np.array_equal(b32["min"], b32["min"].astype(np.float32).astype(np.float64))
            assert np.array_equal(b32["max"], b32["max"].astype(np.float32).astype(np.float64))
            assert np.all(b32["min"] <= b64["min"] + 1e-9) and np.all(b32["max"] >= b64["max"] - 1e-9)
            assert np.array_equal(b32["vertex_ids"], b64["vertex_ids"])
    # one vertex by hand: p = 0.1 (not a float), r = 0
    vb, _, _ = orc.build_boxes(np.array([[0.1, 0.1, 0.1]]), np.array([[0.1, 0.1, 0.1]]), np.zeros((0, 2), np.int32),
                               np.zeros((0, 3), np.int32), 0.0, scalar="f32")
    p = np.float32(0.1)
    tiny = np.nextafter(np.float32(0), np.float32(1))
    assert vb["min"][0, 0] == float(np.float32(np.nextafter(p, np.float32(-np.inf)) - tiny))
    assert vb["max"][0, 0] == float(np.float32(np.nextafter(p, np.float32(np.inf)) + tiny))


@pytest.mark.parametrize("arith", [0, 1])
def test_f32_narrow_phase_known_answers(orc, arith):
    V0 = np.array([[0.25, 0.25, 1.0], [0, 0, 0], [1, 0, 0], [0, 1, 0]], float)
    V1 = V0.copy()
    V1[0, 2] = -1.0
    E = np.array([[1, 2], [2, 3], [1, 3]], np.int32)
    F = np.array([[1, 2, 3]], np.int32)
    toi, _, st = orc.narrow_phase(V0, V1, E, F, [[0, 0]], True, arith=arith, scalar="f32")
    assert 0.5 - 1e-4 <= toi <= 0.5 and toi == float(np.float32(toi))  # a float, within the float tolerance of t* = 0.5
    assert st["n_checks"] > 10
    assert orc.narrow_phase(V0, V1, E, F, [[0, 0]], True, toi=0.25, scalar="f32")[0] == 0.25
    V1[0, 2] = 0.5  # never reaches the triangle
    assert orc.narrow_phase(V0, V1, E, F, [[0, 0]], True, scalar="f32")[0] == 1.0
    # per-query constants: err = max(1, |coords|)^3 * the FLOAT filter (root_finder.cu:103-119)
    v = np.arange(24, dtype=np.float32).reshape(8, 3) * np.float32(0.25)
    tol, err = np.zeros(3, np.float32), np.zeros(3, np.float32)
    orc.lib().orc_query_constants_f32(v.ctypes.data_as(C.c_void_p), C.c_int(1), C.c_int(0), C.c_float(1e-6),
                                      tol.ctypes.data_as(C.c_void_p), err.ctypes.data_as(C.c_void_p))
    m = np.maximum(np.float32(1), np.abs(v).max(axis=0))
    assert np.array_equal(err, (m * m * m * np.float32(3.576279e-06)).astype(np.float32))


def test_f32_traversal_orders_agree_and_track_the_double_result(orc):
    """Level order and depth-first with a shared minimum give the same float TOI bit for bit; the float result
    stays within the float tolerance of the double one; float boxes can only add candidate pairs."""
    for V0, V1, E, F in (scenes.cloth_ball(20, 1, seed=3), scenes.triangle_soup(150, seed=9, size=0.12, motion=0.3)):
        t64, nvf64, nee64 = orc.ccd(V0, V1, E, F, nthreads=4)
        t_a, nvf, nee = orc.ccd(V0, V1, E, F, scalar="f32", nthreads=1)
        t_b, nvf_b, nee_b = orc.ccd(V0, V1, E, F, scalar="f32", nthreads=4)
        assert t_a == t_b and (nvf, nee) == (nvf_b, nee_b)
        assert nvf >= nvf64 and nee >= nee64
        assert abs(t_a - t64) < 1e-3 and t_a == float(np.float32(t_a))


def test_golden_case_that_tells_the_arithmetic_contracts_apart():
    """tests/golden/make_contract_case.py placed a minimum separation where the strict and the fused evaluation of
    root_finder.cu:137-198 decide differently: the two expected values differ, and the oracle reproduces each."""
    import json

    import orc

    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.json")))["contract_split"]
    V0 = np.array([[float.fromhex(x) for x in r] for r in G["V0"]])
    V1 = np.array([[float.fromhex(x) for x in r] for r in G["V1"]])
    E, F, ms = np.array(G["E"], np.int32), np.array(G["F"], np.int32), float.fromhex(G["ms"])
    want = {0: float.fromhex(G["toi_strict"]), 1: float.fromhex(G["toi_fma"])}
    assert want[0] != want[1]
    for arith in (0, 1):
        assert orc.ccd(V0, V1, E, F, ms, -1, 1e-6, True, arith=arith)[0] == want[arith]
