"""The numbers the reference's OWN tests pin, checked the day the reference's sample data is present.

The reference clones Sample-Scalable-CCD-Data@f24a3b15 at configure time (cmake/recipes/sample_data.cmake:25-35); this image
has no network, so every test here SKIPS unless SCCD_SAMPLE_DATA_DIR points at a checkout of that repository
(`SCCD_SAMPLE_DATA_DIR=/path/to/Sample-Scalable-CCD-Data pytest tests/test_reference_constants.py`).  With the data present
they pin BOTH the CPU oracle (the `-m "not gpu"` half) and the HIP library (the `-m gpu` half) to the reference:

* tests/test_broad_phase.cpp:36-38   cloth-ball 92 -> 93: 46,598 vertex / 138,825 edge / 92,230 face boxes
* tests/test_broad_phase.cpp:49,54   next sort axis 0 for both lists
* tests/test_broad_phase.cpp:62-63   1,655,541 vertex-face and 5,197,332 edge-edge overlaps (inflation radius 0: tests/io.cpp:35)
* tests/test_broad_phase.cpp:66-77, tests/ground_truth.cpp:27-64   every ground-truth pair is among the overlaps, on the
  five scenes of tests/test_broad_phase.cu:31-65
* tests/test_narrow_phase.cu:41-45,65   ccd(ms = 0, max_iter = -1, tol = 1e-6, allow_zero_toi) == 3.814697265625e-06 = 2^-18,
  and (TOI_PER_QUERY, :60-62) no query's time of impact is earlier than that
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "scalable-ccd_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

from sccd import io as sio  # noqa: E402

DATA = os.environ.get("SCCD_SAMPLE_DATA_DIR", "")

# tests/test_broad_phase.cu:31-65 (scene directory, frame t0, frame t1, ground-truth prefix)
SCENES = {
    "armadillo-rollers": ("armadillo-rollers", "326.ply", "327.ply", "326"),
    "cloth-ball": ("cloth-ball", "cloth_ball92.ply", "cloth_ball93.ply", "92"),
    "cloth-funnel": ("cloth-funnel", "227.ply", "228.ply", "227"),
    "n-body": ("n-body-simulation", "balls16_18.ply", "balls16_19.ply", "18"),
    "rod-twist": ("rod-twist", "3036.ply", "3037.ply", "3036"),
}
# tests/test_broad_phase.cpp:36-38,62-63 and tests/test_narrow_phase.cu:65
CLOTH_BALL = {"nV": 46_598, "nE": 138_825, "nF": 92_230, "n_vf": 1_655_541, "n_ee": 5_197_332, "toi": 3.814697265625e-06}


def _paths(name):
    d, t0, t1, gt = SCENES[name]
    base = os.path.join(DATA, d)
    return (os.path.join(base, "frames", t0), os.path.join(base, "frames", t1), os.path.join(base, "boxes", gt + "vf.json"),
            os.path.join(base, "boxes", gt + "ee.json"))


def _need(name):
    if not DATA:
        pytest.skip("SCCD_SAMPLE_DATA_DIR is not set: the reference's sample data (Sample-Scalable-CCD-Data) is not in this image")
    for p in _paths(name):
        if not os.path.exists(p):
            pytest.skip(f"{p} is missing")


def _load(name):
    t0, t1, _, _ = _paths(name)
    return sio.parse_mesh(t0, t1)  # V0, V1, E (igl::edges order), F


def _superset(vf, ee, nV, nE, name):
    """tests/test_broad_phase.cpp:66-77: offset the ids the way the ground truth was generated, then no pair may be missing"""
    _, _, gt_vf, gt_ee = _paths(name)
    vf_g, ee_g = sio.offset_for_ground_truth(vf, ee, nV, nE)
    assert len(sio.missing_from(vf_g, sio.read_ground_truth(gt_vf))) == 0
    assert len(sio.missing_from(ee_g, sio.read_ground_truth(gt_ee))) == 0


def test_the_constants_are_the_reference_tests_own():
    """(always runs) the numbers above are the ones in the reference's test sources, not a transcription error: where
    /root/reference is present (this container, never the GPU box) they are looked up in the files."""
    ref = "/root/reference/tests"
    if not os.path.isdir(ref):
        pytest.skip("the reference tree is not on this machine")
    cpp = open(os.path.join(ref, "test_broad_phase.cpp")).read()
    for text in ("46'598", "138'825", "92'230", "1'655'541", "5'197'332", "sort_axis == 0"):
        assert text in cpp
    assert "3.814697265625e-06" in open(os.path.join(ref, "test_narrow_phase.cu")).read()
    assert CLOTH_BALL["toi"] == 2.0 ** -18


# ---- the CPU oracle against the reference's numbers -------------------------------------------------
def test_oracle_cloth_ball_counts_and_axis():
    _need("cloth-ball")
    import orc

    V0, V1, E, F = _load("cloth-ball")
    assert (len(V0), len(E), len(F)) == (CLOTH_BALL["nV"], CLOTH_BALL["nE"], CLOTH_BALL["nF"])
    vb, eb, fb = orc.build_boxes(V0, V1, E, F, 0.0)
    vf, ax_vf, _ = orc.sort_and_sweep(vb, fb, sort_axis=0, nthreads=8)
    ee, ax_ee, _ = orc.sort_and_sweep(eb, sort_axis=0, nthreads=8)
    assert ax_vf == 0 and ax_ee == 0
    assert len(vf) == CLOTH_BALL["n_vf"] and len(ee) == CLOTH_BALL["n_ee"]
    _superset(vf, ee, len(V0), len(E), "cloth-ball")


def test_oracle_cloth_ball_toi():
    _need("cloth-ball")
    import orc

    V0, V1, E, F = _load("cloth-ball")
    toi = orc.ccd(V0, V1, E, F, 0.0, -1, 1e-6, True, nthreads=8)[0]
    assert toi == pytest.approx(CLOTH_BALL["toi"])  # tests/test_narrow_phase.cu:65 uses Catch::Approx


@pytest.mark.parametrize("name", sorted(SCENES))
def test_oracle_reports_every_ground_truth_pair(name):
    _need(name)
    import orc

    V0, V1, E, F = _load(name)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F, 0.0)
    vf, _, _ = orc.sort_and_sweep(vb, fb, nthreads=8)
    ee, _, _ = orc.sort_and_sweep(eb, nthreads=8)
    _superset(vf, ee, len(V0), len(E), name)


# ---- the HIP library against the reference's numbers ------------------------------------------------
@pytest.fixture()
def gpu():
    import sccd

    c = sccd.Context(0)
    yield sccd, c
    c.close()


def _hip_overlaps(sccd, ctx, V0, V1, E, F):
    """vertex-face and edge-edge overlaps of the HIP broad phase (inflation radius 0: tests/io.cpp:35) + the next sort axes"""
    import ctypes as C

    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    vb, eb, fb = sccd.DeviceAABBs.from_mesh(mesh, 0.0)
    bp = sccd.BroadPhase(ctx)
    bp.build(vb, fb)
    vf = np.asarray(bp.detect_overlaps()).reshape(-1, 2)
    bp.build(eb)
    ee = np.asarray(bp.detect_overlaps()).reshape(-1, 2)
    ax = []
    for a, b in ((vb, fb), (eb, None)):
        v = C.c_int(-1)
        ctx._check(sccd.lib().sccd_boxes_variance_axis(ctx._h, a._h, b._h if b is not None else None, C.byref(v)))
        ax.append(v.value)
    return vf, ee, ax


@pytest.mark.gpu
def test_hip_cloth_ball_counts_axis_and_toi(gpu):
    _need("cloth-ball")
    sccd, ctx = gpu
    V0, V1, E, F = _load("cloth-ball")
    assert (len(V0), len(E), len(F)) == (CLOTH_BALL["nV"], CLOTH_BALL["nE"], CLOTH_BALL["nF"])
    vf, ee, (ax_vf, ax_ee) = _hip_overlaps(sccd, ctx, V0, V1, E, F)
    assert ax_vf == 0 and ax_ee == 0  # sort_and_sweep.cpp:176-195
    assert len(vf) == CLOTH_BALL["n_vf"] and len(ee) == CLOTH_BALL["n_ee"]
    _superset(vf, ee, len(V0), len(E), "cloth-ball")
    toi, col = sccd.ccd(V0, V1, E, F, 0.0, -1, 1e-6, True, ctx=ctx, want_collisions=True)
    assert toi == pytest.approx(CLOTH_BALL["toi"])
    assert all(toi <= t for t in col["toi"])  # tests/test_narrow_phase.cu:60-62


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SCENES))
def test_hip_reports_every_ground_truth_pair(gpu, name):
    _need(name)
    sccd, ctx = gpu
    V0, V1, E, F = _load(name)
    vf, ee, _ = _hip_overlaps(sccd, ctx, V0, V1, E, F)
    _superset(vf, ee, len(V0), len(E), name)
