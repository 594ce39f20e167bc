"""The oracle against the REFERENCE'S OWN CPU broad phase, compiled unmodified from /root/reference.

`make -C oracle -f ref_build.mk SCCD_EIGEN_DIR=... SCCD_TBB_DIR=... SCCD_SPDLOG_DIR=...` builds oracle/_ref/ref_driver from the
reference's aabb.cpp / sort_and_sweep.cpp (in place, against REAL Eigen / oneTBB / spdlog trees; the recipe refuses to build
without them and this image has none, so here every test below SKIPS).  With the driver present the tests pin, on seeded scenes:

* build_vertex / edge / face_boxes (aabb.cpp:63-133): every box byte-equal to the oracle's (orc.build_boxes);
* sort_and_sweep (sort_and_sweep.cpp:198-240), two lists (vertices x faces) and one list (edges): sorted pair lists identical,
  same next sort axis;
* 100,000 random boxes, one list: pairs identical.
The `-m gpu` half runs the same scenes through the HIP library.  Together with tests/test_reference_constants.py (needs the
reference's sample data) these are the two independent ways to turn DESIGN.md's "parity unpinned" into "pinned".
"""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "scalable-ccd_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
need_driver = pytest.mark.skipif(
    not os.path.exists(DRIVER),
    reason="oracle/_ref/ref_driver is not built: `make -C oracle -f ref_build.mk SCCD_EIGEN_DIR=.. SCCD_TBB_DIR=.. SCCD_SPDLOG_DIR=..` "
           "(needs real Eigen / oneTBB / spdlog headers, which this image does not have)")


def test_the_recipe_refuses_to_build_without_real_headers():
    """Always runs: no stand-ins -- without the three include trees the recipe stops before compiling anything."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("SCCD_")}
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "ref_build.mk", "check"], env=env,
                       capture_output=True, text=True)
    if os.path.isdir("/root/reference/src/scalable_ccd"):
        assert r.returncode != 0 and "refusing to build" in r.stdout + r.stderr, (r.returncode, r.stdout, r.stderr)
    else:  # (the GPU box: no reference tree at all)
        assert r.returncode != 0


def _run_mesh(tmp_path, V0, V1, E, F, radius=0.0):
    fin, fout = tmp_path / "mesh.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<3i", len(V0), len(E), len(F)))
        for M in (V0, V1):
            f.write(np.asfortranarray(M, dtype="<f8").tobytes(order="F"))
        for M in (E, F):
            f.write(np.asfortranarray(M, dtype="<i4").tobytes(order="F"))
    subprocess.run([DRIVER, "mesh", str(fin), str(fout), repr(float(radius))], check=True, timeout=600)
    import orc

    raw = open(fout, "rb").read()
    nV, nE, nF = struct.unpack_from("<3i", raw, 0)
    off = 12
    boxes = []
    for n in (nV, nE, nF):
        boxes.append(np.frombuffer(raw, dtype=orc.AABB_DTYPE, count=n, offset=off))
        off += 64 * n
    out = []
    for _ in range(2):
        axis, n = struct.unpack_from("<2i", raw, off)
        off += 8
        out.append((axis, np.frombuffer(raw, dtype="<i4", count=2 * n, offset=off).reshape(-1, 2)))
        off += 8 * n
    return boxes, out


def _scene(name):
    from sccd import scenes

    if name == "cloth_ball":
        return scenes.cloth_ball()
    return scenes.triangle_soup(600, seed=5)


@need_driver
@pytest.mark.parametrize("name", ["cloth_ball", "soup"])
def test_oracle_boxes_and_pairs_equal_the_reference(tmp_path, name):
    import orc

    V0, V1, E, F = _scene(name)
    (rvb, reb, rfb), ((ax_vf, vf), (ax_ee, ee)) = _run_mesh(tmp_path, V0, V1, E, F)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    for mine, ref in ((vb, rvb), (eb, reb), (fb, rfb)):
        assert mine.tobytes() == ref.tobytes()
    o_vf, o_ax_vf, _ = orc.sort_and_sweep(vb, fb, 0)
    o_ee, o_ax_ee, _ = orc.sort_and_sweep(eb, None, 0)
    assert (o_ax_vf, o_ax_ee) == (ax_vf, ax_ee)
    assert np.array_equal(o_vf, vf) and np.array_equal(o_ee, ee)


@need_driver
def test_oracle_random_boxes_equal_the_reference(tmp_path):
    import orc
    from sccd import scenes

    boxes = scenes.random_boxes(100_000, seed=11, max_extent=0.06)
    fin, fout = tmp_path / "boxes.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<i", len(boxes)))
        f.write(np.ascontiguousarray(boxes).tobytes())
    subprocess.run([DRIVER, "boxes", str(fin), str(fout)], check=True, timeout=600)
    raw = open(fout, "rb").read()
    axis, n = struct.unpack_from("<2i", raw, 0)
    ref = np.frombuffer(raw, dtype="<i4", count=2 * n, offset=8).reshape(-1, 2)
    mine, o_axis, _ = orc.sort_and_sweep(boxes, None, 0)
    assert o_axis == axis and np.array_equal(mine, ref)


@need_driver
@pytest.mark.gpu
def test_hip_pairs_equal_the_reference(tmp_path):
    import sccd

    V0, V1, E, F = _scene("cloth_ball")
    (rvb, reb, rfb), ((_, vf), (_, ee)) = _run_mesh(tmp_path, V0, V1, E, F)
    ctx = sccd.Context(0)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    vb, eb, fb = sccd.DeviceAABBs.from_mesh(mesh)
    host = [d.download() for d in (vb, eb, fb)]
    for mine, ref in zip(host, (rvb, reb, rfb)):
        assert mine.tobytes() == ref.tobytes()

    def srt(p):
        p = np.asarray(p).reshape(-1, 2)
        return p[np.lexsort((p[:, 1], p[:, 0]))]

    got_vf = srt(sccd.sort_and_sweep(host[0], host[2], 0, ctx=ctx)[0])
    got_ee = srt(sccd.sort_and_sweep(host[1], None, 0, ctx=ctx)[0])
    assert np.array_equal(got_vf, vf) and np.array_equal(got_ee, ee)
