"""Mesh / fixture I/O and the ground-truth comparer (sccd/io.py mirrors the reference's
tests/io.cpp and tests/ground_truth.cpp).  No GPU needed except for the last test."""
import json
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-ccd_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from sccd import io as sio  # noqa: E402
from sccd import scenes  # noqa: E402


def test_igl_edges_order_and_content():
    F = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    # adjacency matrix, upper triangle, column by column: sorted by (larger, smaller) index
    assert sio.igl_edges(F).tolist() == [[0, 1], [0, 2], [1, 2], [0, 3], [2, 3]]
    V, F = scenes.icosphere(2)
    E = sio.igl_edges(F)
    assert len(E) == len(F) * 3 // 2  # closed manifold
    assert (E[:, 0] < E[:, 1]).all()
    key = E[:, 1].astype(np.int64) * len(V) + E[:, 0]
    assert (np.diff(key) > 0).all()
    # same SET as the generator's own edge list
    want = scenes.edges_from_faces(F)
    assert sorted(map(tuple, E.tolist())) == sorted(map(tuple, want.tolist()))
    assert sio.igl_edges(np.zeros((0, 3), np.int32)).shape == (0, 2)


@pytest.mark.parametrize("binary", [True, False])
def test_ply_round_trip(tmp_path, binary):
    V, F = scenes.icosphere(1)
    V = V * 1.2345678901234567 + 0.1
    p = tmp_path / "m.ply"
    sio.write_ply(p, V, F, binary=binary)
    V2, F2 = sio.read_ply(p)
    assert np.array_equal(V2, V) and np.array_equal(F2, F)  # %r / raw doubles: bit-exact


def test_ply_big_endian_float_extra_properties_and_quads(tmp_path):
    # what a typical exporter writes: float coordinates, normals, a quad face, an extra element
    head = (b"ply\nformat binary_big_endian 1.0\ncomment exporter\nelement vertex 4\nproperty float x\nproperty float y\n"
            b"property float z\nproperty uchar red\nelement face 1\nproperty list uchar uint vertex_index\n"
            b"element edge 1\nproperty int vertex1\nproperty int vertex2\nend_header\n")
    body = b""
    pts = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0.5)]
    for x, y, z in pts:
        body += struct.pack(">fffB", x, y, z, 7)
    body += struct.pack(">BIIII", 4, 0, 1, 2, 3)
    body += struct.pack(">ii", 0, 1)
    p = tmp_path / "q.ply"
    p.write_bytes(head + body)
    V, F = sio.read_ply(p)
    assert V.dtype == np.float64 and np.allclose(V, np.array(pts, float))
    assert F.tolist() == [[0, 1, 2], [0, 2, 3]]  # fanned
    with pytest.raises(ValueError):
        (tmp_path / "bad.ply").write_bytes(b"solid\n")
        sio.read_ply(tmp_path / "bad.ply")


def test_parse_mesh_checks_the_two_frames(tmp_path):
    V, F = scenes.icosphere(1)
    sio.write_ply(tmp_path / "a.ply", V, F)
    sio.write_ply(tmp_path / "b.ply", V + 0.01, F)
    V0, V1, E, F2 = sio.parse_mesh(tmp_path / "a.ply", tmp_path / "b.ply")
    assert V0.shape == V1.shape and np.array_equal(F2, F) and np.array_equal(E, sio.igl_edges(F))
    sio.write_ply(tmp_path / "c.ply", V[:-1], F[(F < len(V) - 1).all(axis=1)])
    with pytest.raises(ValueError):
        sio.parse_mesh(tmp_path / "a.ply", tmp_path / "c.ply")


def test_ground_truth_superset_check(tmp_path):
    ov = np.array([[1, 5], [2, 9], [3, 3], [7, 8]])
    (tmp_path / "gt.json").write_text(json.dumps([[2, 9], [7, 8]]))
    gt = sio.read_ground_truth(tmp_path / "gt.json")
    assert len(sio.missing_from(ov, gt)) == 0  # more pairs than the ground truth is fine
    assert sio.missing_from(ov[:2], gt).tolist() == [[7, 8]]  # fewer is not
    assert sio.missing_from(ov, gt, mask=[1, 0, 1, 1]).tolist() == [[2, 9]]  # only flagged results count
    assert sio.missing_from(np.zeros((0, 2)), np.zeros((0, 2))).shape == (0, 2)
    vf, ee = sio.offset_for_ground_truth([[0, 1]], [[2, 3]], n_vertices=10, n_edges=100)
    assert vf.tolist() == [[0, 111]] and ee.tolist() == [[12, 13]]  # V, then E, then F ids


@pytest.mark.gpu
def test_scene_files_through_the_whole_pipeline(tmp_path, capsys):
    """The drop-in check for the reference's sample scenes, on a generated cloth-ball: PLY frames
    + ground-truth JSON (brute-force overlaps, subsampled like a true-positive list) -> main()."""
    import orc

    V0, V1, E_gen, F = scenes.cloth_ball(24, 1, seed=5)
    sio.write_ply(tmp_path / "f0.ply", V0, F, binary=True)
    sio.write_ply(tmp_path / "f1.ply", V1, F, binary=False)
    E = sio.igl_edges(F)
    vb, eb, fb = orc.build_boxes(V0, V1, E, F)
    vf, _, _ = orc.sort_and_sweep(vb, fb)
    ee, _, _ = orc.sort_and_sweep(eb)
    vfo, eeo = sio.offset_for_ground_truth(vf, ee, len(V0), len(E))
    (tmp_path / "vf.json").write_text(json.dumps(vfo[::3].tolist()))
    (tmp_path / "ee.json").write_text(json.dumps(eeo[::3].tolist()))
    rc = sio.main([str(tmp_path / "f0.ply"), str(tmp_path / "f1.ply"), str(tmp_path / "vf.json"), str(tmp_path / "ee.json")])
    out = capsys.readouterr().out
    assert rc == 0, out
    assert f"vf overlaps {len(vf)}  ee overlaps {len(ee)}" in out
    assert "vf: 0 ground-truth pairs missing" in out and "ee: 0 ground-truth pairs missing" in out
    want_toi, _, _ = orc.ccd(V0, V1, E, F)
    assert f"toi {want_toi!r}" in out
    # a ground truth with a pair the broad phase cannot produce must fail the check
    (tmp_path / "bad.json").write_text(json.dumps(vfo[:2].tolist() + [[0, len(V0) + len(E) + len(F) - 1 + 10**6]]))
    assert sio.main([str(tmp_path / "f0.ply"), str(tmp_path / "f1.ply"), str(tmp_path / "bad.json"), str(tmp_path / "ee.json")]) == 1
