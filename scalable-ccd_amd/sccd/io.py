"""Mesh / fixture I/O and the ground-truth comparer of the reference's test suite.

Mirrors tests/io.cpp:10-38 (parse_mesh: two PLY frames -> V0, V1, F and E = igl::edges(F)) and
tests/ground_truth.cpp:14-64 (compare_mathematica: every ground-truth pair must be among the
reported overlaps) so that the reference's sample scenes (cloth-ball 92, armadillo-rollers 326,
cloth-funnel 227, n-body 18, rod-twist 3036 -- Sample-Scalable-CCD-Data, not shipped with the
reference) can be dropped in and checked with `python -m sccd.io <t0.ply> <t1.ply> [vf.json ee.json]`.

libigl (pinned 2.6.0 in the reference, cmake/recipes/libigl.cmake:12) is not available here;
`igl_edges` restates igl::edges: the undirected edges (i < j) of the face adjacency matrix in
the column-major order of that sparse matrix, i.e. sorted by (j, i).  Edge ids in the ground
truth files depend on this order.
"""
import json
import struct
import sys

import numpy as np

_PLY_TYPES = {
    "char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h", "ushort": "H", "uint16": "H",
    "int": "i", "int32": "i", "uint": "I", "uint32": "I", "float": "f", "float32": "f", "double": "d", "float64": "d",
}


def igl_edges(F):
    """E = igl::edges(F): unique undirected edges, rows (i, j) with i < j, ordered by (j, i)."""
    F = np.asarray(F, dtype=np.int64).reshape(-1, 3)
    if len(F) == 0:
        return np.zeros((0, 2), np.int32)
    e = np.concatenate([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]], axis=0)
    e = e[e[:, 0] != e[:, 1]]  # the adjacency matrix has no diagonal
    e.sort(axis=1)
    n = int(F.max()) + 1
    key = np.unique(e[:, 1] * n + e[:, 0])  # column (the larger index) first
    return np.stack([key % n, key // n], axis=1).astype(np.int32)


def read_ply(path):
    """Vertices (n x 3 float64) and triangles (m x 3 int32) of a PLY file: ascii,
    binary_little_endian or binary_big_endian; extra vertex properties and extra elements are
    skipped; polygons with more than three corners are fanned like libigl's reader does."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt = None
        elements = []  # [name, count, [(kind, name, type...)]]
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append([tok[1], int(tok[2]), []])
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1][2].append(("list", tok[4], tok[2], tok[3]))
                else:
                    elements[-1][2].append(("scalar", tok[2], tok[1]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt!r}")
        V, faces = None, []
        if fmt == "ascii":
            tokens = f.read().split()
            pos = 0
            for name, count, props in elements:
                rows = []
                for _ in range(count):
                    row = {}
                    for p in props:
                        if p[0] == "scalar":
                            row[p[1]] = float(tokens[pos])
                            pos += 1
                        else:
                            k = int(tokens[pos])
                            row[p[1]] = [int(float(t)) for t in tokens[pos + 1:pos + 1 + k]]
                            pos += 1 + k
                    rows.append(row)
                if name == "vertex":
                    V = np.array([[r["x"], r["y"], r["z"]] for r in rows], dtype=np.float64).reshape(-1, 3)
                elif name == "face":
                    key = next(p[1] for p in props if p[0] == "list")
                    faces = [r[key] for r in rows]
        else:
            end = "<" if fmt == "binary_little_endian" else ">"
            for name, count, props in elements:
                if all(p[0] == "scalar" for p in props):
                    dt = np.dtype([(p[1], end + _PLY_TYPES[p[2]]) for p in props])
                    data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
                    if name == "vertex":
                        V = np.stack([data["x"], data["y"], data["z"]], axis=1).astype(np.float64)
                else:
                    rows = []
                    for _ in range(count):
                        row = {}
                        for p in props:
                            if p[0] == "scalar":
                                c = _PLY_TYPES[p[2]]
                                (row[p[1]],) = struct.unpack(end + c, f.read(struct.calcsize(c)))
                            else:
                                cc, ci = _PLY_TYPES[p[2]], _PLY_TYPES[p[3]]
                                (k,) = struct.unpack(end + cc, f.read(struct.calcsize(cc)))
                                row[p[1]] = list(struct.unpack(end + ci * k, f.read(struct.calcsize(ci) * k)))
                        rows.append(row)
                    if name == "face":
                        key = next(p[1] for p in props if p[0] == "list")
                        faces = [r[key] for r in rows]
    if V is None:
        raise ValueError(f"{path}: no vertex element")
    tris = []
    for poly in faces:
        for k in range(1, len(poly) - 1):
            tris.append((poly[0], poly[k], poly[k + 1]))
    F = np.array(tris, dtype=np.int32).reshape(-1, 3)
    if len(F) and (F.min() < 0 or F.max() >= len(V)):
        raise ValueError(f"{path}: face index out of range")
    return V, F


def write_ply(path, V, F, binary=True):
    """Minimal writer (double vertices, int32 triangles) for fixtures and round-trip tests."""
    V = np.asarray(V, dtype=np.float64).reshape(-1, 3)
    F = np.asarray(F, dtype=np.int32).reshape(-1, 3)
    head = ("ply\nformat %s 1.0\ncomment written by sccd.io\nelement vertex %d\nproperty double x\nproperty double y\n"
            "property double z\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n"
            % ("binary_little_endian" if binary else "ascii", len(V), len(F)))
    with open(path, "wb") as f:
        f.write(head.encode("ascii"))
        if binary:
            f.write(V.astype("<f8").tobytes())
            rec = np.zeros(len(F), dtype=[("n", "u1"), ("v", "<i4", 3)])
            rec["n"] = 3
            rec["v"] = F
            f.write(rec.tobytes())
        else:
            for v in V:
                f.write(("%r %r %r\n" % (float(v[0]), float(v[1]), float(v[2]))).encode("ascii"))
            for t in F:
                f.write(("3 %d %d %d\n" % (t[0], t[1], t[2])).encode("ascii"))


def parse_mesh(file_t0, file_t1):
    """tests/io.cpp:10-22: both frames of a scene -> (V0, V1, E, F)."""
    V0, F0 = read_ply(file_t0)
    V1, F1 = read_ply(file_t1)
    if V0.shape != V1.shape:
        raise ValueError("the two frames have different vertex counts")
    if F0.shape != F1.shape or not np.array_equal(F0, F1):
        raise ValueError("the two frames have different faces")
    return V0, V1, igl_edges(F1), F1


def read_ground_truth(path):
    """A ground-truth file is a JSON array of [a, b] id pairs (ground_truth.cpp:40-44)."""
    with open(path, "r") as f:
        arr = json.load(f)
    return np.asarray(arr, dtype=np.int64).reshape(-1, 2)


def offset_for_ground_truth(vf, ee, n_vertices, n_edges):
    """The id space the ground truth was generated in (test_broad_phase.cpp:66-74): vertices,
    then edges, then faces.  Returns copies (vf_offset, ee_offset)."""
    vf = np.asarray(vf, dtype=np.int64).reshape(-1, 2).copy()
    ee = np.asarray(ee, dtype=np.int64).reshape(-1, 2).copy()
    ee += n_vertices
    vf[:, 1] += n_vertices + n_edges
    return vf, ee


def missing_from(overlaps, truth, mask=None):
    """compare_mathematica (ground_truth.cpp:27-64): the ground-truth pairs NOT among the
    reported overlaps (optionally only those flagged in `mask`).  Empty <=> the check passes:
    the broad phase may report more pairs than the ground truth, never fewer."""
    ov = np.asarray(overlaps, dtype=np.int64).reshape(-1, 2)
    if mask is not None:
        ov = ov[np.asarray(mask, dtype=bool)]
    tr = np.asarray(truth, dtype=np.int64).reshape(-1, 2)
    if len(tr) == 0:
        return tr
    base = int(max(ov.max(initial=0), tr.max(initial=0))) + 1
    have = np.unique(ov[:, 0] * base + ov[:, 1])
    want = tr[:, 0] * base + tr[:, 1]
    return tr[~np.isin(want, have)]


def main(argv=None):
    """python -m sccd.io t0.ply t1.ply [vf.json ee.json]: run ccd() and the broad phase on a scene
    of the reference's sample data and apply its checks."""
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) not in (2, 4):
        print(main.__doc__)
        return 2
    import sccd

    V0, V1, E, F = parse_mesh(argv[0], argv[1])
    print(f"{len(V0)} vertices, {len(E)} edges, {len(F)} faces")
    ctx = sccd.default_context()
    vb = sccd.build_vertex_boxes(V0, V1, ctx=ctx)
    eb = sccd.build_edge_boxes(vb, E, ctx=ctx)
    fb = sccd.build_face_boxes(vb, F, ctx=ctx)
    bp = sccd.BroadPhase(ctx)
    bp.build(sccd.DeviceAABBs(vb, ctx), sccd.DeviceAABBs(fb, ctx))
    vf = bp.detect_overlaps().reshape(-1, 2)
    bp.build(sccd.DeviceAABBs(eb, ctx))
    ee = bp.detect_overlaps().reshape(-1, 2)
    print(f"vf overlaps {len(vf)}  ee overlaps {len(ee)}")
    rc = 0
    if len(argv) == 4:
        vfo, eeo = offset_for_ground_truth(vf, ee, len(V0), len(E))
        for name, ov, path in (("vf", vfo, argv[2]), ("ee", eeo, argv[3])):
            miss = missing_from(ov, read_ground_truth(path))
            print(f"{name}: {len(miss)} ground-truth pairs missing")
            rc |= int(len(miss) > 0)
    toi = sccd.ccd(V0, V1, E, F, 0.0, -1, 1e-6, True, ctx=ctx)
    print(f"toi {toi!r}")
    return rc


if __name__ == "__main__":
    sys.exit(main())
