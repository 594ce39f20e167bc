"""Deterministic synthetic scenes for the configurations of BASELINE.json.

All randomness is splitmix64 -> (x >> 11) * 2^-53, evaluated with wrapping uint64 numpy
arithmetic, so this container and the GPU box generate bit-identical inputs.

Meshes follow the reference's conventions: V is n x 3 float64, F is k x 3 int32, E is m x 2
int32 holding the unique undirected face edges (what igl::edges produces in tests/io.cpp:19-21).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64_uniform(seed, n):
    """n doubles in [0,1): stream element k is mix(seed + (k+1)*golden)."""
    with np.errstate(over="ignore"):
        k = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def edges_from_faces(F):
    """Unique undirected edges of a triangle mesh, rows sorted (lo, hi), lexicographic order."""
    F = np.asarray(F, dtype=np.int64)
    e = np.concatenate([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]], axis=0)
    e.sort(axis=1)
    key = e[:, 0] * (int(F.max()) + 1) + e[:, 1]
    _, idx = np.unique(key, return_index=True)
    return e[idx].astype(np.int32)


def cloth_grid(n, spacing=None):
    """n x n vertex grid in the xy-plane (z = 0), 2*(n-1)^2 triangles."""
    h = 1.0 / (n - 1) if spacing is None else spacing
    j, i = np.meshgrid(np.arange(n), np.arange(n))
    V = np.stack([j.ravel() * h, i.ravel() * h, np.zeros(n * n)], axis=1).astype(np.float64)
    ii, jj = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    v00 = (ii * n + jj).ravel()
    v01 = v00 + 1
    v10 = v00 + n
    v11 = v10 + 1
    F = np.concatenate([np.stack([v00, v01, v11], 1), np.stack([v00, v11, v10], 1)], axis=0)
    # interleave so the two triangles of a cell are adjacent in id space
    F = F.reshape(2, -1, 3).transpose(1, 0, 2).reshape(-1, 3).astype(np.int32)
    return V, F


def icosphere(subdiv, radius=1.0, centre=(0.0, 0.0, 0.0)):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    V = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    F = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
         (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5),
         (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    V = [np.array(v, dtype=np.float64) / np.linalg.norm(v) for v in V]
    for _ in range(subdiv):
        cache = {}
        F2 = []

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = V[a] + V[b]
                V.append(m / np.linalg.norm(m))
                cache[key] = len(V) - 1
            return cache[key]

        for a, b, c in F:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            F2 += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        F = F2
    V = np.array(V) * radius + np.array(centre, dtype=np.float64)
    return V, np.array(F, dtype=np.int32)


def cloth_ball(n=72, ball_subdiv=3, seed=92):
    """C1/C2: n x n cloth at z = 0.5 dropping 0.2 onto a static icosphere.

    Default: 72 x 72 cloth (10,082 tris) + icosphere-3 (642 V / 1,280 F).
    Returns V0, V1, E, F.
    """
    Vc, Fc = cloth_grid(n)
    Vc[:, 2] = 0.5
    Vb, Fb = icosphere(ball_subdiv, radius=0.25, centre=(0.5, 0.5, 0.2))
    xi = (splitmix64_uniform(seed, Vc.shape[0] * 3).reshape(-1, 3) * 2.0 - 1.0) * 0.01
    Vc1 = Vc + xi
    Vc1[:, 2] -= 0.2
    V0 = np.concatenate([Vc, Vb], axis=0)
    V1 = np.concatenate([Vc1, Vb], axis=0)
    F = np.concatenate([Fc, Fb + Vc.shape[0]], axis=0).astype(np.int32)
    return V0, V1, edges_from_faces(F), F


def folded_cloth(n=708, seed=7):
    """C4/C5: n x n cloth, z0 = 0.05 sin(6 pi x), pushed through itself.

    n = 708 gives V = 501,264, F = 999,698, E = 1,500,961.
    """
    V0, F = cloth_grid(n)
    z0 = 0.05 * np.sin(6.0 * np.pi * V0[:, 0])
    V0[:, 2] = z0
    xi = (splitmix64_uniform(seed, V0.shape[0] * 3).reshape(-1, 3) * 2.0 - 1.0) * 1e-3
    V1 = V0 + xi
    V1[:, 2] += -0.1 * z0 / 0.05
    return V0, V1, edges_from_faces(F), F


AABB_DTYPE = np.dtype(
    [("min", "<f8", (3,)), ("max", "<f8", (3,)), ("vertex_ids", "<i4", (3,)), ("element_id", "<i4")],
    align=True,
)


def random_boxes(n=1_000_000, seed=42, max_extent=0.027, z_scale=1.0):
    """C3: n boxes, centres U[0,1)^3, full extents U(0, max_extent) per axis, nothing filtered
    by shared vertices (vertex_ids = {3i, 3i+1, 3i+2}).  z_scale = 0.01 gives the cloth-like
    variant of SURVEY 8d."""
    u = splitmix64_uniform(seed, 6 * n).reshape(n, 6)
    c = u[:, :3].copy()
    ext = u[:, 3:] * max_extent
    c[:, 2] *= z_scale
    ext[:, 2] *= z_scale
    b = np.zeros(n, AABB_DTYPE)
    b["min"] = c - 0.5 * ext
    b["max"] = c + 0.5 * ext
    i = np.arange(n, dtype=np.int64)
    b["vertex_ids"] = np.stack([3 * i, 3 * i + 1, 3 * i + 2], axis=1).astype(np.int32)
    b["element_id"] = i.astype(np.int32)
    return b


def triangle_soup(n_tris=200, seed=1, size=0.15, motion=0.3):
    """Small stress scene for parity tests: random disconnected triangles flying through each
    other (many genuine VF/EE impacts at assorted times)."""
    u = splitmix64_uniform(seed, n_tris * 9 + n_tris * 3).astype(np.float64)
    c = u[: n_tris * 3].reshape(n_tris, 1, 3)
    off = (u[n_tris * 3: n_tris * 12].reshape(n_tris, 3, 3) - 0.5) * size
    V0 = (c + off).reshape(-1, 3)
    d = (splitmix64_uniform(seed + 1000, n_tris * 3).reshape(n_tris, 1, 3) - 0.5) * 2.0 * motion
    V1 = (c + off + d).reshape(-1, 3)
    F = np.arange(n_tris * 3, dtype=np.int32).reshape(n_tris, 3)
    return V0, V1, edges_from_faces(F), F
