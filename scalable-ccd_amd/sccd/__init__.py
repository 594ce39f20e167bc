"""sccd -- Python host binding of libsccd_hip.so (the C ABI of include/sccd.h).

Mirrors the reference's host API for the CCD hot path (names and argument meaning follow
src/scalable_ccd/cuda/{ccd.cuh, broad_phase/broad_phase.cuh, broad_phase/aabb.cuh,
narrow_phase/narrow_phase.cuh, ipc_ccd_strategy.hpp}):

    toi = sccd.ccd(V0, V1, E, F, min_distance, max_iterations, tolerance, allow_zero_toi)
    vb = sccd.build_vertex_boxes(V0, V1, inflation); eb = sccd.build_edge_boxes(vb, E); ...
    bp = sccd.BroadPhase(); bp.build(sccd.DeviceAABBs(vb), sccd.DeviceAABBs(fb))
    overlaps = bp.detect_overlaps()
    toi = sccd.narrow_phase(mesh, overlaps, is_vf, max_iter, tol, ms, allow_zero_toi, toi)

Errors raise RuntimeError like the reference's std::runtime_error.  There is no CPU fallback:
importing works anywhere, but creating a Context without the HIP extension or without a GPU
raises.
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SCCD_LIB: another build of the same library -- kernel variants under measurement, tools/variants.sh)
_LIB_PATH = os.environ.get("SCCD_LIB") or os.path.join(_HERE, "libsccd_hip.so")
_lib = None

AABB_DTYPE = np.dtype(
    [("min", "<f8", (3,)), ("max", "<f8", (3,)), ("vertex_ids", "<i4", (3,)), ("element_id", "<i4")],
    align=True,
)
COLLISION_DTYPE = np.dtype([("aid", "<i4"), ("bid", "<i4"), ("toi", "<f8")], align=True)
assert AABB_DTYPE.itemsize == 64 and COLLISION_DTYPE.itemsize == 16

# sccd.h option ids
OPT_ARITH, OPT_NARROW_ALGO, OPT_SWEEP_ALGO, OPT_SORT_AXIS = 1, 2, 3, 4
ARITH_STRICT, ARITH_FMA = 0, 1
ARITH_DEFAULT = ARITH_FMA  # the library's default contract (include/sccd.h SCCD_OPT_ARITH): the reference's nvcc build fuses
OPT_SHARD_RANK, OPT_SHARD_COUNT, OPT_OVERLAP_CAPACITY, OPT_PROFILE, OPT_MAX_OVERLAP_CUTOFF = 5, 6, 7, 8, 9
OPT_MEMORY_LIMIT_MB = 10
OPT_SCALAR = 11  # 1: the reference's float build (SCALABLE_CCD_USE_DOUBLE=OFF)
OPT_PASSES_APART = 13  # 1: ccd() runs its two passes one after the other (measurements)
OPT_CELL_FACTOR_MILLI, OPT_BUILD_SCAN = 17, 18  # grid cell size (thousandths of the mean extent); count -> scan -> fill build
OPT_TOI_GUESS, OPT_TOI_GUESS_HITS, OPT_TOI_GUESS_MISSES = 19, 20, 21  # the speculative TOI bound of ccd() on a mesh (see sccd.h)
OPT_CULL = 24  # 1 (default): ccd()'s passes drop pairs that provably have no impact before the bisection (csrc/narrow_cull.inc)
OPT_TWO_HALVES = 25  # 1 (default): plain narrow launches from a TOI above 0.5 run as two launches over the halves of time (see sccd.h)
OPT_ALLOC_COUNT = 23  # read-only: device allocations made by the library's grow-only buffers (a step that allocates is a slow step)
OPT_SPEC_HITS, OPT_SPEC_MISSES = 15, 16  # read-only counters of the speculative build (set: reset)
OPT_READ_BACKS = 28  # read-only: ReadBack launches so far (a default step on a warm context needs none)
OPT_DEVICE_SPAN_NS, OPT_HOST_WAITS = 26, 27  # read-only: the device's own span of the last ccd() call (ns); host waits (read-backs, verdicts) so far
OPT_LIMIT_LEVEL_ORDER = 14  # 1: check limits always on the level-synchronous kernels; default: fast kernel + certificate (same result)
PROF_NAMES = ["boxes", "sort", "cull", "sweep", "narrow_vf", "narrow_ee", "sweep_ee"]  # SCCD_PROF_* ("sweep_ee": a mesh's edge list; "sweep": every other sweep)

# every symbol include/sccd.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = [
    "sccd_create", "sccd_destroy", "sccd_last_error", "sccd_version", "sccd_set_stream", "sccd_synchronize",
    "sccd_set_option", "sccd_get_option", "sccd_mesh_create", "sccd_mesh_update_vertices", "sccd_mesh_assign", "sccd_mesh_destroy",
    "sccd_build_vertex_boxes", "sccd_build_edge_boxes", "sccd_build_face_boxes", "sccd_boxes_create",
    "sccd_boxes_from_mesh", "sccd_boxes_size", "sccd_boxes_download", "sccd_boxes_destroy",
    "sccd_broad_phase_create", "sccd_broad_phase_destroy", "sccd_broad_phase_build",
    "sccd_broad_phase_detect_overlaps_partial", "sccd_broad_phase_detect_overlaps", "sccd_broad_phase_is_complete",
    "sccd_broad_phase_num_boxes", "sccd_broad_phase_candidates", "sccd_free", "sccd_narrow_phase", "sccd_ccd",
    "sccd_ccd_mesh", "sccd_ccd_mesh_prepare", "sccd_ccd_mesh_pass", "sccd_ipc_ccd_strategy", "sccd_get_profile", "sccd_reset_profile", "sccd_sort_pairs_u32",
    "sccd_shard_bounds", "sccd_boxes_variance_axis", "sccd_selftest_lds_gather",
    "sccd_dev_alloc", "sccd_dev_free", "sccd_dev_upload", "sccd_dev_download", "sccd_dev_copy", "sccd_ccd_collisions",
    "sccd_ccd_mesh_dev", "sccd_get_stream", "sccd_query_cull", "sccd_query_cull_slab", "sccd_ccd_mesh_from",
    "sccd_abi_sizeof_stats", "sccd_abi_prof_count",
]
ABI_VERSION_PREFIX = "sccd-hip 0.4"  # what this binding was written against (sccd_version)


class Stats(C.Structure):
    _fields_ = [
        ("n_vf_pairs", C.c_int64), ("n_ee_pairs", C.c_int64),
        ("n_vf_candidates", C.c_int64), ("n_ee_candidates", C.c_int64),
        ("n_vf_checks", C.c_int64), ("n_ee_checks", C.c_int64),
        ("ms_boxes", C.c_double), ("ms_sort", C.c_double), ("ms_sweep", C.c_double),
        ("ms_narrow", C.c_double), ("ms_total", C.c_double),
        ("n_vf_culled", C.c_int64), ("n_ee_culled", C.c_int64),  # (0.3) overlaps the projection cull dropped before the bisection
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def lib():
    """Load libsccd_hip.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make).  There is no CPU fallback."
            )
        # PyTorch-ROCm bundles its own HIP runtime under the same soname (libamdhip64.so.7).  Two
        # HIP runtimes in one process do not work, so when torch is installed it is imported
        # FIRST and libsccd_hip.so binds to the runtime torch already loaded.
        if os.environ.get("SCCD_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(_LIB_PATH)
        L.sccd_last_error.restype = C.c_char_p
        L.sccd_version.restype = C.c_char_p
        L.sccd_get_option.restype = C.c_int64
        L.sccd_broad_phase_num_boxes.restype = C.c_int64
        L.sccd_broad_phase_candidates.restype = C.c_int64
        L.sccd_destroy.restype = None
        L.sccd_mesh_destroy.restype = None
        L.sccd_boxes_destroy.restype = None
        L.sccd_broad_phase_destroy.restype = None
        L.sccd_free.restype = None
        L.sccd_free.argtypes = [C.c_void_p]
        # a library built from another sccd.h writes sccd_stats / profile arrays of ANOTHER size into this binding's buffers: refuse it
        ver = L.sccd_version().decode()
        if not hasattr(L, "sccd_abi_sizeof_stats") or not ver.startswith(ABI_VERSION_PREFIX):
            raise RuntimeError(f"{_LIB_PATH} is {ver!r}; this binding needs {ABI_VERSION_PREFIX!r} (rebuild: make)")
        L.sccd_abi_sizeof_stats.restype = C.c_size_t
        if L.sccd_abi_sizeof_stats() != C.sizeof(Stats) or L.sccd_abi_prof_count() != len(PROF_NAMES):
            raise RuntimeError(f"{_LIB_PATH}: sccd_stats is {L.sccd_abi_sizeof_stats()} bytes / {L.sccd_abi_prof_count()} profile classes there, "
                               f"{C.sizeof(Stats)} / {len(PROF_NAMES)} here")
        _lib = L
    return _lib


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def _f64cm(M):
    return np.asfortranarray(np.asarray(M, dtype=np.float64))


def _i32cm(M):
    return np.asfortranarray(np.asarray(M, dtype=np.int32))


class Context:
    """One HIP device + stream + scratch memory (sccd_ctx)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        self._children = weakref.WeakSet()  # meshes / boxes / broad phases living in this context
        rc = lib().sccd_create(C.c_int(device), C.byref(self._h))
        if rc != 0:
            raise RuntimeError("sccd_create failed: " + lib().sccd_last_error(None).decode())

    def close(self):
        """Destroys the objects created in this context first, then the context."""
        if self._h:
            for ch in list(self._children):
                ch.close()
            lib().sccd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"sccd error {rc}: " + lib().sccd_last_error(self._h).decode())

    def set_option(self, opt, value):
        self._check(lib().sccd_set_option(self._h, C.c_int(opt), C.c_int64(int(value))))

    def get_option(self, opt):
        return int(lib().sccd_get_option(self._h, C.c_int(opt)))

    def set_stream(self, hip_stream):
        self._check(lib().sccd_set_stream(self._h, C.c_void_p(int(hip_stream) if hip_stream else 0)))

    def stream_ptr(self):
        """The hipStream_t the context launches on, as an integer (torch.cuda.ExternalStream(ptr) wraps it)."""
        f = lib().sccd_get_stream
        f.restype = C.c_void_p
        return int(f(self._h) or 0)

    def synchronize(self):
        self._check(lib().sccd_synchronize(self._h))

    def profile(self):
        ms = (C.c_double * len(PROF_NAMES))()
        n = (C.c_int64 * len(PROF_NAMES))()
        self._check(lib().sccd_get_profile(self._h, ms, n))
        return {k: (ms[i], int(n[i])) for i, k in enumerate(PROF_NAMES)}

    def reset_profile(self):
        self._check(lib().sccd_reset_profile(self._h))

    def selftest_lds_gather(self, n_waves=64, n_active=64):
        """number of words that differ from the expected layout (0 = pass); sccd.h sccd_selftest_lds_gather"""
        bad = C.c_int64(-1)
        self._check(lib().sccd_selftest_lds_gather(self._h, C.c_int(n_waves), C.c_int(n_active), C.byref(bad)))
        return bad.value

    def sort_pairs_u32(self, d_keys, d_vals, n):
        """in-place radix sort of device arrays (raw device pointers)"""
        self._check(lib().sccd_sort_pairs_u32(self._h, C.c_void_p(int(d_keys)), C.c_void_p(int(d_vals)), C.c_int64(n)))


_default_ctx = None


def shard_bounds(weights, parts):
    """sccd_shard_bounds: the host-side split of weighted grid cells into `parts` contiguous
    windows that SCCD_OPT_SHARD_RANK/COUNT applies (needs no GPU)."""
    w = np.ascontiguousarray(weights, dtype=np.uint32)
    out = (C.c_int * (max(int(parts), 0) + 1))()
    rc = lib().sccd_shard_bounds(w.ctypes.data_as(C.c_void_p), C.c_int(len(w)), C.c_int(int(parts)), out)
    if rc != 0:
        raise ValueError(f"sccd_shard_bounds failed ({rc})")
    return list(out)


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class Mesh:
    """Device-resident V0, V1, E, F (the four DeviceMatrix objects of ccd.cu:103-106).

    Host numpy arrays, or raw device pointers (ints, column-major) with on_device=True and
    explicit sizes.
    """

    def __init__(self, V0, V1, E, F, ctx=None, on_device=False, nV=None, nE=None, nF=None):
        self.ctx = ctx or default_context()
        self._h = C.c_void_p()
        if on_device:
            self.nV, self.nE, self.nF = int(nV), int(nE), int(nF)
            args = (_ptr(int(V0)), _ptr(int(V1)), C.c_int(self.nV), _ptr(int(E)), C.c_int(self.nE), _ptr(int(F)), C.c_int(self.nF))
        else:
            V0c, V1c, Ec, Fc = _f64cm(V0), _f64cm(V1), _i32cm(E), _i32cm(F)
            if V0c.ndim != 2 or V0c.shape[1] != 3 or V0c.shape != V1c.shape:
                raise RuntimeError("V0, V1 must both be n x 3")  # ccd.cu:94-96
            if Ec.size and (Ec.ndim != 2 or Ec.shape[1] != 2):
                raise RuntimeError("E must be m x 2")  # ccd.cu:97
            if Fc.size and (Fc.ndim != 2 or Fc.shape[1] != 3):
                raise RuntimeError("F must be k x 3")  # ccd.cu:98
            self.nV, self.nE, self.nF = V0c.shape[0], Ec.shape[0] if Ec.size else 0, Fc.shape[0] if Fc.size else 0
            # (vertex indices are validated by the library, on the device, while it packs E and F)
            self._keep = (V0c, V1c, Ec, Fc)
            args = (_ptr(V0c), _ptr(V1c), C.c_int(self.nV), _ptr(Ec), C.c_int(self.nE), _ptr(Fc), C.c_int(self.nF))
        self.ctx._check(lib().sccd_mesh_create(self.ctx._h, *args, C.c_int(int(on_device)), C.byref(self._h)))
        self._keep = None
        self.ctx._children.add(self)

    def update_vertices(self, V0, V1, on_device=False):
        if on_device:
            self.ctx._check(lib().sccd_mesh_update_vertices(self._h, _ptr(int(V0)), _ptr(int(V1)), C.c_int(1)))
        else:
            V0c, V1c = _f64cm(V0), _f64cm(V1)
            if V0c.shape != (self.nV, 3) or V1c.shape != (self.nV, 3):
                raise RuntimeError("vertex matrices must keep their shape")
            self.ctx._check(lib().sccd_mesh_update_vertices(self._h, _ptr(V0c), _ptr(V1c), C.c_int(0)))

    def close(self):
        if self._h:
            lib().sccd_mesh_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- boxes: build_*_boxes of aabb.cuh:156-188 ----------------------------------------------
def build_vertex_boxes(V0, V1=None, inflation_radius=0.0, ctx=None):
    ctx = ctx or default_context()
    V0c = _f64cm(V0)
    V1c = V0c if V1 is None else _f64cm(V1)
    if V0c.ndim != 2 or V0c.shape[1] != 3 or V0c.shape != V1c.shape:
        raise RuntimeError("vertices must be n x 3")
    out = np.zeros(V0c.shape[0], AABB_DTYPE)
    ctx._check(lib().sccd_build_vertex_boxes(ctx._h, _ptr(V0c), _ptr(V1c), C.c_int(V0c.shape[0]), C.c_double(inflation_radius), _ptr(out)))
    return out


def build_edge_boxes(vertex_boxes, E, ctx=None):
    ctx = ctx or default_context()
    vb = np.ascontiguousarray(vertex_boxes)
    Ec = _i32cm(E).reshape(-1, 2, order="F") if np.size(E) else np.zeros((0, 2), np.int32, order="F")
    out = np.zeros(Ec.shape[0], AABB_DTYPE)
    ctx._check(lib().sccd_build_edge_boxes(ctx._h, _ptr(vb), C.c_int(len(vb)), _ptr(Ec), C.c_int(Ec.shape[0]), _ptr(out)))
    return out


def build_face_boxes(vertex_boxes, F, ctx=None):
    ctx = ctx or default_context()
    vb = np.ascontiguousarray(vertex_boxes)
    Fc = _i32cm(F).reshape(-1, 3, order="F") if np.size(F) else np.zeros((0, 3), np.int32, order="F")
    out = np.zeros(Fc.shape[0], AABB_DTYPE)
    ctx._check(lib().sccd_build_face_boxes(ctx._h, _ptr(vb), C.c_int(len(vb)), _ptr(Fc), C.c_int(Fc.shape[0]), _ptr(out)))
    return out


class DeviceAABBs:
    """Sorted device boxes (DeviceAABBs of aabb.cuh:122-150)."""

    def __init__(self, boxes=None, ctx=None, _handle=None):
        self.ctx = ctx or default_context()
        self._h = C.c_void_p()
        self.ctx._children.add(self)
        if _handle is not None:
            self._h = _handle
            return
        b = np.ascontiguousarray(boxes, dtype=AABB_DTYPE)
        self.ctx._check(lib().sccd_boxes_create(self.ctx._h, _ptr(b), C.c_int(len(b)), C.c_int(0), C.byref(self._h)))

    @staticmethod
    def from_mesh(mesh, inflation_radius=0.0):
        """(vertex, edge, face) boxes built and sorted on the device (ccd.cu:112-121)."""
        v, e, f = C.c_void_p(), C.c_void_p(), C.c_void_p()
        mesh.ctx._check(lib().sccd_boxes_from_mesh(mesh.ctx._h, mesh._h, C.c_double(inflation_radius), C.byref(v), C.byref(e), C.byref(f)))
        return tuple(DeviceAABBs(ctx=mesh.ctx, _handle=h) for h in (v, e, f))

    def size(self):
        return int(lib().sccd_boxes_size(self._h))

    def __len__(self):
        return self.size()

    def download(self):
        out = np.zeros(self.size(), AABB_DTYPE)
        self.ctx._check(lib().sccd_boxes_download(self._h, _ptr(out)))
        return out

    def close(self):
        if self._h:
            lib().sccd_boxes_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BroadPhase:
    """class BroadPhase of broad_phase.cuh:15-92."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()
        self._h = C.c_void_p()
        self.ctx._check(lib().sccd_broad_phase_create(self.ctx._h, C.byref(self._h)))
        self._boxes = ()
        self.ctx._children.add(self)

    def build(self, boxes_a, boxes_b=None):
        self._boxes = (boxes_a, boxes_b)  # shared ownership, like the reference's shared_ptr
        self.ctx._check(lib().sccd_broad_phase_build(self._h, boxes_a._h, boxes_b._h if boxes_b is not None else None))

    def detect_overlaps_partial(self):
        """-> (device pointer, n): valid until the next call (broad_phase.cuh:41-44)."""
        p = C.c_void_p()
        n = C.c_int64()
        self.ctx._check(lib().sccd_broad_phase_detect_overlaps_partial(self._h, C.byref(p), C.byref(n)))
        return (p.value or 0), n.value

    def detect_overlaps(self):
        """-> int32[n, 2] host array (std::vector<std::pair<int,int>> of broad_phase.cuh:50)."""
        p = C.c_void_p()
        n = C.c_int64()
        self.ctx._check(lib().sccd_broad_phase_detect_overlaps(self._h, C.byref(p), C.byref(n)))
        if n.value == 0:
            lib().sccd_free(p)
            return np.zeros((0, 2), np.int32)
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(n.value, 2)).copy()
        lib().sccd_free(p)
        return arr

    def is_complete(self):
        return bool(lib().sccd_broad_phase_is_complete(self._h))

    def num_boxes(self):
        return int(lib().sccd_broad_phase_num_boxes(self._h))

    def candidates(self):
        return int(lib().sccd_broad_phase_candidates(self._h))

    def close(self):
        if self._h:
            lib().sccd_broad_phase_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sort_and_sweep(boxes, boxes_b=None, sort_axis=0, ctx=None):
    """The reference's CPU entry point (broad_phase/sort_and_sweep.hpp:28-42) on the device path:
    returns (pairs[n, 2] int32, next_sort_axis) -- one list: (min id, max id); two lists: (A id, B id);
    next_sort_axis = arg-max variance of the box centres (sort_and_sweep.cpp:176-195)."""
    ctx = ctx or default_context()
    if len(boxes) == 0 or (boxes_b is not None and len(boxes_b) == 0):
        return np.zeros((0, 2), np.int32), sort_axis
    saved = ctx.get_option(OPT_SORT_AXIS)
    ctx.set_option(OPT_SORT_AXIS, sort_axis)
    try:
        a = DeviceAABBs(boxes, ctx)
        b = DeviceAABBs(boxes_b, ctx) if boxes_b is not None else None
        bp = BroadPhase(ctx)
        bp.build(a, b)
        pairs = bp.detect_overlaps().reshape(-1, 2)
        ax = C.c_int(0)
        ctx._check(lib().sccd_boxes_variance_axis(ctx._h, a._h, b._h if b is not None else None, C.byref(ax)))
    finally:
        ctx.set_option(OPT_SORT_AXIS, saved)
    return pairs, ax.value


def sort_along_axis(axis, boxes):
    """sort_along_axis() of broad_phase/sort_and_sweep.hpp:11 (sort_and_sweep.cpp:126-141): the boxes ordered by min[axis]."""
    if axis not in (0, 1, 2):
        raise RuntimeError("sort_along_axis: axis must be 0, 1 or 2")
    b = np.asarray(boxes)
    return b[np.argsort(b["min"][:, axis], kind="stable")]


def sweep(boxes, sort_axis=0, two_lists=False, ctx=None):
    """sweep<is_two_lists>() of sort_and_sweep.hpp:18-22 (sort_and_sweep.cpp:143-195): one list, or two lists MERGED with the first
    list's element ids flipped to -id - 1 (:228-237).  -> (pairs, next_sort_axis); served by the device path."""
    b = np.asarray(boxes)
    if not two_lists:
        return sort_and_sweep(b, None, sort_axis, ctx)
    first = b[b["element_id"] < 0].copy()
    first["element_id"] = -first["element_id"] - 1
    return sort_and_sweep(first, b[b["element_id"] >= 0], sort_axis, ctx)


def narrow_phase(mesh, overlaps, is_vf, max_iter=-1, tol=1e-6, ms=0.0, allow_zero_toi=True, toi=1.0,
                 want_collisions=False, n=None):
    """narrow_phase<is_vf>() of narrow_phase.cuh:30-46.  overlaps: int32[n,2] host array, or a
    raw device pointer with n given.  Returns toi, or (toi, collisions) with want_collisions."""
    ctx = mesh.ctx
    t = C.c_double(toi)
    cp = C.c_void_p()
    cn = C.c_int64()
    if isinstance(overlaps, int):
        args = (C.c_void_p(overlaps), C.c_int64(int(n)), C.c_int(1))
    else:
        ov = np.ascontiguousarray(overlaps, dtype=np.int32).reshape(-1, 2)
        args = (_ptr(ov), C.c_int64(len(ov)), C.c_int(0))
    ctx._check(lib().sccd_narrow_phase(
        ctx._h, mesh._h, *args, C.c_int(int(is_vf)), C.c_int(max_iter), C.c_double(tol), C.c_double(ms),
        C.c_int(int(allow_zero_toi)), C.byref(t), C.byref(cp) if want_collisions else None,
        C.byref(cn) if want_collisions else None))
    if not want_collisions:
        return t.value
    if cn.value:
        col = np.frombuffer(C.string_at(cp, cn.value * COLLISION_DTYPE.itemsize), dtype=COLLISION_DTYPE).copy()
    else:
        col = np.zeros(0, COLLISION_DTYPE)
    lib().sccd_free(cp)
    return t.value, col


def query_cull(mesh, overlaps, is_vf, ms=0.0, tol=1e-6):
    """The projection cull on its own (sccd_query_cull): the pairs of `overlaps` (int32[n,2]) that MAY have an impact, in any order;
    the others provably have no domain the reference's bisection could accept (csrc/narrow_cull.inc)."""
    ctx = mesh.ctx
    ov = np.ascontiguousarray(overlaps, dtype=np.int32).reshape(-1, 2)
    kept = np.zeros_like(ov)
    k = C.c_int64()
    ctx._check(lib().sccd_query_cull(ctx._h, mesh._h, _ptr(ov), C.c_int64(len(ov)), C.c_int(int(is_vf)), C.c_double(ms), C.c_double(tol),
                                     _ptr(kept), C.byref(k)))
    return kept[: k.value].copy()


def query_cull_slab(mesh, overlaps, is_vf, ms=0.0, tol=1e-6, t_lo=0.0, t_hi=1.0):
    """The projection cull for one slab of time (sccd_query_cull_slab): the pairs that may have an accepted domain meeting
    t in [t_lo, t_hi] -- what a narrow launch asking about that part of the step has to look at."""
    ctx = mesh.ctx
    ov = np.ascontiguousarray(overlaps, dtype=np.int32).reshape(-1, 2)
    kept = np.zeros_like(ov)
    k = C.c_int64()
    ctx._check(lib().sccd_query_cull_slab(ctx._h, mesh._h, _ptr(ov), C.c_int64(len(ov)), C.c_int(int(is_vf)), C.c_double(ms), C.c_double(tol),
                                          C.c_double(t_lo), C.c_double(t_hi), _ptr(kept), C.byref(k)))
    return kept[: k.value].copy()


def ccd(V0, V1, E, F, min_distance=0.0, max_iterations=-1, tolerance=1e-6, allow_zero_toi=True, memory_limit_GB=0, ctx=None,
        want_collisions=False):
    """scalable_ccd::cuda::ccd() of ccd.cuh:26-38: earliest time of impact in [0,1] (1 = none).
    want_collisions: the SCALABLE_CCD_TOI_PER_QUERY signature -- returns (toi, collisions) with
    the (aid, bid, toi) records of the vertex-face pass followed by those of the edge-edge pass."""
    ctx = ctx or default_context()
    if want_collisions:
        return _ccd_collisions(V0, V1, E, F, min_distance, max_iterations, tolerance, allow_zero_toi, memory_limit_GB, ctx)
    V0c, V1c, Ec, Fc = _f64cm(V0), _f64cm(V1), _i32cm(E), _i32cm(F)
    if V0c.ndim != 2 or V0c.shape[1] != 3 or V0c.shape != V1c.shape:
        raise RuntimeError("V0, V1 must both be n x 3")
    nE = Ec.shape[0] if Ec.size else 0
    nF = Fc.shape[0] if Fc.size else 0
    if (nE and Ec.shape[1] != 2) or (nF and Fc.shape[1] != 3):
        raise RuntimeError("E must be m x 2 and F k x 3")
    t = C.c_double(1.0)
    ctx._check(lib().sccd_ccd(
        ctx._h, _ptr(V0c), _ptr(V1c), C.c_int(V0c.shape[0]), _ptr(Ec) if nE else None, C.c_int(nE),
        _ptr(Fc) if nF else None, C.c_int(nF), C.c_double(min_distance), C.c_int(max_iterations),
        C.c_double(tolerance), C.c_int(int(allow_zero_toi)), C.c_int(memory_limit_GB), C.byref(t)))
    return t.value


def _ccd_collisions(V0, V1, E, F, min_distance, max_iterations, tolerance, allow_zero_toi, memory_limit_GB, ctx):
    """ccd.cu:14-78 with the per-query list: ONE call into the library (sccd_ccd_collisions); the overlap pairs stay on
    the device between the broad and the narrow phase."""
    V0c, V1c, Ec, Fc = _f64cm(V0), _f64cm(V1), _i32cm(E), _i32cm(F)
    if V0c.ndim != 2 or V0c.shape[1] != 3 or V0c.shape != V1c.shape:
        raise RuntimeError("V0, V1 must both be n x 3")
    nE = Ec.shape[0] if Ec.size else 0
    nF = Fc.shape[0] if Fc.size else 0
    if (nE and Ec.shape[1] != 2) or (nF and Fc.shape[1] != 3):
        raise RuntimeError("E must be m x 2 and F k x 3")
    t = C.c_double(1.0)
    cp = C.c_void_p()
    cn = C.c_int64()
    ctx._check(lib().sccd_ccd_collisions(
        ctx._h, _ptr(V0c), _ptr(V1c), C.c_int(V0c.shape[0]), _ptr(Ec) if nE else None, C.c_int(nE),
        _ptr(Fc) if nF else None, C.c_int(nF), C.c_double(min_distance), C.c_int(max_iterations), C.c_double(tolerance),
        C.c_int(int(allow_zero_toi)), C.c_int(int(memory_limit_GB)), C.byref(t), C.byref(cp), C.byref(cn)))
    if cn.value:
        col = np.frombuffer(C.string_at(cp, cn.value * COLLISION_DTYPE.itemsize), dtype=COLLISION_DTYPE).copy()
    else:
        col = np.zeros(0, COLLISION_DTYPE)
    lib().sccd_free(cp)
    return t.value, col


def ccd_mesh(mesh, min_distance=0.0, max_iterations=-1, tolerance=1e-6, allow_zero_toi=True, want_stats=False):
    """ccd() on a device-resident mesh (what bench.py times).  -> toi or (toi, stats dict)."""
    t = C.c_double(1.0)
    st = Stats()
    mesh.ctx._check(lib().sccd_ccd_mesh(
        mesh.ctx._h, mesh._h, C.c_double(min_distance), C.c_int(max_iterations), C.c_double(tolerance),
        C.c_int(int(allow_zero_toi)), C.byref(t), C.byref(st) if want_stats else None))
    return (t.value, st.as_dict()) if want_stats else t.value


def ccd_mesh_from(mesh, bound, min_distance=0.0, max_iterations=-1, tolerance=1e-6, allow_zero_toi=True, want_stats=False):
    """ccd_mesh() that starts from the caller's bound in (0, 1] (sccd_ccd_mesh_from): min(bound, earliest impact below it); the context's
    own history is neither used nor kept.  -> toi or (toi, stats dict)."""
    t = C.c_double(1.0)
    st = Stats()
    mesh.ctx._check(lib().sccd_ccd_mesh_from(
        mesh.ctx._h, mesh._h, C.c_double(min_distance), C.c_int(max_iterations), C.c_double(tolerance),
        C.c_int(int(allow_zero_toi)), C.c_double(bound), C.byref(t), C.byref(st) if want_stats else None))
    return (t.value, st.as_dict()) if want_stats else t.value


def ccd_mesh_dev(mesh, d_toi, min_distance=0.0, max_iterations=-1, tolerance=1e-6, allow_zero_toi=True, want_stats=False):
    """ccd_mesh() whose result is ALSO left in device memory: `d_toi` is the address of a device double (e.g. a persistent
    torch tensor's data_ptr()) that receives the TOI by a copy on the context's stream -- what a multi-GPU caller all-reduces in
    place (sccd.dist.allreduce_min_device).  -> toi or (toi, stats dict), like ccd_mesh()."""
    t = C.c_double(1.0)
    st = Stats()
    mesh.ctx._check(lib().sccd_ccd_mesh_dev(
        mesh.ctx._h, mesh._h, C.c_double(min_distance), C.c_int(max_iterations), C.c_double(tolerance),
        C.c_int(int(allow_zero_toi)), C.c_void_p(int(d_toi)), C.byref(t), C.byref(st) if want_stats else None))
    return (t.value, st.as_dict()) if want_stats else t.value


def ccd_mesh_prepare(mesh, min_distance=0.0):
    mesh.ctx._check(lib().sccd_ccd_mesh_prepare(mesh.ctx._h, mesh._h, C.c_double(min_distance)))


def ccd_mesh_pass(mesh, is_vf, toi, min_distance=0.0, max_iterations=-1, tolerance=1e-6, allow_zero_toi=True):
    """One half of ccd() on this rank's shard.  -> (toi, stats dict)."""
    t = C.c_double(toi)
    st = Stats()
    mesh.ctx._check(lib().sccd_ccd_mesh_pass(
        mesh.ctx._h, mesh._h, C.c_int(int(is_vf)), C.c_double(min_distance), C.c_int(max_iterations),
        C.c_double(tolerance), C.c_int(int(allow_zero_toi)), C.byref(t), C.byref(st)))
    return t.value, st.as_dict()


def ipc_ccd_strategy(V0, V1, E, F, min_distance=0.0, max_iterations=-1, tolerance=1e-6, ctx=None):
    """ipc_ccd_strategy() of ipc_ccd_strategy.hpp:17-24."""
    ctx = ctx or default_context()
    V0c, V1c, Ec, Fc = _f64cm(V0), _f64cm(V1), _i32cm(E), _i32cm(F)
    t = C.c_double(1.0)
    ctx._check(lib().sccd_ipc_ccd_strategy(
        ctx._h, _ptr(V0c), _ptr(V1c), C.c_int(V0c.shape[0]), _ptr(Ec), C.c_int(Ec.shape[0]), _ptr(Fc),
        C.c_int(Fc.shape[0]), C.c_double(min_distance), C.c_int(max_iterations), C.c_double(tolerance), C.byref(t)))
    return t.value
