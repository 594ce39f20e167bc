"""ctypes binding of the CPU oracle (oracle/sccd_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product (scalable-ccd_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ARITH_STRICT = 0
ARITH_FMA = 1
# the contract callers get when they name none: the reference's CUDA build fuses (CMakeLists.txt:219-225 --use_fast_math => -fmad=true)
ARITH_DEFAULT = ARITH_FMA

AABB_DTYPE = np.dtype(
    [("min", "<f8", (3,)), ("max", "<f8", (3,)), ("vertex_ids", "<i4", (3,)), ("element_id", "<i4")],
    align=True,
)
assert AABB_DTYPE.itemsize == 64


class NPStats(C.Structure):
    _fields_ = [
        ("n_queries", C.c_int64),
        ("n_checks", C.c_int64),
        ("n_domains", C.c_int64),
        ("n_root_survive", C.c_int64),
        ("max_checks_per_query", C.c_int64),
        ("max_queue", C.c_int64),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build():
    """Compile the oracle with gcc (no GPU needed)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libsccd_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_sort_and_sweep.restype = C.c_int64
        L.orc_sort_and_sweep_two_lists.restype = C.c_int64
        L.orc_brute_force.restype = C.c_int64
        L.orc_last_candidate_tests.restype = C.c_int64
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64cm(M):
    """column-major float64 copy (Eigen::MatrixXd storage)"""
    return np.asfortranarray(np.asarray(M, dtype=np.float64))


def _i32cm(M):
    return np.asfortranarray(np.asarray(M, dtype=np.int32))


# scalar="f32": the SCALABLE_CCD_USE_DOUBLE = OFF twin (Scalar = float): vertices are cast to float first
def _rcm(M, scalar):
    return np.asfortranarray(np.asarray(M, dtype=np.float32 if scalar == "f32" else np.float64))


def _rty(scalar):
    return (C.c_float, np.float32, "_f32") if scalar == "f32" else (C.c_double, np.float64, "")


def build_boxes(V0, V1, E, F, inflation=0.0, scalar="f64"):
    """(vertex_boxes, edge_boxes, face_boxes) as structured arrays (64-byte cuda::AABB layout)."""
    L = lib()
    c_real, _, sfx = _rty(scalar)
    V0c, V1c, Ec, Fc = _rcm(V0, scalar), _rcm(V1, scalar), _i32cm(E), _i32cm(F)
    nV, nE, nF = V0c.shape[0], Ec.shape[0], Fc.shape[0]
    vb = np.zeros(nV, AABB_DTYPE)
    eb = np.zeros(nE, AABB_DTYPE)
    fb = np.zeros(nF, AABB_DTYPE)
    getattr(L, "orc_build_vertex_boxes" + sfx)(_p(V0c), _p(V1c), C.c_int(nV), c_real(inflation), _p(vb))
    L.orc_build_edge_boxes(_p(vb), _p(Ec), C.c_int(nE), _p(eb))
    L.orc_build_face_boxes(_p(vb), _p(Fc), C.c_int(nF), _p(fb))
    return vb, eb, fb


def _take_pairs(L, ptr, n, sort):
    if n == 0:
        if ptr:
            L.orc_free(ptr)
        return np.zeros((0, 2), np.int32)
    if sort:
        L.orc_sort_pairs(ptr, C.c_int64(n))
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=(n, 2)).copy()
    L.orc_free(ptr)
    return arr


def sort_and_sweep(boxes, boxes_b=None, sort_axis=0, nthreads=1, sort=True):
    """Reference CPU broad phase.  Returns (pairs[n,2] int32, next_sort_axis, candidate_tests)."""
    L = lib()
    ax = C.c_int(sort_axis)
    ptr = C.c_void_p()
    boxes = np.ascontiguousarray(boxes)
    if boxes_b is None:
        n = L.orc_sort_and_sweep(_p(boxes), C.c_int(len(boxes)), C.byref(ax), C.byref(ptr), C.c_int(nthreads))
    else:
        boxes_b = np.ascontiguousarray(boxes_b)
        n = L.orc_sort_and_sweep_two_lists(
            _p(boxes), C.c_int(len(boxes)), _p(boxes_b), C.c_int(len(boxes_b)), C.byref(ax), C.byref(ptr), C.c_int(nthreads)
        )
    tests = int(L.orc_last_candidate_tests())
    return _take_pairs(L, ptr, n, sort), ax.value, tests


def brute_force(boxes, boxes_b=None):
    L = lib()
    ptr = C.c_void_p()
    boxes = np.ascontiguousarray(boxes)
    if boxes_b is None:
        n = L.orc_brute_force(_p(boxes), C.c_int(len(boxes)), None, C.c_int(0), C.byref(ptr))
    else:
        boxes_b = np.ascontiguousarray(boxes_b)
        n = L.orc_brute_force(_p(boxes), C.c_int(len(boxes)), _p(boxes_b), C.c_int(len(boxes_b)), C.byref(ptr))
    return _take_pairs(L, ptr, n, True)


def narrow_phase(V0, V1, E, F, pairs, is_vf, ms=0.0, max_iter=-1, tol=1e-6, allow_zero_toi=True,
                 arith=ARITH_DEFAULT, toi=1.0, per_query=False, scalar="f64"):
    """Level-synchronous restatement.  Returns (toi, per_query_toi|None, stats dict)."""
    L = lib()
    c_real, np_real, sfx = _rty(scalar)
    V0c, V1c, Ec, Fc = _rcm(V0, scalar), _rcm(V1, scalar), _i32cm(E), _i32cm(F)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    t = c_real(toi)
    st = NPStats()
    pq = np.full(len(pairs), np.inf, dtype=np_real) if per_query else None
    rc = getattr(L, "orc_narrow_phase" + sfx)(
        _p(V0c), _p(V1c), C.c_int(V0c.shape[0]), _p(Ec), C.c_int(Ec.shape[0]), _p(Fc), C.c_int(Fc.shape[0]),
        _p(pairs), C.c_int64(len(pairs)), C.c_int(int(is_vf)), c_real(ms), C.c_int(max_iter), c_real(tol),
        C.c_int(int(allow_zero_toi)), C.c_int(arith), C.byref(t), _p(pq) if per_query else None, C.byref(st),
    )
    if rc != 0:  # ORC_E_BUDGET: a level of the (level-order) restatement outgrew its domain budget
        raise MemoryError("oracle narrow phase: level exceeds the live-domain budget (contact-rich queries in level order)")
    return t.value, pq, st.as_dict()


def narrow_phase_mt(V0, V1, E, F, pairs, is_vf, ms=0.0, max_iter=-1, tol=1e-6, allow_zero_toi=True,
                    arith=ARITH_DEFAULT, toi=1.0, nthreads=1, want_checks=False, scalar="f64"):
    L = lib()
    c_real, _, sfx = _rty(scalar)
    V0c, V1c, Ec, Fc = _rcm(V0, scalar), _rcm(V1, scalar), _i32cm(E), _i32cm(F)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    t = c_real(toi)
    chk = np.zeros(len(pairs), np.int32) if want_checks else None
    getattr(L, "orc_narrow_phase_mt" + sfx)(
        _p(V0c), _p(V1c), C.c_int(V0c.shape[0]), _p(Ec), C.c_int(Ec.shape[0]), _p(Fc), C.c_int(Fc.shape[0]),
        _p(pairs), C.c_int64(len(pairs)), C.c_int(int(is_vf)), c_real(ms), C.c_int(max_iter), c_real(tol),
        C.c_int(int(allow_zero_toi)), C.c_int(arith), C.byref(t), _p(chk) if want_checks else None, C.c_int(nthreads),
    )
    return t.value, chk


def ccd(V0, V1, E, F, ms=0.0, max_iter=-1, tol=1e-6, allow_zero_toi=True, arith=ARITH_DEFAULT, nthreads=1, scalar="f64"):
    """Restatement of scalable_ccd::cuda::ccd (ccd.cu:80-146).  Returns (toi, n_vf, n_ee)."""
    L = lib()
    c_real, _, sfx = _rty(scalar)
    V0c, V1c, Ec, Fc = _rcm(V0, scalar), _rcm(V1, scalar), _i32cm(E), _i32cm(F)
    t = c_real(1.0)
    nvf, nee = C.c_int64(0), C.c_int64(0)
    rc = getattr(L, "orc_ccd" + sfx)(
        _p(V0c), _p(V1c), C.c_int(V0c.shape[0]), _p(Ec), C.c_int(Ec.shape[0]), _p(Fc), C.c_int(Fc.shape[0]),
        c_real(ms), C.c_int(max_iter), c_real(tol), C.c_int(int(allow_zero_toi)), C.c_int(arith),
        C.c_int(nthreads), C.byref(t), C.byref(nvf), C.byref(nee),
    )
    if rc != 0:
        raise MemoryError("oracle ccd: a level of the serial (level-order) narrow phase exceeds the live-domain budget")
    return t.value, nvf.value, nee.value


def ipc_ccd_strategy(V0, V1, E, F, ms=0.0, max_iter=-1, tol=1e-6, arith=ARITH_DEFAULT, want_branches=False):
    """Restatement of scalable_ccd::cuda::ipc_ccd_strategy (ipc_ccd_strategy.cu:12-152), one chunk per pass
    (the reference's `while (!broad_phase.is_complete())` runs once when the overlaps fit its buffer).
    Returns earliest_toi (and, on request, which passes took the conservative re-run branch :72-91)."""
    vb, eb, fb = build_boxes(V0, V1, E, F, ms)  # :123-125: inflation radius = min_distance
    earliest = 1.0  # :136
    reran = []
    for run_vf in (True, False):  # :138-148
        pairs = sort_and_sweep(vb, fb)[0] if run_vf else sort_and_sweep(eb)[0]
        before = earliest  # :52
        earliest = narrow_phase(V0, V1, E, F, pairs, run_vf, ms, max_iter, tol, True, arith, toi=earliest)[0]  # :61-69
        if earliest < 1e-6:  # :72
            earliest = before  # :76
            # :79-87: max_iterations = -1, ms = 0, allow_zero_toi = false
            earliest = narrow_phase(V0, V1, E, F, pairs, run_vf, 0.0, -1, tol, False, arith, toi=earliest)[0]
            earliest *= 0.8  # :88
            reran.append(run_vf)
    return (earliest, reran) if want_branches else earliest


def query_constants(v24, is_vf, use_ms, tol):
    L = lib()
    v = np.ascontiguousarray(v24, dtype=np.float64).reshape(24)
    t3, e3 = np.zeros(3), np.zeros(3)
    L.orc_query_constants(_p(v), C.c_int(int(is_vf)), C.c_int(int(use_ms)), C.c_double(tol), _p(t3), _p(e3))
    return t3, e3


def inclusion(v24, dom6, err3, ms, is_vf, arith=ARITH_DEFAULT):
    L = lib()
    v = np.ascontiguousarray(v24, dtype=np.float64).reshape(24)
    d = np.ascontiguousarray(dom6, dtype=np.float64).reshape(6)
    e = np.ascontiguousarray(err3, dtype=np.float64).reshape(3)
    tt = C.c_double(0)
    bi = C.c_int(0)
    r = L.orc_origin_in_inclusion_function(_p(v), _p(d), _p(e), C.c_double(ms), C.c_int(int(is_vf)), C.c_int(arith), C.byref(tt), C.byref(bi))
    return bool(r), tt.value, bool(bi.value)
