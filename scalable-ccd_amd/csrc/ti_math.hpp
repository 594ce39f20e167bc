// ti_math.hpp -- Tight-Inclusion arithmetic shared by the narrow-phase kernels.
//
// Restates, operation for operation, src/scalable_ccd/cuda/narrow_phase/root_finder.cu:21-254
// and narrow_phase.cu:24-74 of the reference.  Compiled with -ffp-contract=off: the only fused
// multiply-adds are the explicit __builtin_fma calls of ARITH == 1 (DESIGN.md "Arithmetic
// contract"); ARITH == 0 rounds every product and sum separately in source order.
//
// Host-compilable: tests/cpp/test_ti_host.cpp builds this header with g++ and checks ti_step / nq_step operation
// for operation against the CPU oracle (the arithmetic is plain C++; only the gathers need HIP vector types).
#pragma once
#if defined(__HIPCC__)
#include "common.hpp"
#else
#include <algorithm>
#include <cmath>
#include <cstring>
#if !defined(TIF_HOST_DEFS)
#define TIF_HOST_DEFS
#define __device__
#define __forceinline__ inline
#endif
static inline long long __double_as_longlong(double x)
{
    long long b;
    std::memcpy(&b, &x, 8);
    return b;
}
using std::fabs;
using std::ldexp;
using std::min;
#endif

#define TI_DBL_MAX 1.7976931348623157e308
#define TI_DBL_EPS 2.220446049250313e-16

// v[0..3] = v0s..v3s (t = 0), v[4..7] = v0e..v3e (t = 1)        (ccd_data.cuh:8-26)
struct TIQuery {
    double v[8][3];
    double err[3];
    double tol[3];
    // np_walk_k only: fl(1 / tol[k]) and whether the reciprocal shortcut below is exact for it
    double inv_tol[3];
    bool inv_ok;
};

// split_dimension compares w_k / tol_k (root_finder.cu:202).  For a bisected interval w_k is a
// power of two, and scaling by a power of two never changes a rounding:
//   fl(w_k / tol_k) == w_k * fl(1 / tol_k)      (no overflow/underflow)
// so the three divisions per check become three multiplications by per-query constants, with
// the same bits.  Guard: every reciprocal is 0, +inf or within [2^-500, 2^500].
__device__ __forceinline__ bool ti_inv_tol_ok_one(double inv_tol_k)
{
    const double a = fabs(inv_tol_k);
    return a == 0.0 || a == __builtin_huge_val() || (a >= 0x1p-500 && a <= 0x1p500);
}
__device__ __forceinline__ bool ti_inv_tol_ok(const double inv_tol[3])
{
    return ti_inv_tol_ok_one(inv_tol[0]) && ti_inv_tol_ok_one(inv_tol[1]) && ti_inv_tol_ok_one(inv_tol[2]);
}
__device__ __forceinline__ void ti_prepare_inv_tol(TIQuery& q)
{
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double r = 1.0 / q.tol[k];
        q.inv_tol[k] = r;
        const double a = fabs(r);
        ok = ok && (a == 0.0 || a == __builtin_huge_val() || (a >= 0x1p-500 && a <= 0x1p500));
    }
    q.inv_ok = ok;
}

// Eigen's Array::min/max (root_finder.cu:178-179) select with a compare; v_min_f64 / v_max_f64
// return the same VALUE for non-NaN operands in one instruction instead of three (a different
// sign of zero is the only possible difference, and nothing downstream distinguishes +-0).
__device__ __forceinline__ double ti_min(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double ti_max(double a, double b) { return __builtin_fmax(a, b); }

#if defined(__HIPCC__)
// add_data<is_vf> (narrow_phase.cu:24-74) on the packed mesh, in two steps so that the narrow-phase
// kernel can software-pipeline them: the four vertex ids of a query ...
template <bool VF>
__device__ __forceinline__ int4 ti_indices(const int2* __restrict__ E, const int4* __restrict__ F, int2 pr)
{
    if (VF) { // :41-53  v0 = vertex, v1..v3 = face corners
        const int4 f = F[pr.y];
        return make_int4(pr.x, f.x, f.y, f.z);
    } else { // :54-66  v0,v1 = edge a, v2,v3 = edge b
        const int2 ea = E[pr.x], eb = E[pr.y];
        return make_int4(ea.x, ea.y, eb.x, eb.y);
    }
}
// ... and the gather of their 48-byte records {x0,y0,z0,x1,y1,z1}
__device__ __forceinline__ void ti_gather_ids(const double* __restrict__ V, int4 ids, double v[8][3])
{
    const int id[4] = { ids.x, ids.y, ids.z, ids.w };
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double2* rec = reinterpret_cast<const double2*>(V) + 3 * (size_t)id[k];
        const double2 a = rec[0], b = rec[1], c = rec[2];
        v[k][0] = a.x;
        v[k][1] = a.y;
        v[k][2] = b.x;
        v[k + 4][0] = b.y;
        v[k + 4][1] = c.x;
        v[k + 4][2] = c.y;
    }
}
template <bool VF>
__device__ __forceinline__ void ti_gather(const double* __restrict__ V, const int2* __restrict__ E,
                                          const int4* __restrict__ F, int2 pr, double v[8][3])
{
    ti_gather_ids(V, ti_indices<VF>(E, F, pr), v);
}

#endif // __HIPCC__

__device__ __forceinline__ double ti_linf(const double a[3], const double b[3])
{
    double m = fabs(b[0] - a[0]);
    m = ti_max(m, fabs(b[1] - a[1]));
    m = ti_max(m, fabs(b[2] - a[2]));
    return m;
}
// max_Linf_4 (root_finder.cu:31-46)
__device__ __forceinline__ double ti_max_linf_4(const double* p1, const double* p2, const double* p3,
                                                const double* p4, const double* p1e, const double* p2e,
                                                const double* p3e, const double* p4e)
{
    return ti_max(ti_max(ti_linf(p1, p1e), ti_linf(p2, p2e)), ti_max(ti_linf(p3, p3e), ti_linf(p4, p4e)));
}

// compute_face_vertex_tolerance / compute_edge_edge_tolerance (root_finder.cu:48-88)
template <bool VF> __device__ __forceinline__ void ti_tolerance(const double v[8][3], double co_domain_tol, double tol[3])
{
    double p000[3], p001[3], p011[3], p010[3], p100[3], p101[3], p111[3], p110[3];
    if (VF) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            p000[k] = v[0][k] - v[1][k];
            p001[k] = v[0][k] - v[3][k];
            p011[k] = v[0][k] - (v[2][k] + v[3][k] - v[1][k]);
            p010[k] = v[0][k] - v[2][k];
            p100[k] = v[4][k] - v[5][k];
            p101[k] = v[4][k] - v[7][k];
            p111[k] = v[4][k] - (v[6][k] + v[7][k] - v[5][k]);
            p110[k] = v[4][k] - v[6][k];
        }
        tol[0] = co_domain_tol / (3 * ti_max_linf_4(p000, p001, p011, p010, p100, p101, p111, p110));
        tol[1] = co_domain_tol / (3 * ti_max_linf_4(p000, p100, p101, p001, p010, p110, p111, p011));
        tol[2] = co_domain_tol / (3 * ti_max_linf_4(p000, p100, p110, p010, p001, p101, p111, p011));
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            p000[k] = v[0][k] - v[2][k];
            p001[k] = v[0][k] - v[3][k];
            p010[k] = v[1][k] - v[2][k];
            p011[k] = v[1][k] - v[3][k];
            p100[k] = v[4][k] - v[6][k];
            p101[k] = v[4][k] - v[7][k];
            p110[k] = v[5][k] - v[6][k];
            p111[k] = v[5][k] - v[7][k];
        }
        // :82-87 -- tol[1] repeats the tol[0] pairing in the reference ("differs from
        // Tight-Inclusion", :71-72); reproduced as is.
        tol[0] = co_domain_tol / (3 * ti_max_linf_4(p000, p001, p011, p010, p100, p101, p111, p110));
        tol[1] = tol[0];
        tol[2] = co_domain_tol / (3 * ti_max_linf_4(p000, p100, p101, p001, p010, p110, p111, p011));
    }
}

// The same tolerances and error bounds ONE COORDINATE AT A TIME, for callers that cannot afford
// all 24 coordinates plus the eight differences in registers at once (np_walk_k's ingest reads
// the coordinates back from LDS).  x[j] = v[j][k]; m[0..2] accumulate the three max_Linf_4 values
// over k (max is exact, so the order of the maxima does not change a bit), start them at 0.
template <bool VF> __device__ __forceinline__ void ti_tolerance_dim(const double x[8], double m[3])
{
    double p000, p001, p011, p010, p100, p101, p111, p110;
    if (VF) {
        p000 = x[0] - x[1];
        p001 = x[0] - x[3];
        p011 = x[0] - (x[2] + x[3] - x[1]);
        p010 = x[0] - x[2];
        p100 = x[4] - x[5];
        p101 = x[4] - x[7];
        p111 = x[4] - (x[6] + x[7] - x[5]);
        p110 = x[4] - x[6];
    } else {
        p000 = x[0] - x[2];
        p001 = x[0] - x[3];
        p010 = x[1] - x[2];
        p011 = x[1] - x[3];
        p100 = x[4] - x[6];
        p101 = x[4] - x[7];
        p110 = x[5] - x[6];
        p111 = x[5] - x[7];
    }
    // pairings of ti_tolerance's three max_Linf_4 calls
    const double ma = ti_max(ti_max(fabs(p100 - p000), fabs(p101 - p001)), ti_max(fabs(p111 - p011), fabs(p110 - p010)));
    const double mb = ti_max(ti_max(fabs(p010 - p000), fabs(p110 - p100)), ti_max(fabs(p111 - p101), fabs(p011 - p001)));
    m[0] = ti_max(m[0], ma);
    if (VF) {
        const double mc = ti_max(ti_max(fabs(p001 - p000), fabs(p101 - p100)), ti_max(fabs(p111 - p110), fabs(p011 - p010)));
        m[1] = ti_max(m[1], mb);
        m[2] = ti_max(m[2], mc);
    } else {
        m[2] = ti_max(m[2], mb);
    }
}
template <bool VF> __device__ __forceinline__ void ti_tolerance_finish(const double m[3], double co_domain_tol, double tol[3])
{
    tol[0] = co_domain_tol / (3 * m[0]);
    if (VF) {
        tol[1] = co_domain_tol / (3 * m[1]);
        tol[2] = co_domain_tol / (3 * m[2]);
    } else {
        tol[1] = tol[0];
        tol[2] = co_domain_tol / (3 * m[2]);
    }
}
template <bool VF> __device__ __forceinline__ double ti_error_dim(const double x[8], bool use_ms)
{
    double filter;
    if (!use_ms) filter = VF ? 6.661338147750939e-15 : 6.217248937900877e-15;
    else filter = VF ? 7.549516567451064e-15 : 7.105427357601002e-15;
    double m = fabs(x[0]);
#pragma unroll
    for (int j = 1; j < 8; j++) m = ti_max(m, fabs(x[j]));
    m = ti_max(m, 1.0);
    return m * m * m * filter;
}

// get_numerical_error (root_finder.cu:90-135), double constants
template <bool VF> __device__ __forceinline__ void ti_error(const double v[8][3], bool use_ms, double err[3])
{
    double filter;
    if (!use_ms) filter = VF ? 6.661338147750939e-15 : 6.217248937900877e-15;
    else filter = VF ? 7.549516567451064e-15 : 7.105427357601002e-15;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double m = fabs(v[0][k]);
#pragma unroll
        for (int j = 1; j < 8; j++) m = ti_max(m, fabs(v[j][k]));
        m = ti_max(m, 1.0);
        err[k] = m * m * m * filter;
    }
}

// origin_in_inclusion_function (root_finder.cu:157-198) with calculate_vf / calculate_ee
// (:137-155).  The reference evaluates all eight corners from scratch; the values that do not
// depend on u or v are computed once per t here -- the same operations on the same operands,
// hence the same bits.  DIFF: v[4..7] already hold the displacements v_e - v_s (np_walk_k computes
// them once per query instead of once per check).
template <bool VF, int ARITH, bool DIFF = false>
__device__ __forceinline__ bool ti_inclusion(const double v[8][3], const double lo[3], const double hi[3],
                                             const double err[3], double ms, double& true_tol, bool& box_in)
{
    double cmin[3], cmax[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double mn = TI_DBL_MAX, mx = -TI_DBL_MAX;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const double t = it ? hi[0] : lo[0];
            double a0, a1, a2, a3; // the four vertices at time t
            const double e0 = DIFF ? v[4][k] : v[4][k] - v[0][k], e1 = DIFF ? v[5][k] : v[5][k] - v[1][k],
                         e2 = DIFF ? v[6][k] : v[6][k] - v[2][k], e3 = DIFF ? v[7][k] : v[7][k] - v[3][k];
            if (ARITH == 1) {
                a0 = __builtin_fma(e0, t, v[0][k]);
                a1 = __builtin_fma(e1, t, v[1][k]);
                a2 = __builtin_fma(e2, t, v[2][k]);
                a3 = __builtin_fma(e3, t, v[3][k]);
            } else {
                a0 = e0 * t + v[0][k];
                a1 = e1 * t + v[1][k];
                a2 = e2 * t + v[2][k];
                a3 = e3 * t + v[3][k];
            }
            if (VF) { // v - (t1 - t0)*u - (t2 - t0)*v - t0   with v=a0, t0=a1, t1=a2, t2=a3
                const double d1 = a2 - a1, d2 = a3 - a1;
#pragma unroll
                for (int iu = 0; iu < 2; iu++) {
                    const double u = iu ? hi[1] : lo[1];
                    const double r1 = (ARITH == 1) ? __builtin_fma(-d1, u, a0) : a0 - d1 * u;
#pragma unroll
                    for (int iw = 0; iw < 2; iw++) {
                        const double w = iw ? hi[2] : lo[2];
                        const double r2 = (ARITH == 1) ? __builtin_fma(-d2, w, r1) : r1 - d2 * w;
                        const double c = r2 - a1;
                        mn = ti_min(mn, c);
                        mx = ti_max(mx, c);
                    }
                }
            } else { // ((ea1 - ea0)*u + ea0) - ((eb1 - eb0)*v + eb0)
                const double da = a1 - a0, db = a3 - a2;
                double x[2], y[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const double u = i ? hi[1] : lo[1];
                    const double w = i ? hi[2] : lo[2];
                    x[i] = (ARITH == 1) ? __builtin_fma(da, u, a0) : da * u + a0;
                    y[i] = (ARITH == 1) ? __builtin_fma(db, w, a2) : db * w + a2;
                }
#pragma unroll
                for (int iu = 0; iu < 2; iu++)
#pragma unroll
                    for (int iw = 0; iw < 2; iw++) {
                        const double c = x[iu] - y[iw];
                        mn = ti_min(mn, c);
                        mx = ti_max(mx, c);
                    }
            }
        }
        cmin[k] = mn;
        cmax[k] = mx;
    }
    double wdt = cmax[0] - cmin[0];
    wdt = ti_max(wdt, cmax[1] - cmin[1]);
    wdt = ti_max(wdt, cmax[2] - cmin[2]);
    true_tol = ti_max(0.0, wdt); // :183
    box_in = true;
    bool out = false, notin = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        out = out || (cmin[k] - ms > err[k]) || (cmax[k] + ms < -err[k]);   // :187-190
        notin = notin || (cmin[k] + ms < -err[k]) || (cmax[k] - ms > err[k]); // :192-195
    }
    if (out) return false;
    box_in = !notin;
    return true;
}

// One ccd_kernel invocation (root_finder.cu:277-370) without the queue mechanics.
// Returns true if the domain is accepted (min_t is a TOI candidate).  nk = 0,1,2 children:
// child 0 = first half, child 1 = second half of dimension `split` at `mid`.
struct TIStep {
    bool accept;
    int nk;
    int split;
    double mid;
    bool checked;
};
template <bool VF, int ARITH, bool INVTOL = false>
__device__ __forceinline__ TIStep ti_step(const TIQuery& q, const double lo[3], const double hi[3], double ms,
                                          double co_domain_tol, bool allow_zero_toi, double prune_toi)
{
    TIStep r;
    r.accept = false;
    r.nk = 0;
    r.split = 0;
    r.mid = 0;
    r.checked = false;
    const double min_t = lo[0];
    if (min_t >= prune_toi) return r; // :295
    double true_tol;
    bool box_in;
    r.checked = true;
    if (!ti_inclusion<VF, ARITH>(q.v, lo, hi, q.err, ms, true_tol, box_in)) return r;
    const double w0 = hi[0] - lo[0], w1 = hi[1] - lo[1], w2 = hi[2] - lo[2];
    const bool zero_ok = allow_zero_toi || min_t > 0;
    if ((w0 <= q.tol[0] && w1 <= q.tol[1] && w2 <= q.tol[2])   // Condition 1 :322
        || (box_in && zero_ok)                                 // Condition 2 :331
        || (true_tol <= co_domain_tol && zero_ok)) {           // Condition 3 :340
        r.accept = true;
        return r;
    }
    // split_dimension :200-211
    double r0, r1, r2;
    bool fast = false;
    if (INVTOL) { // all widths exact powers of two >= 2^-200 (mantissa bits zero, biased exponent >= 823)
        const unsigned long long b0 = (unsigned long long)__double_as_longlong(w0),
                                 b1 = (unsigned long long)__double_as_longlong(w1),
                                 b2 = (unsigned long long)__double_as_longlong(w2);
        const unsigned long long emin = min(min(b0, b1), b2) >> 52; // widths are positive here
        fast = q.inv_ok && ((b0 | b1 | b2) & 0xFFFFFFFFFFFFFull) == 0ull && emin >= 823ull;
    }
    if (fast) {
        r0 = w0 * q.inv_tol[0];
        r1 = w1 * q.inv_tol[1];
        r2 = w2 * q.inv_tol[2];
    } else {
        r0 = w0 / q.tol[0];
        r1 = w1 / q.tol[1];
        r2 = w2 / q.tol[2];
    }
    int split;
    if (r0 >= r1 && r0 >= r2) split = 0;
    else if (r1 >= r0 && r1 >= r2) split = 1;
    else split = 2;
    // bisect :213-254, SplitInterval interval.cuh:18-28
    const double slo = split == 0 ? lo[0] : (split == 1 ? lo[1] : lo[2]);
    const double shi = split == 0 ? hi[0] : (split == 1 ? hi[1] : hi[2]);
    const double mid = (slo + shi) / 2;
    if (slo >= mid || mid >= shi) { // Condition 4 :222-225, :362
        r.accept = true;
        return r;
    }
    r.split = split;
    r.mid = mid;
    r.nk = 1;
    bool second;
    if (split == 0) second = mid <= prune_toi; // :229-232
    else if (VF) {
        // sum_less_than_one (:21-29): u + v <= 1 / (1 - DBL_EPSILON)
        const double other = (split == 1) ? lo[2] : lo[1];
        second = (mid + other) <= 1 / (1 - TI_DBL_EPS);
    } else second = true; // :248-250
    if (second) r.nk = 2;
    return r;
}

// ---- np_walk_k's domain entries and step (used by narrow_walk.inc) ---------------------------------------------
// A DOMAIN ENTRY.  Every interval the bisection produces is [k 2^-d, (k + 1) 2^-d] (root [0, 1]: k = 0,
// d = 0; halves of (k, d): (2k, d + 1) and (2k + 1, d + 1)), and all of those numbers -- like the
// reference's mid = (lo + hi) / 2 (interval.cuh:21) -- are exact in double.  The kernel therefore keeps
// a domain as three u32 numerators + three 8-bit levels (16 bytes: registers, LDS stack, HBM spill and
// hand-over all hold the same thing), bisects in integers, and only the inclusion check sees doubles:
// the same bits as carrying (lo, hi) pairs around, a quarter of the registers, no conversions on push / pop.
// Levels stop at NQ_MAX_LEVEL (2^-31 is four orders of magnitude below any tolerance in use; deeper => NQ_OVF_INTERVAL
// and the caller falls back to the level-synchronous kernels).
struct NQDom {
    unsigned k0, k1, k2, d; // d = d_t | d_u << 8 | d_v << 16
};
__device__ __forceinline__ void nq_bounds(unsigned k, unsigned d, double& lo, double& hi, double& w)
{
    const int e = -(int)d;
    lo = ldexp((double)k, e);
    hi = ldexp((double)(k + 1u), e);
    w = ldexp(1.0, e);
}

// One ccd_kernel invocation (root_finder.cu:277-370) on a domain entry: ti_step of ti_math.hpp with the
// interval arithmetic done in integers.  nk = 0, 1, 2 children: first half always, second half per
// bisect() (root_finder.cu:213-254).  Condition 4 (:222-225, an empty half) cannot occur above 2^-53.
struct NQStep {
    bool accept, checked;
    int nk, split;
    double min_t;
};
template <bool VF, int ARITH>
__device__ __forceinline__ NQStep nq_step(const TIQuery& q, const NQDom& dm, double ms, double co_domain_tol,
                                          bool allow_zero_toi, double prune_toi)
{
    NQStep r;
    r.accept = false;
    r.checked = false;
    r.nk = 0;
    r.split = 0;
    double lo[3], hi[3], w[3];
    nq_bounds(dm.k0, dm.d & 255u, lo[0], hi[0], w[0]);
    nq_bounds(dm.k1, (dm.d >> 8) & 255u, lo[1], hi[1], w[1]);
    nq_bounds(dm.k2, (dm.d >> 16) & 255u, lo[2], hi[2], w[2]);
    const double min_t = lo[0];
    r.min_t = min_t;
    if (min_t >= prune_toi) return r; // :295
    double true_tol;
    bool box_in;
    r.checked = true;
    if (!ti_inclusion<VF, ARITH, true>(q.v, lo, hi, q.err, ms, true_tol, box_in)) return r;
    const bool zero_ok = allow_zero_toi || min_t > 0;
    if ((w[0] <= q.tol[0] && w[1] <= q.tol[1] && w[2] <= q.tol[2]) // Condition 1 :322
        || (box_in && zero_ok)                                     // Condition 2 :331
        || (true_tol <= co_domain_tol && zero_ok)) {               // Condition 3 :340
        r.accept = true;
        return r;
    }
    // split_dimension :200-211 (widths are powers of two: ti_math.hpp on the reciprocal shortcut)
    double r0, r1, r2;
    if (q.inv_ok) {
        r0 = w[0] * q.inv_tol[0];
        r1 = w[1] * q.inv_tol[1];
        r2 = w[2] * q.inv_tol[2];
    } else {
        r0 = w[0] / q.tol[0];
        r1 = w[1] / q.tol[1];
        r2 = w[2] / q.tol[2];
    }
    int split;
    if (r0 >= r1 && r0 >= r2) split = 0;
    else if (r1 >= r0 && r1 >= r2) split = 1;
    else split = 2;
    r.split = split;
    r.nk = 1;
    // mid = (lo + hi) / 2 = lo + w / 2, exact
    const double slo = split == 0 ? lo[0] : (split == 1 ? lo[1] : lo[2]);
    const double sw = split == 0 ? w[0] : (split == 1 ? w[1] : w[2]);
    const double mid = slo + 0.5 * sw;
    bool second;
    if (split == 0) second = mid <= prune_toi; // :229-232
    else if (VF) {
        // sum_less_than_one (:21-29): u + v <= 1 / (1 - DBL_EPSILON)
        const double other = (split == 1) ? lo[2] : lo[1];
        second = (mid + other) <= 1 / (1 - TI_DBL_EPS);
    } else second = true; // :248-250
    if (second) r.nk = 2;
    return r;
}


// ---- stackless depth-first walk (np_walk_k) ---------------------------------------------------------------------
// WHICH dimension a domain is split in (split_dimension, root_finder.cu:200-211) depends on its three widths and
// the query's tolerances only -- not on the inclusion check.  By induction the level triple (d_t, d_u, d_v) of a
// node depends on its DEPTH n = d_t + d_u + d_v alone: the bisection tree of a query is a complete binary tree in
// which every node of depth n is split in the same dimension s_n.  A depth-first walk of such a tree needs no
// stack of deferred halves:
//   S0, S1   bit n set <=> s_n = 0 / 1 (neither: 2).  Written on the way down (idempotent), valid for every path.
//   pend     bit n set <=> on the CURRENT root-to-node path the split at depth n was left towards the first half
//            and its second half is still to be visited.
// Going from the current node (depth n) to the deepest pending second half (split depth j = top bit of pend):
// undo the splits of depths j+1 .. n-1 -- their number per dimension is a population count of S0 / S1 over those
// bits, and undoing c splits of a dimension is `numerator >>= c` -- then set the low bit of the numerator of
// dimension s_j.  Handing work to another lane gives away the SHALLOWEST pending half (low bit of pend: the largest
// subtree), computed the same way without moving.  Levels stop at 31 per dimension (NQ_MAX_LEVEL): depth <= 93 < 96.
// The second half of a split is registered when the reference would push it (bisect(), root_finder.cu:229-250).
struct NQBits { // 96 bits
    unsigned long long lo;
    unsigned hi;
};
__device__ __forceinline__ bool nqb_any(const NQBits& b) { return (b.lo | b.hi) != 0; }
__device__ __forceinline__ void nqb_set(NQBits& b, unsigned n)
{
    b.lo |= n < 64u ? 1ull << (n & 63u) : 0ull;
    b.hi |= n < 64u ? 0u : 1u << (n & 31u);
}
__device__ __forceinline__ void nqb_clear(NQBits& b, unsigned n)
{
    b.lo &= ~(n < 64u ? 1ull << (n & 63u) : 0ull);
    b.hi &= ~(n < 64u ? 0u : 1u << (n & 31u));
}
__device__ __forceinline__ bool nqb_test(const NQBits& b, unsigned n)
{
    return n < 64u ? ((b.lo >> (n & 63u)) & 1ull) != 0 : ((b.hi >> (n & 31u)) & 1u) != 0;
}
__device__ __forceinline__ unsigned nqb_top(const NQBits& b) // highest set bit (b != 0)
{
    return b.hi ? 95u - (unsigned)__builtin_clz(b.hi) : 63u - (unsigned)__builtin_clzll(b.lo | 1ull);
}
__device__ __forceinline__ unsigned nqb_low(const NQBits& b) // lowest set bit (b != 0)
{
    return b.lo ? (unsigned)__builtin_ctzll(b.lo) : 64u + (unsigned)__builtin_ctz(b.hi | 0x80000000u);
}
// number of set bits of b at positions j+1 .. n-1
__device__ __forceinline__ unsigned nqb_count_between(const NQBits& b, unsigned j, unsigned n)
{
    // below(x) = bits [0, x)
    const unsigned a = j + 1u;
    const unsigned long long lo_n = n >= 64u ? ~0ull : (1ull << (n & 63u)) - 1ull;
    const unsigned long long lo_a = a >= 64u ? ~0ull : (1ull << (a & 63u)) - 1ull;
    const unsigned hn = n <= 64u ? 0u : n - 64u, ha = a <= 64u ? 0u : a - 64u; // < 32 each (depth <= 95)
    const unsigned hi_n = (1u << (hn & 31u)) - 1u, hi_a = (1u << (ha & 31u)) - 1u;
    return (unsigned)__builtin_popcountll(b.lo & lo_n & ~lo_a) + (unsigned)__builtin_popcount(b.hi & hi_n & ~hi_a);
}
struct NQWalk {
    NQBits pend, s0, s1;
};
__device__ __forceinline__ unsigned nq_depth(const NQDom& d) { return (d.d & 255u) + ((d.d >> 8) & 255u) + ((d.d >> 16) & 255u); }
// the second half of the split at depth j of the path that ends in `cur` (depth n > j)
__device__ __forceinline__ NQDom nq_second_half_at(const NQWalk& w, const NQDom& cur, unsigned j)
{
    const unsigned n = nq_depth(cur);
    const unsigned c0 = nqb_count_between(w.s0, j, n), c1 = nqb_count_between(w.s1, j, n);
    const unsigned c2 = (n - 1u - j) - c0 - c1;
    NQDom r;
    r.k0 = cur.k0 >> c0;
    r.k1 = cur.k1 >> c1;
    r.k2 = cur.k2 >> c2;
    r.d = cur.d - (c0 | (c1 << 8) | (c2 << 16));
    // now the FIRST half of the split at depth j (its numerator in dimension s_j is even): step to the sibling
    const bool is0 = nqb_test(w.s0, j), is1 = nqb_test(w.s1, j);
    r.k0 |= is0 ? 1u : 0u;
    r.k1 |= (!is0 && is1) ? 1u : 0u;
    r.k2 |= (!is0 && !is1) ? 1u : 0u;
    return r;
}
// descend: the node `cur` (depth n) was split in dimension `split`; `second` = its later half is to be visited too
__device__ __forceinline__ NQDom nq_descend(NQWalk& w, const NQDom& cur, int split, bool second)
{
    const unsigned n = nq_depth(cur);
    if (split == 0) nqb_set(w.s0, n);
    if (split == 1) nqb_set(w.s1, n);
    if (second) nqb_set(w.pend, n);
    NQDom r;
    r.k0 = split == 0 ? 2u * cur.k0 : cur.k0;
    r.k1 = split == 1 ? 2u * cur.k1 : cur.k1;
    r.k2 = split == 2 ? 2u * cur.k2 : cur.k2;
    r.d = cur.d + (1u << (8 * split));
    return r;
}
// descend from a node of depth < 32: the low words only
__device__ __forceinline__ NQDom nq_descend32(NQWalk& w, const NQDom& cur, int split, bool second)
{
    const unsigned bit = 1u << (nq_depth(cur) & 31u);
    w.s0.lo |= split == 0 ? bit : 0u;
    w.s1.lo |= split == 1 ? bit : 0u;
    w.pend.lo |= second ? bit : 0u;
    NQDom r;
    r.k0 = split == 0 ? 2u * cur.k0 : cur.k0;
    r.k1 = split == 1 ? 2u * cur.k1 : cur.k1;
    r.k2 = split == 2 ? 2u * cur.k2 : cur.k2;
    r.d = cur.d + (1u << (8 * split));
    return r;
}
// the same for depths below 32 (every bit involved sits in the low words: a third of the instructions)
__device__ __forceinline__ NQDom nq_second_half_at32(const NQWalk& w, const NQDom& cur, unsigned j)
{
    const unsigned n = nq_depth(cur); // j < n <= 32
    const unsigned below_n = n >= 32u ? ~0u : (1u << (n & 31u)) - 1u;
    const unsigned m = below_n & ~((2u << j) - 1u); // bits j+1 .. n-1
    const unsigned s0 = (unsigned)w.s0.lo, s1 = (unsigned)w.s1.lo;
    const unsigned c0 = (unsigned)__builtin_popcount(s0 & m), c1 = (unsigned)__builtin_popcount(s1 & m);
    const unsigned c2 = (n - 1u - j) - c0 - c1;
    NQDom r;
    r.k0 = cur.k0 >> c0;
    r.k1 = cur.k1 >> c1;
    r.k2 = cur.k2 >> c2;
    r.d = cur.d - (c0 | (c1 << 8) | (c2 << 16));
    const bool is0 = ((s0 >> j) & 1u) != 0, is1 = ((s1 >> j) & 1u) != 0;
    r.k0 |= is0 ? 1u : 0u;
    r.k1 |= (!is0 && is1) ? 1u : 0u;
    r.k2 |= (!is0 && !is1) ? 1u : 0u;
    return r;
}
__device__ __forceinline__ NQDom nq_backtrack32(NQWalk& w, const NQDom& cur) // depth(cur) <= 32, w.pend != 0
{
    const unsigned p = (unsigned)w.pend.lo;
    const unsigned j = 31u - (unsigned)__builtin_clz(p | 1u);
    w.pend.lo = (unsigned long long)(p & ~(1u << j)) | (w.pend.lo & 0xFFFFFFFF00000000ull);
    return nq_second_half_at32(w, cur, j);
}
// backtrack to the deepest pending second half (w.pend != 0)
__device__ __forceinline__ NQDom nq_backtrack(NQWalk& w, const NQDom& cur)
{
    const unsigned j = nqb_top(w.pend);
    nqb_clear(w.pend, j);
    return nq_second_half_at(w, cur, j);
}
// give away the shallowest pending second half (w.pend != 0)
__device__ __forceinline__ NQDom nq_donate(NQWalk& w, const NQDom& cur)
{
    const unsigned j = nqb_low(w.pend);
    nqb_clear(w.pend, j);
    return nq_second_half_at(w, cur, j);
}

// origin_in_inclusion_function for np_walk_k: the same bounding box of the eight corner images as ti_inclusion, from
// FEWER operations.  Rounding is monotone -- fl(x - y) never decreases when x grows or y shrinks, and the same holds
// for one fused fl(x - d * w) -- so the extreme values over the corners are the images of the extreme operands:
//   edge-edge      c_ij = fl(x_i - y_j)                  min c = fl(min x - max y),    max c = fl(max x - min y)
//   vertex-face    c_ij = fl(fl(r1_i - p_j) - a1)        min c = fl(fl(min r1 - max p) - a1), ...   (p_j = fl(d2 w_j))
//   (fused form    r2_ij = fl(r1_i - d2 w_j): both w are tried for the extreme r1, 4 fma + 2 min/max)
// Identical cmin / cmax up to the sign of a zero (which no later comparison or subtraction can see), 8 instead of
// 12-14 operations per (coordinate, time).  v[4..7] hold displacements.  (tests/cpp/test_ti_host.cpp walks whole
// queries with it against the oracle, bit for bit with equal check counts.)
template <bool VF, int ARITH>
__device__ __forceinline__ bool ti_inclusion_mm(const double v[8][3], const double lo[3], const double hi[3],
                                                const double err[3], double ms, double& true_tol, bool& box_in)
{
    double cmin[3], cmax[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double mn = TI_DBL_MAX, mx = -TI_DBL_MAX;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const double t = it ? hi[0] : lo[0];
            double a0, a1, a2, a3; // the four vertices at time t
            if (ARITH == 1) {
                a0 = __builtin_fma(v[4][k], t, v[0][k]);
                a1 = __builtin_fma(v[5][k], t, v[1][k]);
                a2 = __builtin_fma(v[6][k], t, v[2][k]);
                a3 = __builtin_fma(v[7][k], t, v[3][k]);
            } else {
                a0 = v[4][k] * t + v[0][k];
                a1 = v[5][k] * t + v[1][k];
                a2 = v[6][k] * t + v[2][k];
                a3 = v[7][k] * t + v[3][k];
            }
            double c_lo, c_hi;
            if (VF) { // v - (t1 - t0)*u - (t2 - t0)*v - t0   with v=a0, t0=a1, t1=a2, t2=a3
                const double d1 = a2 - a1, d2 = a3 - a1;
                double r1a, r1b;
                if (ARITH == 1) {
                    r1a = __builtin_fma(-d1, lo[1], a0);
                    r1b = __builtin_fma(-d1, hi[1], a0);
                } else {
                    r1a = a0 - d1 * lo[1];
                    r1b = a0 - d1 * hi[1];
                }
                const double r1min = ti_min(r1a, r1b), r1max = ti_max(r1a, r1b);
                double r2min, r2max;
                if (ARITH == 1) {
                    r2min = ti_min(__builtin_fma(-d2, lo[2], r1min), __builtin_fma(-d2, hi[2], r1min));
                    r2max = ti_max(__builtin_fma(-d2, lo[2], r1max), __builtin_fma(-d2, hi[2], r1max));
                } else {
                    const double pa = d2 * lo[2], pb = d2 * hi[2];
                    r2min = r1min - ti_max(pa, pb);
                    r2max = r1max - ti_min(pa, pb);
                }
                c_lo = r2min - a1;
                c_hi = r2max - a1;
            } else { // ((ea1 - ea0)*u + ea0) - ((eb1 - eb0)*v + eb0)
                const double da = a1 - a0, db = a3 - a2;
                double xa, xb, ya, yb;
                if (ARITH == 1) {
                    xa = __builtin_fma(da, lo[1], a0);
                    xb = __builtin_fma(da, hi[1], a0);
                    ya = __builtin_fma(db, lo[2], a2);
                    yb = __builtin_fma(db, hi[2], a2);
                } else {
                    xa = da * lo[1] + a0;
                    xb = da * hi[1] + a0;
                    ya = db * lo[2] + a2;
                    yb = db * hi[2] + a2;
                }
                c_lo = ti_min(xa, xb) - ti_max(ya, yb);
                c_hi = ti_max(xa, xb) - ti_min(ya, yb);
            }
            mn = ti_min(mn, c_lo);
            mx = ti_max(mx, c_hi);
        }
        cmin[k] = mn;
        cmax[k] = mx;
    }
    double wdt = cmax[0] - cmin[0];
    wdt = ti_max(wdt, cmax[1] - cmin[1]);
    wdt = ti_max(wdt, cmax[2] - cmin[2]);
    true_tol = ti_max(0.0, wdt); // :183
    // (no short-circuits: every lane of the wave runs the check in lockstep, so a skipped comparison saves nothing and
    // the branches around it cost scalar instructions)
    bool out = false, notin = false;
    if (ms == 0) { // (x - 0 and x + 0 are x up to the sign of a zero, which no comparison sees: twelve additions fewer per check)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            out = out | (cmin[k] > err[k]) | (cmax[k] < -err[k]);
            notin = notin | (cmin[k] < -err[k]) | (cmax[k] > err[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            out = out | (cmin[k] - ms > err[k]) | (cmax[k] + ms < -err[k]);   // :187-190
            notin = notin | (cmin[k] + ms < -err[k]) | (cmax[k] - ms > err[k]); // :192-195
        }
    }
    box_in = !notin;
    return !out;
}

// ---- np_walk_k's query and step ---------------------------------------------------------------------------------
// Condition 1 (root_finder.cu:322) compares the three widths with the three tolerances.  A width is 2^-d, so
// 2^-d <= tol  <=>  d >= D with D = max(0, -floor(log2 tol)) (the largest power of two <= tol is 2^floor(log2 tol));
// tol = +inf (a static query) gives D = 0, a NaN tolerance D = 255 (never).  The walk kernel therefore keeps three
// 8-bit levels instead of three doubles; bit 24 marks tolerances whose reciprocal shortcut (ti_inv_tol_ok) is not
// exact -- astronomically large or small displacements -- which the kernel hands to the level-synchronous path.
struct NWQuery {
    double v[8][3]; // v[0..3] start positions, v[4..7] displacements
    double err[3];
    double inv_tol[3];
    unsigned dlev; // D_t | D_u << 8 | D_v << 16 | (inexact reciprocal) << 24
};
__device__ __forceinline__ unsigned nw_tol_level_of(double tol_k) // one dimension's level (8 bits)
{
    const int be = (int)(((unsigned long long)__double_as_longlong(tol_k) >> 52) & 0x7FFull); // (tolerances are >= 0)
    int D = 1023 - be; // -floor(log2 tol) for a normal number; subnormal or zero: 1023 (never reached)
    D = D < 0 ? 0 : (D > 255 ? 255 : D);
    if (tol_k != tol_k) D = 255;
    return (unsigned)D;
}
__device__ __forceinline__ unsigned nw_tol_levels(const double tol[3], bool inv_ok)
{
    unsigned r = inv_ok ? 0u : 1u << 24;
#pragma unroll
    for (int k = 0; k < 3; k++) r |= nw_tol_level_of(tol[k]) << (8 * k);
    return r;
}
// split_dimension :200-211 by the reciprocal shortcut (exact: see ti_inv_tol_ok; inexact ones never get here)
__device__ __forceinline__ int nw_split_of(const NWQuery& q, const double w[3])
{
    const double r0 = w[0] * q.inv_tol[0], r1 = w[1] * q.inv_tol[1], r2 = w[2] * q.inv_tol[2];
    if (r0 >= r1 && r0 >= r2) return 0;
    if (r1 >= r0 && r1 >= r2) return 1;
    return 2;
}
// the same from the packed levels of a domain entry alone (the dimension a node is split in depends on its depth only)
__device__ __forceinline__ int nw_split_dim(const NWQuery& q, unsigned d)
{
    const double w[3] = { ldexp(1.0, -(int)(d & 255u)), ldexp(1.0, -(int)((d >> 8) & 255u)), ldexp(1.0, -(int)((d >> 16) & 255u)) };
    return nw_split_of(q, w);
}
// nq_bounds with one conversion: hi = (k + 1) 2^-d = lo + w exactly (k + 1 <= 2^31, a multiple of 2^-d below 2^53 of them)
__device__ __forceinline__ void nw_bounds(unsigned k, unsigned d, double& lo, double& hi, double& w)
{
    const int e = -(int)d;
    lo = ldexp((double)k, e);
    w = ldexp(1.0, e);
    hi = lo + w;
}
// t_floor: a domain that ends at or before it is dead -- the SECOND of two launches over one list (narrow_walk.inc, "two halves of
// time") only bisects what the first, pruned at t_floor, did not reach; 0: no floor (every domain ends after 0)
template <bool VF, int ARITH>
__device__ __forceinline__ NQStep nw_step(const NWQuery& q, const NQDom& dm, double ms, double co_domain_tol,
                                          bool allow_zero_toi, double prune_toi, double t_floor = 0.0)
{
    NQStep r;
    r.accept = false;
    r.checked = false;
    r.nk = 0;
    r.split = 0;
    double lo[3], hi[3], w[3];
    const unsigned d0 = dm.d & 255u, d1 = (dm.d >> 8) & 255u, d2 = (dm.d >> 16) & 255u;
    nw_bounds(dm.k0, d0, lo[0], hi[0], w[0]);
    nw_bounds(dm.k1, d1, lo[1], hi[1], w[1]);
    nw_bounds(dm.k2, d2, lo[2], hi[2], w[2]);
    const double min_t = lo[0];
    r.min_t = min_t;
    // (straight-line on purpose: the lanes of a wave run this in lockstep; the reference's early exits are the masks below)
    const bool live = !(min_t >= prune_toi) & !(hi[0] <= t_floor); // :295
    double true_tol;
    bool box_in;
    const bool in = ti_inclusion_mm<VF, ARITH>(q.v, lo, hi, q.err, ms, true_tol, box_in);
    const bool zero_ok = allow_zero_toi | (min_t > 0);
    const bool c1 = (d0 >= (q.dlev & 255u)) & (d1 >= ((q.dlev >> 8) & 255u)) & (d2 >= ((q.dlev >> 16) & 255u)); // Condition 1 :322
    const bool acc = c1 | (box_in & zero_ok)              // Condition 2 :331
        | ((true_tol <= co_domain_tol) & zero_ok);        // Condition 3 :340
    r.checked = live;
    r.accept = live & in & acc;
    const bool splits = live & in & !acc;
    const int split = nw_split_of(q, w);
    r.split = split;
    bool second;
    if (VF) {
        const double slo = split == 0 ? lo[0] : (split == 1 ? lo[1] : lo[2]);
        const double sw = split == 0 ? w[0] : (split == 1 ? w[1] : w[2]);
        const double mid = slo + 0.5 * sw; // = (lo + hi) / 2, exact
        const double other = (split == 1) ? lo[2] : lo[1];
        second = split == 0 ? (mid <= prune_toi)                              // :229-232
                            : ((mid + other) <= 1 / (1 - TI_DBL_EPS));        // sum_less_than_one :21-29
    } else {
        // (edge-edge: the mid-point matters for a split in time only -- no selects over the three dimensions)
        const double mid_t = lo[0] + 0.5 * w[0];
        second = split == 0 ? (mid_t <= prune_toi) : true; // :229-232, :248-250
    }
    r.nk = splits ? (second ? 2 : 1) : 0;
    return r;
}
