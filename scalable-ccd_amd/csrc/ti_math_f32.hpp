// ti_math_f32.hpp -- Tight-Inclusion arithmetic of the reference's FLOAT build (SCALABLE_CCD_USE_DOUBLE = OFF,
// scalar.hpp:13-21): the float twin of ti_math.hpp, used by the level-synchronous kernels when the context runs
// with SCCD_OPT_SCALAR = 1.  Every operation is a float operation in the reference's order
// (root_finder.cu:21-254); float constants: filters :103-119, FLT_EPSILON in sum_less_than_one :21-29.
//
// Self-contained on purpose (no HIP headers): tests/cpp/test_ti_f32_host.cpp compiles this file with the host
// compiler and checks it operation for operation against the CPU oracle's float twin (orc_*_f32).
// Compiled with -ffp-contract=off: the only fused multiply-adds are the explicit ones of ARITH == 1.
#pragma once

#if !defined(__HIPCC__) && !defined(TIF_HOST_DEFS)
#define TIF_HOST_DEFS
#define __device__
#define __forceinline__ inline
#endif

#define TIF_FLT_MAX 3.402823466e+38f
#define TIF_FLT_EPS 1.192092896e-07f

// nextafterf(x, +-FLT_MAX) of the float build (scalar.hpp:31-49) by bit arithmetic, like the double twins of common.hpp
__device__ __forceinline__ float tif_from_bits(int b)
{
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}
__device__ __forceinline__ int tif_bits(float f)
{
    int b;
    __builtin_memcpy(&b, &f, 4);
    return b;
}
__device__ __forceinline__ float nextafter_up_f(float x)
{
    if (x != x) return x;
    if (x == TIF_FLT_MAX) return x;         // x == y
    if (x > TIF_FLT_MAX) return TIF_FLT_MAX; // +inf steps down towards y
    if (x == 0.0f) return tif_from_bits(1);  // smallest positive subnormal
    return tif_from_bits(tif_bits(x) + ((x > 0.0f) ? 1 : -1));
}
__device__ __forceinline__ float nextafter_down_f(float x)
{
    if (x != x) return x;
    if (x == -TIF_FLT_MAX) return x;
    if (x < -TIF_FLT_MAX) return -TIF_FLT_MAX;
    if (x == 0.0f) return tif_from_bits((int)0x80000001u);
    return tif_from_bits(tif_bits(x) + ((x > 0.0f) ? -1 : 1));
}

struct TIQueryF {
    float v[8][3]; // v0s..v3s (t = 0), v0e..v3e (t = 1): the vertices CAST TO FLOAT FIRST (ccd.cu:103-106)
    float err[3];
    float tol[3];
};

// Eigen's Array::min/max select with a compare (root_finder.cu:178-179)
__device__ __forceinline__ float tif_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float tif_max(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float tif_abs(float a) { return __builtin_fabsf(a); }

__device__ __forceinline__ float tif_linf(const float a[3], const float b[3])
{
    float m = tif_abs(b[0] - a[0]);
    m = tif_max(m, tif_abs(b[1] - a[1]));
    m = tif_max(m, tif_abs(b[2] - a[2]));
    return m;
}
// max_Linf_4 (root_finder.cu:31-46)
__device__ __forceinline__ float tif_max_linf_4(const float* p1, const float* p2, const float* p3, const float* p4,
                                                const float* p1e, const float* p2e, const float* p3e, const float* p4e)
{
    return tif_max(tif_max(tif_linf(p1, p1e), tif_linf(p2, p2e)), tif_max(tif_linf(p3, p3e), tif_linf(p4, p4e)));
}

// compute_face_vertex_tolerance / compute_edge_edge_tolerance (root_finder.cu:48-88)
template <bool VF> __device__ __forceinline__ void tif_tolerance(const float v[8][3], float co_domain_tol, float tol[3])
{
    float p000[3], p001[3], p011[3], p010[3], p100[3], p101[3], p111[3], p110[3];
    if (VF) {
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
        tol[0] = co_domain_tol / (3 * tif_max_linf_4(p000, p001, p011, p010, p100, p101, p111, p110));
        tol[1] = co_domain_tol / (3 * tif_max_linf_4(p000, p100, p101, p001, p010, p110, p111, p011));
        tol[2] = co_domain_tol / (3 * tif_max_linf_4(p000, p100, p110, p010, p001, p101, p111, p011));
    } else {
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
        // :82-87 -- tol[1] repeats the tol[0] pairing in the reference; reproduced as is
        tol[0] = co_domain_tol / (3 * tif_max_linf_4(p000, p001, p011, p010, p100, p101, p111, p110));
        tol[1] = tol[0];
        tol[2] = co_domain_tol / (3 * tif_max_linf_4(p000, p100, p101, p001, p010, p110, p111, p011));
    }
}

// get_numerical_error (root_finder.cu:90-135), float constants (:103-119)
template <bool VF> __device__ __forceinline__ void tif_error(const float v[8][3], bool use_ms, float err[3])
{
    float filter;
    if (!use_ms) filter = VF ? 3.576279e-06f : 3.337861e-06f;
    else filter = VF ? 4.053116e-06f : 3.814698e-06f;
    for (int k = 0; k < 3; k++) {
        float m = tif_abs(v[0][k]);
        for (int j = 1; j < 8; j++) m = tif_max(m, tif_abs(v[j][k]));
        m = tif_max(m, 1.0f);
        err[k] = m * m * m * filter;
    }
}

// origin_in_inclusion_function (root_finder.cu:157-198) with calculate_vf / calculate_ee (:137-155); the values that
// do not depend on u or v are computed once per t -- the same operations on the same operands
template <bool VF, int ARITH>
__device__ __forceinline__ bool tif_inclusion(const float v[8][3], const float lo[3], const float hi[3], const float err[3],
                                              float ms, float& true_tol, bool& box_in)
{
    float cmin[3], cmax[3];
    for (int k = 0; k < 3; k++) {
        float mn = TIF_FLT_MAX, mx = -TIF_FLT_MAX;
        for (int it = 0; it < 2; it++) {
            const float t = it ? hi[0] : lo[0];
            float a0, a1, a2, a3; // the four vertices at time t
            if (ARITH == 1) {
                a0 = __builtin_fmaf(v[4][k] - v[0][k], t, v[0][k]);
                a1 = __builtin_fmaf(v[5][k] - v[1][k], t, v[1][k]);
                a2 = __builtin_fmaf(v[6][k] - v[2][k], t, v[2][k]);
                a3 = __builtin_fmaf(v[7][k] - v[3][k], t, v[3][k]);
            } else {
                a0 = (v[4][k] - v[0][k]) * t + v[0][k];
                a1 = (v[5][k] - v[1][k]) * t + v[1][k];
                a2 = (v[6][k] - v[2][k]) * t + v[2][k];
                a3 = (v[7][k] - v[3][k]) * t + v[3][k];
            }
            if (VF) { // v - (t1 - t0)*u - (t2 - t0)*v - t0   with v=a0, t0=a1, t1=a2, t2=a3
                const float d1 = a2 - a1, d2 = a3 - a1;
                for (int iu = 0; iu < 2; iu++) {
                    const float u = iu ? hi[1] : lo[1];
                    const float r1 = (ARITH == 1) ? __builtin_fmaf(-d1, u, a0) : a0 - d1 * u;
                    for (int iw = 0; iw < 2; iw++) {
                        const float w = iw ? hi[2] : lo[2];
                        const float r2 = (ARITH == 1) ? __builtin_fmaf(-d2, w, r1) : r1 - d2 * w;
                        const float c = r2 - a1;
                        mn = tif_min(mn, c);
                        mx = tif_max(mx, c);
                    }
                }
            } else { // ((ea1 - ea0)*u + ea0) - ((eb1 - eb0)*v + eb0)
                const float da = a1 - a0, db = a3 - a2;
                float x[2], y[2];
                for (int i = 0; i < 2; i++) {
                    const float u = i ? hi[1] : lo[1];
                    const float w = i ? hi[2] : lo[2];
                    x[i] = (ARITH == 1) ? __builtin_fmaf(da, u, a0) : da * u + a0;
                    y[i] = (ARITH == 1) ? __builtin_fmaf(db, w, a2) : db * w + a2;
                }
                for (int iu = 0; iu < 2; iu++)
                    for (int iw = 0; iw < 2; iw++) {
                        const float c = x[iu] - y[iw];
                        mn = tif_min(mn, c);
                        mx = tif_max(mx, c);
                    }
            }
        }
        cmin[k] = mn;
        cmax[k] = mx;
    }
    float wdt = cmax[0] - cmin[0];
    wdt = tif_max(wdt, cmax[1] - cmin[1]);
    wdt = tif_max(wdt, cmax[2] - cmin[2]);
    true_tol = tif_max(0.0f, wdt); // :183
    box_in = true;
    bool out = false, notin = false;
    for (int k = 0; k < 3; k++) {
        out = out || (cmin[k] - ms > err[k]) || (cmax[k] + ms < -err[k]);     // :187-190
        notin = notin || (cmin[k] + ms < -err[k]) || (cmax[k] - ms > err[k]); // :192-195
    }
    if (out) return false;
    box_in = !notin;
    return true;
}

// One ccd_kernel invocation (root_finder.cu:277-370) without the queue mechanics; the same contract as ti_step.
struct TIStepF {
    bool accept;
    int nk;
    int split;
    float mid;
    bool checked;
};
template <bool VF, int ARITH>
__device__ __forceinline__ TIStepF tif_step(const TIQueryF& q, const float lo[3], const float hi[3], float ms,
                                            float co_domain_tol, bool allow_zero_toi, float prune_toi)
{
    TIStepF r;
    r.accept = false;
    r.nk = 0;
    r.split = 0;
    r.mid = 0;
    r.checked = false;
    const float min_t = lo[0];
    if (min_t >= prune_toi) return r; // :295
    float true_tol;
    bool box_in;
    r.checked = true;
    if (!tif_inclusion<VF, ARITH>(q.v, lo, hi, q.err, ms, true_tol, box_in)) return r;
    const float w0 = hi[0] - lo[0], w1 = hi[1] - lo[1], w2 = hi[2] - lo[2];
    const bool zero_ok = allow_zero_toi || min_t > 0;
    if ((w0 <= q.tol[0] && w1 <= q.tol[1] && w2 <= q.tol[2]) // Condition 1 :322
        || (box_in && zero_ok)                               // Condition 2 :331
        || (true_tol <= co_domain_tol && zero_ok)) {         // Condition 3 :340
        r.accept = true;
        return r;
    }
    // split_dimension :200-211
    const float r0 = w0 / q.tol[0], r1 = w1 / q.tol[1], r2 = w2 / q.tol[2];
    int split;
    if (r0 >= r1 && r0 >= r2) split = 0;
    else if (r1 >= r0 && r1 >= r2) split = 1;
    else split = 2;
    // bisect :213-254, SplitInterval interval.cuh:18-28
    const float slo = split == 0 ? lo[0] : (split == 1 ? lo[1] : lo[2]);
    const float shi = split == 0 ? hi[0] : (split == 1 ? hi[1] : hi[2]);
    const float mid = (slo + shi) / 2;
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
        // sum_less_than_one (:21-29): u + v <= 1 / (1 - FLT_EPSILON)
        const float other = (split == 1) ? lo[2] : lo[1];
        second = (mid + other) <= 1 / (1 - TIF_FLT_EPS);
    } else second = true; // :248-250
    if (second) r.nk = 2;
    return r;
}
