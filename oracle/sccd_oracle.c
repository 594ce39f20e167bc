/*
 * sccd_oracle.c -- CPU restatement of the Scalable-CCD hot path.  TEST INFRASTRUCTURE ONLY.
 * See sccd_oracle.h for the scope statement ("PARITY UNPINNED") and the import rules.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference
 * root).  Build with -ffp-contract=off: the only fused multiply-adds are the explicit fma()
 * calls of ORC_ARITH_FMA.
 */
#define _GNU_SOURCE
#include "sccd_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* src/scalable_ccd/scalar.hpp:31-49 */
static inline double nextafter_down(double x) { return nextafter(x, -DBL_MAX); }
static inline double nextafter_up(double x) { return nextafter(x, DBL_MAX); }

static inline double dmin(double a, double b) { return (b < a) ? b : a; }
static inline double dmax(double a, double b) { return (a < b) ? b : a; }

/* ------------------------------------------------------------------------------------------ */
/* Boxes                                                                                      */

/* AABB::conservative_inflation + AABB::from_point(p_t0, p_t1, r)
 * (src/scalable_ccd/broad_phase/aabb.cpp:17-36, aabb.hpp:43-51, 19-22). */
static void point_box(const double p[3], double r, double lo[3], double hi[3])
{
    for (int k = 0; k < 3; k++) {
        lo[k] = nextafter_down(p[k]) - nextafter_up(r);
        hi[k] = nextafter_up(p[k]) + nextafter_up(r);
    }
}

/* build_vertex_boxes(V0, V1, boxes, r): aabb.cpp:62-88 */
void orc_build_vertex_boxes(const double* V0, const double* V1, int nV, double r, orc_aabb* out)
{
    for (int i = 0; i < nV; i++) {
        double p0[3] = { V0[i], V0[i + (size_t)nV], V0[i + 2 * (size_t)nV] };
        double p1[3] = { V1[i], V1[i + (size_t)nV], V1[i + 2 * (size_t)nV] };
        double lo0[3], hi0[3], lo1[3], hi1[3];
        point_box(p0, r, lo0, hi0);
        point_box(p1, r, lo1, hi1);
        for (int k = 0; k < 3; k++) {
            out[i].min[k] = dmin(lo0[k], lo1[k]);
            out[i].max[k] = dmax(hi0[k], hi1[k]);
        }
        out[i].vertex_ids[0] = i;
        out[i].vertex_ids[1] = -i - 1;
        out[i].vertex_ids[2] = -i - 1;
        out[i].element_id = i;
    }
}

/* build_edge_boxes: aabb.cpp:90-112 */
void orc_build_edge_boxes(const orc_aabb* vb, const int32_t* E, int nE, orc_aabb* out)
{
    for (int i = 0; i < nE; i++) {
        const int e0 = E[i], e1 = E[i + (size_t)nE];
        for (int k = 0; k < 3; k++) {
            out[i].min[k] = dmin(vb[e0].min[k], vb[e1].min[k]);
            out[i].max[k] = dmax(vb[e0].max[k], vb[e1].max[k]);
        }
        out[i].vertex_ids[0] = e0;
        out[i].vertex_ids[1] = e1;
        out[i].vertex_ids[2] = -e0 - 1;
        out[i].element_id = i;
    }
}

/* build_face_boxes: aabb.cpp:114-133 */
void orc_build_face_boxes(const orc_aabb* vb, const int32_t* F, int nF, orc_aabb* out)
{
    for (int i = 0; i < nF; i++) {
        const int f0 = F[i], f1 = F[i + (size_t)nF], f2 = F[i + 2 * (size_t)nF];
        for (int k = 0; k < 3; k++) {
            out[i].min[k] = dmin(dmin(vb[f0].min[k], vb[f1].min[k]), vb[f2].min[k]);
            out[i].max[k] = dmax(dmax(vb[f0].max[k], vb[f1].max[k]), vb[f2].max[k]);
        }
        out[i].vertex_ids[0] = f0;
        out[i].vertex_ids[1] = f1;
        out[i].vertex_ids[2] = f2;
        out[i].element_id = i;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Broad phase                                                                                */

/* AABB::intersects: aabb.cpp:24-29 (inclusive on every axis) */
static inline int box_intersects(const orc_aabb* a, const orc_aabb* b)
{
    return a->min[0] <= b->max[0] && a->min[1] <= b->max[1] && a->min[2] <= b->max[2]
        && b->min[0] <= a->max[0] && b->min[1] <= a->max[1] && b->min[2] <= a->max[2];
}

/* share_a_vertex: sort_and_sweep.cpp:21-28 */
static inline int share_a_vertex(const int32_t* a, const int32_t* b)
{
    return a[0] == b[0] || a[0] == b[1] || a[0] == b[2] || a[1] == b[0] || a[1] == b[1]
        || a[1] == b[2] || a[2] == b[0] || a[2] == b[1] || a[2] == b[2];
}

/* is_valid_pair<is_two_lists>: sort_and_sweep.cpp:30-38 */
static inline int is_valid_pair(int two_lists, int32_t ida, int32_t idb)
{
    return !two_lists || (ida >= 0 && idb < 0) || (ida < 0 && idb >= 0);
}

static inline int32_t flip_id(int32_t id) { return -id - 1; } /* sort_and_sweep.cpp:17 */

typedef struct {
    int32_t* p;
    int64_t n, cap;
} pairvec;

static void pv_push(pairvec* v, int32_t a, int32_t b)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 1024;
        v->p = (int32_t*)realloc(v->p, sizeof(int32_t) * 2 * (size_t)v->cap);
    }
    v->p[2 * v->n] = a;
    v->p[2 * v->n + 1] = b;
    v->n++;
}

static int g_sort_axis;
static int cmp_box_axis(const void* pa, const void* pb)
{
    /* SortBoxes: sort_and_sweep.cpp:60-72 */
    const double a = ((const orc_aabb*)pa)->min[g_sort_axis];
    const double b = ((const orc_aabb*)pb)->min[g_sort_axis];
    return (a < b) ? -1 : (b < a) ? 1 : 0;
}

static int64_t g_last_tests = 0;
int64_t orc_last_candidate_tests(void) { return g_last_tests; }

/* batched_sweep + sweep: sort_and_sweep.cpp:77-196.  boxes sorted by min[axis].
 * (tbb::parallel_for over i with thread-local vectors + serial merge == the OpenMP loop below;
 * the output order is unspecified in the reference as well.) */
static int64_t sweep(const orc_aabb* boxes, int n, int two_lists, int* sort_axis, int32_t** out,
                     int nthreads)
{
    const int axis = *sort_axis;
    int nt = nthreads > 0 ? nthreads : 1;
    pairvec* locals = (pairvec*)calloc((size_t)nt, sizeof(pairvec));
    int64_t tests = 0;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nt) reduction(+ : tests)
    for (int i = 0; i < n; i++) {
#ifdef _OPENMP
        pairvec* lv = &locals[omp_get_thread_num()];
#else
        pairvec* lv = &locals[0];
#endif
        const orc_aabb* a = &boxes[i];
        for (int j = i + 1; j < n; j++) {
            const orc_aabb* b = &boxes[j];
            if (a->max[axis] < b->min[axis]) {
                break; /* :98 */
            }
            tests++;
            if (is_valid_pair(two_lists, a->element_id, b->element_id) && box_intersects(a, b)
                && !share_a_vertex(a->vertex_ids, b->vertex_ids)) {
                if (two_lists) { /* :106-112: negative ids are list A */
                    pv_push(lv, a->element_id < 0 ? flip_id(a->element_id) : flip_id(b->element_id),
                            a->element_id < 0 ? b->element_id : a->element_id);
                } else { /* :113-118 */
                    pv_push(lv, a->element_id < b->element_id ? a->element_id : b->element_id,
                            a->element_id < b->element_id ? b->element_id : a->element_id);
                }
            }
        }
    }
    g_last_tests = tests;

    /* merge_local_overlaps: src/scalable_ccd/utils/merge_local_overlaps.cpp:5-19 */
    int64_t total = 0;
    for (int t = 0; t < nt; t++) total += locals[t].n;
    int32_t* pairs = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(total ? total : 1));
    int64_t at = 0;
    for (int t = 0; t < nt; t++) {
        if (locals[t].n) memcpy(pairs + 2 * at, locals[t].p, sizeof(int32_t) * 2 * (size_t)locals[t].n);
        at += locals[t].n;
        free(locals[t].p);
    }
    free(locals);
    *out = pairs;

    /* next sort axis = arg-max variance of box centres: sort_and_sweep.cpp:176-195 */
    double s[3] = { 0, 0, 0 }, s2[3] = { 0, 0, 0 };
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) {
            const double c = (boxes[i].min[k] + boxes[i].max[k]) / 2;
            s[k] += c;
            s2[k] += c * c;
        }
    }
    double var[3];
    for (int k = 0; k < 3; k++) var[k] = s2[k] - s[k] * s[k] / n;
    int ax = 0;
    if (var[1] > var[0]) ax = 1;
    if (var[2] > var[ax]) ax = 2;
    *sort_axis = ax;
    return total;
}

/* sort_and_sweep (one list): sort_and_sweep.cpp:198-211 */
int64_t orc_sort_and_sweep(const orc_aabb* boxes_in, int n, int* sort_axis, int32_t** pairs,
                           int nthreads)
{
    *pairs = NULL;
    g_last_tests = 0;
    if (n == 0) return 0;
    orc_aabb* boxes = (orc_aabb*)malloc(sizeof(orc_aabb) * (size_t)n);
    memcpy(boxes, boxes_in, sizeof(orc_aabb) * (size_t)n);
    g_sort_axis = *sort_axis;
    qsort(boxes, (size_t)n, sizeof(orc_aabb), cmp_box_axis); /* sort_along_axis :128-142 */
    const int64_t r = sweep(boxes, n, 0, sort_axis, pairs, nthreads);
    free(boxes);
    return r;
}

/* sort_and_sweep (two lists): sort_and_sweep.cpp:213-240 */
int64_t orc_sort_and_sweep_two_lists(const orc_aabb* A, int nA, const orc_aabb* B, int nB,
                                     int* sort_axis, int32_t** pairs, int nthreads)
{
    *pairs = NULL;
    g_last_tests = 0;
    if (nA == 0 || nB == 0) return 0;
    const int n = nA + nB;
    orc_aabb* boxes = (orc_aabb*)malloc(sizeof(orc_aabb) * (size_t)n);
    memcpy(boxes, A, sizeof(orc_aabb) * (size_t)nA);
    memcpy(boxes + nA, B, sizeof(orc_aabb) * (size_t)nB);
    for (int i = 0; i < nA; i++) boxes[i].element_id = flip_id(boxes[i].element_id); /* :228-231 */
    /* sort each + std::merge == one sort of the union as far as the pair SET is concerned
     * (tie order only changes the emission order). */
    g_sort_axis = *sort_axis;
    qsort(boxes, (size_t)n, sizeof(orc_aabb), cmp_box_axis);
    const int64_t r = sweep(boxes, n, 1, sort_axis, pairs, nthreads);
    free(boxes);
    return r;
}

/* Independent check: all pairs, no sort, no break. */
int64_t orc_brute_force(const orc_aabb* A, int nA, const orc_aabb* B, int nB, int32_t** pairs)
{
    pairvec v = { 0, 0, 0 };
    if (B == NULL) {
        for (int i = 0; i < nA; i++)
            for (int j = i + 1; j < nA; j++)
                if (box_intersects(&A[i], &A[j]) && !share_a_vertex(A[i].vertex_ids, A[j].vertex_ids)) {
                    const int32_t a = A[i].element_id, b = A[j].element_id;
                    pv_push(&v, a < b ? a : b, a < b ? b : a);
                }
    } else {
        for (int i = 0; i < nA; i++)
            for (int j = 0; j < nB; j++)
                if (box_intersects(&A[i], &B[j]) && !share_a_vertex(A[i].vertex_ids, B[j].vertex_ids))
                    pv_push(&v, A[i].element_id, B[j].element_id);
    }
    if (!v.p) v.p = (int32_t*)malloc(8);
    *pairs = v.p;
    return v.n;
}

void orc_free(void* p) { free(p); }

static int cmp_pair(const void* pa, const void* pb)
{
    const int32_t* a = (const int32_t*)pa;
    const int32_t* b = (const int32_t*)pb;
    if (a[0] != b[0]) return a[0] < b[0] ? -1 : 1;
    if (a[1] != b[1]) return a[1] < b[1] ? -1 : 1;
    return 0;
}
void orc_sort_pairs(int32_t* pairs, int64_t n) { qsort(pairs, (size_t)n, 8, cmp_pair); }

/* ------------------------------------------------------------------------------------------ */
/* Narrow phase                                                                               */

/* CCDData: src/scalable_ccd/cuda/narrow_phase/ccd_data.cuh:8-26 */
typedef struct {
    double v[8][3]; /* v0s v1s v2s v3s v0e v1e v2e v3e */
    double err[3];
    double tol[3];
    double ms;
    double toi; /* TOI_PER_QUERY */
    int nbr_checks;
} ccd_data;

/* CCDDomain: interval.cuh:30-44 */
typedef struct {
    double lo[3], hi[3]; /* t,u,v */
    int query_id;
} ccd_domain;

/* add_data<is_vf>: narrow_phase.cu:24-74 */
static void gather_query(const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                         const int32_t* F, int nF, int a, int b, int is_vf, double v[8][3])
{
    int id[4];
    if (is_vf) { /* :41-53 */
        id[0] = a;
        id[1] = F[b];
        id[2] = F[b + (size_t)nF];
        id[3] = F[b + 2 * (size_t)nF];
    } else { /* :54-66 */
        id[0] = E[a];
        id[1] = E[a + (size_t)nE];
        id[2] = E[b];
        id[3] = E[b + (size_t)nE];
    }
    for (int k = 0; k < 4; k++)
        for (int c = 0; c < 3; c++) {
            v[k][c] = V0[id[k] + (size_t)c * nV];
            v[k + 4][c] = V1[id[k] + (size_t)c * nV];
        }
}

static inline double linf_diff(const double a[3], const double b[3])
{
    /* (b - a).lpNorm<Infinity>() */
    double m = fabs(b[0] - a[0]);
    m = dmax(m, fabs(b[1] - a[1]));
    m = dmax(m, fabs(b[2] - a[2]));
    return m;
}

/* max_Linf_4: root_finder.cu:31-46 */
static double max_linf_4(const double* p1, const double* p2, const double* p3, const double* p4,
                         const double* p1e, const double* p2e, const double* p3e, const double* p4e)
{
    return dmax(dmax(linf_diff(p1, p1e), linf_diff(p2, p2e)), dmax(linf_diff(p3, p3e), linf_diff(p4, p4e)));
}

/* compute_face_vertex_tolerance / compute_edge_edge_tolerance (root_finder.cu:48-88) and
 * get_numerical_error (root_finder.cu:90-135). v = v0s..v3s,v0e..v3e */
void orc_query_constants(const double* vv, int is_vf, int use_ms, double co_domain_tol, double* tol,
                         double* err)
{
    const double(*v)[3] = (const double(*)[3])vv;
    const double *v0s = v[0], *v1s = v[1], *v2s = v[2], *v3s = v[3];
    const double *v0e = v[4], *v1e = v[5], *v2e = v[6], *v3e = v[7];
    double p000[3], p001[3], p011[3], p010[3], p100[3], p101[3], p111[3], p110[3];
    if (is_vf) { /* :50-59 */
        for (int k = 0; k < 3; k++) {
            p000[k] = v0s[k] - v1s[k];
            p001[k] = v0s[k] - v3s[k];
            p011[k] = v0s[k] - (v2s[k] + v3s[k] - v1s[k]);
            p010[k] = v0s[k] - v2s[k];
            p100[k] = v0e[k] - v1e[k];
            p101[k] = v0e[k] - v3e[k];
            p111[k] = v0e[k] - (v2e[k] + v3e[k] - v1e[k]);
            p110[k] = v0e[k] - v2e[k];
        }
        tol[0] = co_domain_tol / (3 * max_linf_4(p000, p001, p011, p010, p100, p101, p111, p110));
        tol[1] = co_domain_tol / (3 * max_linf_4(p000, p100, p101, p001, p010, p110, p111, p011));
        tol[2] = co_domain_tol / (3 * max_linf_4(p000, p100, p110, p010, p001, p101, p111, p011));
    } else { /* :73-87 -- tol[1] deliberately repeats the tol[0] pairing, as the reference does */
        for (int k = 0; k < 3; k++) {
            p000[k] = v0s[k] - v2s[k];
            p001[k] = v0s[k] - v3s[k];
            p010[k] = v1s[k] - v2s[k];
            p011[k] = v1s[k] - v3s[k];
            p100[k] = v0e[k] - v2e[k];
            p101[k] = v0e[k] - v3e[k];
            p110[k] = v1e[k] - v2e[k];
            p111[k] = v1e[k] - v3e[k];
        }
        tol[0] = co_domain_tol / (3 * max_linf_4(p000, p001, p011, p010, p100, p101, p111, p110));
        tol[1] = co_domain_tol / (3 * max_linf_4(p000, p001, p011, p010, p100, p101, p111, p110));
        tol[2] = co_domain_tol / (3 * max_linf_4(p000, p100, p101, p001, p010, p110, p111, p011));
    }
    /* get_numerical_error: :93-134 (double constants) */
    double filter;
    if (!use_ms) {
        filter = is_vf ? 6.661338147750939e-15 : 6.217248937900877e-15;
    } else {
        filter = is_vf ? 7.549516567451064e-15 : 7.105427357601002e-15;
    }
    for (int k = 0; k < 3; k++) {
        double m = fabs(v0s[k]);
        m = dmax(m, fabs(v1s[k]));
        m = dmax(m, fabs(v2s[k]));
        m = dmax(m, fabs(v3s[k]));
        m = dmax(m, fabs(v0e[k]));
        m = dmax(m, fabs(v1e[k]));
        m = dmax(m, fabs(v2e[k]));
        m = dmax(m, fabs(v3e[k]));
        m = dmax(m, 1.0);
        err[k] = m * m * m * filter;
    }
}

/* calculate_vf / calculate_ee: root_finder.cu:137-155.  One coordinate. */
static inline double lerp_strict(double s, double e, double t) { return (e - s) * t + s; }
static inline double lerp_fma(double s, double e, double t) { return fma(e - s, t, s); }

static inline double eval_vf(const double (*v)[3], int k, double t, double u, double w, int arith)
{
    if (arith == ORC_ARITH_FMA) {
        const double p = lerp_fma(v[0][k], v[4][k], t);
        const double t0 = lerp_fma(v[1][k], v[5][k], t);
        const double t1 = lerp_fma(v[2][k], v[6][k], t);
        const double t2 = lerp_fma(v[3][k], v[7][k], t);
        /* v - (t1-t0)*u - (t2-t0)*v - t0 with both products fused into the subtraction */
        double r = fma(-(t1 - t0), u, p);
        r = fma(-(t2 - t0), w, r);
        return r - t0;
    } else {
        const double p = lerp_strict(v[0][k], v[4][k], t);
        const double t0 = lerp_strict(v[1][k], v[5][k], t);
        const double t1 = lerp_strict(v[2][k], v[6][k], t);
        const double t2 = lerp_strict(v[3][k], v[7][k], t);
        return p - (t1 - t0) * u - (t2 - t0) * w - t0;
    }
}

static inline double eval_ee(const double (*v)[3], int k, double t, double u, double w, int arith)
{
    if (arith == ORC_ARITH_FMA) {
        const double ea0 = lerp_fma(v[0][k], v[4][k], t);
        const double ea1 = lerp_fma(v[1][k], v[5][k], t);
        const double eb0 = lerp_fma(v[2][k], v[6][k], t);
        const double eb1 = lerp_fma(v[3][k], v[7][k], t);
        return fma(ea1 - ea0, u, ea0) - fma(eb1 - eb0, w, eb0);
    } else {
        const double ea0 = lerp_strict(v[0][k], v[4][k], t);
        const double ea1 = lerp_strict(v[1][k], v[5][k], t);
        const double eb0 = lerp_strict(v[2][k], v[6][k], t);
        const double eb1 = lerp_strict(v[3][k], v[7][k], t);
        return ((ea1 - ea0) * u + ea0) - ((eb1 - eb0) * w + eb0);
    }
}

/* origin_in_inclusion_function: root_finder.cu:157-198 */
static int origin_in_inclusion(const double (*v)[3], const double* lo, const double* hi,
                               const double* err, double ms, int is_vf, int arith, double* true_tol,
                               int* box_in)
{
    double cmin[3] = { DBL_MAX, DBL_MAX, DBL_MAX }, cmax[3] = { -DBL_MAX, -DBL_MAX, -DBL_MAX };
    for (int corner = 0; corner < 8; corner++) { /* DomainCorner::update_tuv interval.cuh:51-56 */
        const double t = (corner & 1) ? hi[0] : lo[0];
        const double u = (corner & 2) ? hi[1] : lo[1];
        const double w = (corner & 4) ? hi[2] : lo[2];
        for (int k = 0; k < 3; k++) {
            const double c = is_vf ? eval_vf(v, k, t, u, w, arith) : eval_ee(v, k, t, u, w, arith);
            cmin[k] = dmin(cmin[k], c);
            cmax[k] = dmax(cmax[k], c);
        }
    }
    double w = cmax[0] - cmin[0];
    w = dmax(w, cmax[1] - cmin[1]);
    w = dmax(w, cmax[2] - cmin[2]);
    *true_tol = dmax(0.0, w); /* :183 */
    *box_in = 1;
    for (int k = 0; k < 3; k++) /* :187-190 */
        if (cmin[k] - ms > err[k] || cmax[k] + ms < -err[k]) return 0;
    for (int k = 0; k < 3; k++) /* :192-195 */
        if (cmin[k] + ms < -err[k] || cmax[k] - ms > err[k]) *box_in = 0;
    return 1;
}

int orc_origin_in_inclusion_function(const double* v, const double* dom, const double* err, double ms,
                                     int is_vf, int arith, double* true_tol, int* box_in)
{
    const double lo[3] = { dom[0], dom[2], dom[4] }, hi[3] = { dom[1], dom[3], dom[5] };
    return origin_in_inclusion((const double(*)[3])v, lo, hi, err, ms, is_vf, arith, true_tol, box_in);
}

/* split_dimension: root_finder.cu:200-211 */
static int split_dimension(const double* tol, const double* w)
{
    const double r0 = w[0] / tol[0], r1 = w[1] / tol[1], r2 = w[2] / tol[2];
    if (r0 >= r1 && r0 >= r2) return 0;
    if (r1 >= r0 && r1 >= r2) return 1;
    return 2;
}

/* sum_less_than_one: root_finder.cu:21-29 */
static inline int sum_less_than_one(double a, double b) { return a + b <= 1 / (1 - DBL_EPSILON); }

/* One ccd_kernel invocation (root_finder.cu:277-370) on `dom`.
 * prune_toi = the value the reference compares against (*toi, or data.toi in TOI_PER_QUERY).
 * Returns: bit0 = accepted (min_t is a TOI candidate); children written to kids[], *nk = 0..2.
 * *checked = 1 if the inclusion function was evaluated. */
static int ccd_step(const ccd_data* d, const ccd_domain* dom, int is_vf, int arith, double tol,
                    int allow_zero_toi, int max_iter, int checks_before, double prune_toi,
                    ccd_domain kids[2], int* nk, int* checked)
{
    *nk = 0;
    *checked = 0;
    const double min_t = dom->lo[0];
    if (min_t >= prune_toi) return 0;                      /* :295-300 */
    if (max_iter >= 0 && checks_before > max_iter) return 0; /* :303 */
    double true_tol = 0;
    int box_in;
    *checked = 1;
    if (!origin_in_inclusion(d->v, dom->lo, dom->hi, d->err, d->ms, is_vf, arith, &true_tol, &box_in))
        return 0;
    const double w[3] = { dom->hi[0] - dom->lo[0], dom->hi[1] - dom->lo[1], dom->hi[2] - dom->lo[2] };
    if (w[0] <= d->tol[0] && w[1] <= d->tol[1] && w[2] <= d->tol[2]) return 1;  /* C1 :322 */
    if (box_in && (allow_zero_toi || min_t > 0)) return 1;                      /* C2 :331 */
    if (true_tol <= tol && (allow_zero_toi || min_t > 0)) return 1;             /* C3 :340 */
    const int split = split_dimension(d->tol, w);                               /* :350 */
    /* bisect: :213-254, SplitInterval interval.cuh:18-28 */
    const double mid = (dom->lo[split] + dom->hi[split]) / 2;
    if (dom->lo[split] >= mid || mid >= dom->hi[split]) return 1; /* C4 :222-225,:362 */
    kids[0] = *dom;
    kids[0].hi[split] = mid;
    *nk = 1;
    int second = 0;
    if (split == 0) {
        second = (mid <= prune_toi); /* :229-232 */
    } else if (is_vf) {
        if (split == 1) second = sum_less_than_one(mid, dom->lo[2]); /* :235-240 */
        else second = sum_less_than_one(mid, dom->lo[1]);            /* :241-246 */
    } else {
        second = 1; /* :248-250 */
    }
    if (second) {
        kids[1] = *dom;
        kids[1].lo[split] = mid;
        *nk = 2;
    }
    return 0;
}

static int64_t g_level_budget = ORC_MAX_LEVEL_DOMAINS;
void orc_set_level_budget(int64_t domains) { g_level_budget = domains > 0 ? domains : ORC_MAX_LEVEL_DOMAINS; }

int orc_narrow_phase(const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                     const int32_t* F, int nF, const int32_t* pairs, int64_t n, int is_vf, double ms,
                     int max_iter, double tol, int allow_zero_toi, int arith, double* toi_io,
                     double* per_query_toi, orc_np_stats* stats)
{
    orc_np_stats st;
    memset(&st, 0, sizeof st);
    st.n_queries = n;
    double toi = *toi_io;
    const int use_ms = ms > 0; /* narrow_phase.cu:128 */
    const int per_query = per_query_toi != NULL;
    /* narrow_phase.cu:136: loop guard toi > 0 (not in TOI_PER_QUERY builds) */
    if (n > 0 && (toi > 0 || per_query)) {
        ccd_data* data = (ccd_data*)malloc(sizeof(ccd_data) * (size_t)n);
        /* two growable level buffers instead of the reference's fixed ring (ccd_buffer.cuh:7-83) */
        int64_t cap_cur = n + 16, cap_nxt = 2 * n + 16, n_cur = 0, n_nxt = 0;
        ccd_domain* cur = (ccd_domain*)malloc(sizeof(ccd_domain) * (size_t)cap_cur);
        ccd_domain* nxt = (ccd_domain*)malloc(sizeof(ccd_domain) * (size_t)cap_nxt);
        for (int64_t i = 0; i < n; i++) {
            gather_query(V0, V1, nV, E, nE, F, nF, pairs[2 * i], pairs[2 * i + 1], is_vf, data[i].v);
            data[i].ms = ms;
            data[i].toi = INFINITY; /* narrow_phase.cu:70 */
            data[i].nbr_checks = 0;
            orc_query_constants(&data[i].v[0][0], is_vf, use_ms, tol, data[i].tol, data[i].err); /* compute_tolerance :260-275 */
            ccd_domain d0 = { { 0, 0, 0 }, { 1, 1, 1 }, (int)i }; /* initialize_buffer ccd_buffer.cuh:70-77 */
            cur[n_cur++] = d0;
        }
        /* level loop root_finder.cu:431-447 */
        int level = 0;
        while (n_cur > 0) {
            if (n_cur > st.max_queue) st.max_queue = n_cur;
            n_nxt = 0;
            for (int64_t h = 0; h < n_cur; h++) {
                const ccd_domain dom = cur[h];
                ccd_data* d = &data[dom.query_id];
                const int before = d->nbr_checks; /* data_in copy :288 */
                d->nbr_checks++;                  /* atomicAdd :289 */
                st.n_domains++;
                ccd_domain kids[2];
                int nk, checked;
                const double prune = per_query ? d->toi : toi;
                const int acc = ccd_step(d, &dom, is_vf, arith, tol, allow_zero_toi, max_iter, before,
                                         prune, kids, &nk, &checked);
                st.n_checks += checked;
                if (acc) {
                    if (dom.lo[0] < toi) toi = dom.lo[0]; /* atomicMin(toi, min_t) */
                    if (dom.lo[0] < d->toi) d->toi = dom.lo[0];
                }
                if (nk && level == 0) st.n_root_survive++;
                if (n_nxt + 2 > cap_nxt) {
                    /* Level order keeps every live domain of a level: a contact-rich query set grows like
                     * (1/tolerance)^2 (per-query mode has no global bound to prune by).  Give up at a fixed
                     * budget instead of taking the host's memory (that took two test machines down). */
                    if (cap_nxt >= g_level_budget) {
                        free(cur);
                        free(nxt);
                        free(data);
                        *toi_io = NAN;
                        if (stats) *stats = st;
                        return ORC_E_BUDGET;
                    }
                    cap_nxt *= 2;
                    nxt = (ccd_domain*)realloc(nxt, sizeof(ccd_domain) * (size_t)cap_nxt);
                }
                for (int k = 0; k < nk; k++) nxt[n_nxt++] = kids[k];
            }
            /* shift_queue_start ccd_buffer.cuh:41-52: the children become the next level */
            ccd_domain* tp = cur; cur = nxt; nxt = tp;
            int64_t tc = cap_cur; cap_cur = cap_nxt; cap_nxt = tc;
            n_cur = n_nxt;
            level++;
        }
        for (int64_t i = 0; i < n; i++) {
            if (data[i].nbr_checks > st.max_checks_per_query) st.max_checks_per_query = data[i].nbr_checks;
            if (per_query_toi) per_query_toi[i] = data[i].toi;
        }
        free(cur);
        free(nxt);
        free(data);
    }
    *toi_io = toi;
    if (stats) *stats = st;
    return 0;
}

/* shared non-negative double min via the IEEE bit pattern (atomic_min_float.cuh:17-29) */
static inline void atomic_min_double(double* addr, double val)
{
    int64_t v, old;
    memcpy(&v, &val, 8);
    old = __atomic_load_n((int64_t*)addr, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n((int64_t*)addr, &old, v, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
}

int orc_narrow_phase_mt(const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                        const int32_t* F, int nF, const int32_t* pairs, int64_t n, int is_vf, double ms,
                        int max_iter, double tol, int allow_zero_toi, int arith, double* toi_io,
                        int32_t* checks_per_query, int nthreads)
{
    double toi = *toi_io;
    const int use_ms = ms > 0;
    if (n <= 0 || !(toi > 0)) return 0;
    int nt = nthreads > 0 ? nthreads : 1;
#pragma omp parallel num_threads(nt)
    {
        int cap = 256;
        ccd_domain* stack = (ccd_domain*)malloc(sizeof(ccd_domain) * (size_t)cap);
#pragma omp for schedule(dynamic, 512)
        for (int64_t i = 0; i < n; i++) {
            ccd_data d;
            gather_query(V0, V1, nV, E, nE, F, nF, pairs[2 * i], pairs[2 * i + 1], is_vf, d.v);
            d.ms = ms;
            orc_query_constants(&d.v[0][0], is_vf, use_ms, tol, d.tol, d.err);
            int sp = 0, checks = 0;
            ccd_domain d0 = { { 0, 0, 0 }, { 1, 1, 1 }, (int)i };
            stack[sp++] = d0;
            while (sp > 0) {
                const ccd_domain dom = stack[--sp];
                ccd_domain kids[2];
                int nk, checked;
                double cur;
                {
                    const int64_t bits = __atomic_load_n((int64_t*)&toi, __ATOMIC_RELAXED);
                    memcpy(&cur, &bits, 8);
                }
                const int acc = ccd_step(&d, &dom, is_vf, arith, tol, allow_zero_toi, max_iter, checks, cur,
                                         kids, &nk, &checked);
                checks++;
                if (acc) atomic_min_double(&toi, dom.lo[0]);
                if (sp + 2 > cap) {
                    cap *= 2;
                    stack = (ccd_domain*)realloc(stack, sizeof(ccd_domain) * (size_t)cap);
                }
                /* push the later half first so the earlier half is explored first */
                if (nk == 2) stack[sp++] = kids[1];
                if (nk >= 1) stack[sp++] = kids[0];
            }
            if (checks_per_query) checks_per_query[i] = checks;
        }
        free(stack);
    }
    *toi_io = toi;
    return 0;
}

/* ccd(): src/scalable_ccd/cuda/ccd.cu:80-146 */
int orc_ccd(const double* V0, const double* V1, int nV, const int32_t* E, int nE, const int32_t* F,
            int nF, double ms, int max_iter, double tol, int allow_zero_toi, int arith, int nthreads,
            double* toi_out, int64_t* n_vf, int64_t* n_ee)
{
    orc_aabb* vb = (orc_aabb*)malloc(sizeof(orc_aabb) * (size_t)(nV ? nV : 1));
    orc_aabb* eb = (orc_aabb*)malloc(sizeof(orc_aabb) * (size_t)(nE ? nE : 1));
    orc_aabb* fb = (orc_aabb*)malloc(sizeof(orc_aabb) * (size_t)(nF ? nF : 1));
    orc_build_vertex_boxes(V0, V1, nV, ms, vb); /* ccd.cu:112: inflation radius = min_distance */
    orc_build_edge_boxes(vb, E, nE, eb);
    orc_build_face_boxes(vb, F, nF, fb);
    double toi = 1; /* ccd.cu:125 */
    int axis = 0;   /* device path always sorts on x: aabb.cu:85-86 */
    int32_t* pairs = NULL;
    int64_t nvf = orc_sort_and_sweep_two_lists(vb, nV, fb, nF, &axis, &pairs, nthreads);
    int rc = 0;
    if (nthreads > 1)
        orc_narrow_phase_mt(V0, V1, nV, E, nE, F, nF, pairs, nvf, 1, ms, max_iter, tol, allow_zero_toi, arith, &toi, NULL, nthreads);
    else
        rc = orc_narrow_phase(V0, V1, nV, E, nE, F, nF, pairs, nvf, 1, ms, max_iter, tol, allow_zero_toi, arith, &toi, NULL, NULL);
    free(pairs);
    pairs = NULL;
    axis = 0;
    int64_t nee = orc_sort_and_sweep(eb, nE, &axis, &pairs, nthreads);
    if (rc == 0) { /* (a vertex-face pass that ran out of its level budget leaves nothing to seed the edge-edge pass with) */
        if (nthreads > 1)
            orc_narrow_phase_mt(V0, V1, nV, E, nE, F, nF, pairs, nee, 0, ms, max_iter, tol, allow_zero_toi, arith, &toi, NULL, nthreads);
        else
            rc = orc_narrow_phase(V0, V1, nV, E, nE, F, nF, pairs, nee, 0, ms, max_iter, tol, allow_zero_toi, arith, &toi, NULL, NULL);
    }
    free(pairs);
    free(vb);
    free(eb);
    free(fb);
    *toi_out = toi;
    if (n_vf) *n_vf = nvf;
    if (n_ee) *n_ee = nee;
    return rc;
}
