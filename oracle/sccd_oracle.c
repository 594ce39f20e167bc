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
/* Narrow phase and ccd(): np_core.inc, once per scalar type                                   */
static int64_t g_level_budget = ORC_MAX_LEVEL_DOMAINS;
void orc_set_level_budget(int64_t domains) { g_level_budget = domains > 0 ? domains : ORC_MAX_LEVEL_DOMAINS; }

#define REAL double
#define NM(x) x
#define R_MAX DBL_MAX
#define R_EPS DBL_EPSILON
#define R_ABS(x) fabs(x)
#define R_FMA(a, b, c) fma(a, b, c)
#define R_BITS int64_t
#define R_FILTERS { 6.661338147750939e-15, 6.217248937900877e-15, 7.549516567451064e-15, 7.105427357601002e-15 }
#include "np_core.inc"
#undef REAL
#undef NM
#undef R_MAX
#undef R_EPS
#undef R_ABS
#undef R_FMA
#undef R_BITS
#undef R_FILTERS

/* SCALABLE_CCD_USE_DOUBLE off: Scalar = float.  Vertices are cast to float FIRST (aabb.cpp:43-47, ccd.cu:103-106),
 * then everything -- boxes, tolerances, error bounds, the inclusion function, mid-points, the TOI -- is float
 * arithmetic; the float filter constants are root_finder.cu:103-119, the bound of sum_less_than_one uses
 * FLT_EPSILON (:21-29).  The *_f32 entry points take float arrays. */
static inline float nextafter_down_f32(float x) { return nextafterf(x, -FLT_MAX); }
static inline float nextafter_up_f32(float x) { return nextafterf(x, FLT_MAX); }
void orc_build_vertex_boxes_f32(const float* V0, const float* V1, int nV, float r, orc_aabb* out)
{
    const float ru = nextafter_up_f32(r);
    for (int i = 0; i < nV; i++) {
        for (int k = 0; k < 3; k++) {
            const float p0 = V0[i + (size_t)k * nV], p1 = V1[i + (size_t)k * nV];
            const float l0 = nextafter_down_f32(p0) - ru, l1 = nextafter_down_f32(p1) - ru;
            const float h0 = nextafter_up_f32(p0) + ru, h1 = nextafter_up_f32(p1) + ru;
            out[i].min[k] = (l1 < l0) ? l1 : l0; /* float values in the double fields: every compare downstream is exact */
            out[i].max[k] = (h0 < h1) ? h1 : h0;
        }
        out[i].vertex_ids[0] = i;
        out[i].vertex_ids[1] = -i - 1;
        out[i].vertex_ids[2] = -i - 1;
        out[i].element_id = i;
    }
}
#define REAL float
#define NM(x) x##_f32
#define R_MAX FLT_MAX
#define R_EPS FLT_EPSILON
#define R_ABS(x) fabsf(x)
#define R_FMA(a, b, c) fmaf(a, b, c)
#define R_BITS int32_t
#define R_FILTERS { 3.576279e-06, 3.337861e-06, 4.053116e-06, 3.814698e-06 }
#include "np_core.inc"
