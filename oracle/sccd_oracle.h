/*
 * sccd_oracle.h -- CPU restatement of the Scalable-CCD hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the checker, never the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  The product path (libsccd_hip.so) never calls it.
 *
 * PARITY UNPINNED: the reference cannot be built in this image (its CPU path needs Eigen and
 * oneTBB headers, its narrow phase is CUDA-only; neither is installed and stand-in headers are
 * not allowed), and the reference's only golden numbers need the absent sample-data repository
 * (tests/test_broad_phase.cpp:36-38,62-63, tests/test_narrow_phase.cu:65).  The restatement is
 * therefore checked against (a) an independent O(n^2) brute-force pair finder, (b) analytic
 * known-answer TOI cases and (c) line-by-line citation of the reference below.
 *
 * Scalar = double (reference default, CMakeLists.txt:69 SCALABLE_CCD_USE_DOUBLE=ON) for every orc_* function;
 * the orc_*_f32 twins at the end restate the SCALABLE_CCD_USE_DOUBLE=OFF build (Scalar = float).  The narrow
 * phase and ccd() are written once in np_core.inc and compiled for both.
 */
#ifndef SCCD_ORACLE_H
#define SCCD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same 64-byte layout as the reference's scalable_ccd::cuda::AABB
 * (src/scalable_ccd/cuda/broad_phase/aabb.cuh:82-92). */
typedef struct {
    double min[3];
    double max[3];
    int32_t vertex_ids[3];
    int32_t element_id;
} orc_aabb;

/* arithmetic mode for the inclusion function (see DESIGN.md "Arithmetic contract") */
#define ORC_ARITH_STRICT 0 /* every * and +/- rounded separately, in source order     */
#define ORC_ARITH_FMA 1    /* a*b+c patterns of root_finder.cu:140-154 fused (nvcc -fmad) */

/* ---- boxes: src/scalable_ccd/broad_phase/aabb.cpp:17-133 ------------------------------- */
/* V0,V1: column-major nV x 3 (Eigen::MatrixXd storage). */
void orc_build_vertex_boxes(const double* V0, const double* V1, int nV,
                            double inflation_radius, orc_aabb* out);
/* E: column-major nE x 2, F: column-major nF x 3 (Eigen::MatrixXi storage). */
void orc_build_edge_boxes(const orc_aabb* vertex_boxes, const int32_t* E, int nE, orc_aabb* out);
void orc_build_face_boxes(const orc_aabb* vertex_boxes, const int32_t* F, int nF, orc_aabb* out);

/* ---- broad phase: src/scalable_ccd/broad_phase/sort_and_sweep.cpp:128-240 --------------- */
/* One list.  *sort_axis in: axis to sort on; out: arg-max centre variance (:176-195).
 * Returns number of pairs; *pairs is malloc'ed int32[2*n] (caller frees with orc_free). */
int64_t orc_sort_and_sweep(const orc_aabb* boxes, int n, int* sort_axis, int32_t** pairs,
                           int nthreads);
/* Two lists: pairs are (A element id, B element id). */
int64_t orc_sort_and_sweep_two_lists(const orc_aabb* boxesA, int nA, const orc_aabb* boxesB,
                                     int nB, int* sort_axis, int32_t** pairs, int nthreads);
/* Independent O(n^2) checker with the same predicate (no sort, no early break). */
int64_t orc_brute_force(const orc_aabb* boxesA, int nA, const orc_aabb* boxesB, int nB,
                        int32_t** pairs);
/* number of major-axis candidate tests the sweep performed in the last call (work metric) */
int64_t orc_last_candidate_tests(void);
void orc_free(void* p);
/* sort rows of an int32[n][2] pair list lexicographically in place */
void orc_sort_pairs(int32_t* pairs, int64_t n);

/* ---- narrow phase: src/scalable_ccd/cuda/narrow_phase/{narrow_phase,root_finder}.cu ----- */
typedef struct {
    int64_t n_queries;
    int64_t n_checks;       /* inclusion-function evaluations (ccd_kernel invocations that reach :313) */
    int64_t n_domains;      /* all ccd_kernel invocations incl. pruned ones                            */
    int64_t n_root_survive; /* queries whose root domain was split                                    */
    int64_t max_checks_per_query;
    int64_t max_queue;
} orc_np_stats;

/* Level-synchronous BFS exactly as root_finder.cu:372-457 (one global queue, global toi).
 * pairs: int32[n][2] rows (a,b); is_vf: (vertex,face) else (edge,edge).
 * toi is in/out (narrow_phase.cu:124-136).  per_query_toi (may be NULL): INFINITY-initialised
 * per-query minimum as in SCALABLE_CCD_TOI_PER_QUERY (narrow_phase.cu:69-73) -- when non-NULL
 * pruning uses the per-query value like the reference does in that build.
 * Returns 0 on success, ORC_E_BUDGET (toi = NaN) when one level would hold more than ORC_MAX_LEVEL_DOMAINS
 * live domains (1.9 GB): contact-rich query sets grow without bound in level order. */
#define ORC_MAX_LEVEL_DOMAINS ((int64_t)1 << 25)
#define ORC_E_BUDGET (-2)
void orc_set_level_budget(int64_t domains); /* tests: a smaller budget (<= 0 restores the default) */
int orc_narrow_phase(const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                     const int32_t* F, int nF, const int32_t* pairs, int64_t n, int is_vf,
                     double ms, int max_iter, double tol, int allow_zero_toi, int arith,
                     double* toi, double* per_query_toi, orc_np_stats* stats);

/* Same result for max_iter < 0, but per-query depth-first with a shared atomic toi and OpenMP
 * over queries: this is the multi-threaded CPU baseline (bench.py cpu_baseline, kind "port"). */
int orc_narrow_phase_mt(const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                        const int32_t* F, int nF, const int32_t* pairs, int64_t n, int is_vf,
                        double ms, int max_iter, double tol, int allow_zero_toi, int arith,
                        double* toi, int32_t* checks_per_query, int nthreads);

/* ccd(): src/scalable_ccd/cuda/ccd.cu:80-146 -- boxes (inflation = ms), VF pass, EE pass.
 * n_vf/n_ee (may be NULL) receive the overlap counts. */
int orc_ccd(const double* V0, const double* V1, int nV, const int32_t* E, int nE,
            const int32_t* F, int nF, double ms, int max_iter, double tol, int allow_zero_toi,
            int arith, int nthreads, double* toi, int64_t* n_vf, int64_t* n_ee);

/* single inclusion-function evaluation, exposed for unit tests of the arithmetic
 * (root_finder.cu:157-198).  v[24] = v0s,v1s,v2s,v3s,v0e,v1e,v2e,v3e; dom[6] = tlo,thi,ulo,uhi,vlo,vhi */
int orc_origin_in_inclusion_function(const double* v, const double* dom, const double* err,
                                     double ms, int is_vf, int arith, double* true_tol,
                                     int* box_in);
/* per-query constants (root_finder.cu:48-135) */
void orc_query_constants(const double* v, int is_vf, int use_ms, double co_domain_tol,
                         double* tol3, double* err3);

/* ---- SCALABLE_CCD_USE_DOUBLE = OFF: the same path with Scalar = float (scalar.hpp:13-21).
 * Vertices are cast to float first (aabb.cpp:43-47), then boxes, tolerances, error bounds (float filter
 * constants root_finder.cu:103-119), inclusion function, mid-points and the TOI are float arithmetic.
 * Float arrays in and out; boxes keep the 64-byte layout with the float values widened (every comparison
 * of the broad phase is exact on them, so orc_sort_and_sweep* serve both scalar types). */
void orc_build_vertex_boxes_f32(const float* V0, const float* V1, int nV, float inflation_radius, orc_aabb* out);
int orc_narrow_phase_f32(const float* V0, const float* V1, int nV, const int32_t* E, int nE, const int32_t* F, int nF,
                         const int32_t* pairs, int64_t n, int is_vf, float ms, int max_iter, float tol,
                         int allow_zero_toi, int arith, float* toi, float* per_query_toi, orc_np_stats* stats);
int orc_narrow_phase_mt_f32(const float* V0, const float* V1, int nV, const int32_t* E, int nE, const int32_t* F,
                            int nF, const int32_t* pairs, int64_t n, int is_vf, float ms, int max_iter, float tol,
                            int allow_zero_toi, int arith, float* toi, int32_t* checks_per_query, int nthreads);
int orc_ccd_f32(const float* V0, const float* V1, int nV, const int32_t* E, int nE, const int32_t* F, int nF, float ms,
                int max_iter, float tol, int allow_zero_toi, int arith, int nthreads, float* toi, int64_t* n_vf,
                int64_t* n_ee);
int orc_origin_in_inclusion_function_f32(const float* v, const float* dom, const float* err, float ms, int is_vf,
                                         int arith, float* true_tol, int* box_in);
void orc_query_constants_f32(const float* v, int is_vf, int use_ms, float co_domain_tol, float* tol3, float* err3);

#ifdef __cplusplus
}
#endif
#endif
