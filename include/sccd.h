/*
 * sccd.h -- C ABI of libsccd_hip.so: the MI355X (gfx950) implementation of the Scalable-CCD
 * hot path (STQ broad phase + Tight-Inclusion narrow phase).
 *
 * The reference (Continuous-Collision-Detection/Scalable-CCD) has no FFI layer; its boundary
 * is the C++ host API under src/scalable_ccd/cuda/.  Each entry point below names the
 * reference interface it replaces (file:line relative to the reference root).  The C++ header
 * include/scalable_ccd/hip/ccd.hpp re-creates those C++ signatures on top of this ABI.
 *
 * Conventions
 *  - every call returns SCCD_OK (0) or a negative error code; sccd_last_error() gives the text.
 *  - inputs are borrowed; host matrices are COLUMN-MAJOR (Eigen default), V: n x 3 double,
 *    E: m x 2 int32, F: k x 3 int32 -- exactly the storage of the Eigen arguments of ccd().
 *  - Scalar = double (reference default SCALABLE_CCD_USE_DOUBLE=ON, CMakeLists.txt:69).
 *  - calls are blocking on the context's stream unless stated; a context is not thread-safe,
 *    different contexts are independent (no global state, unlike the reference's
 *    __constant__ CONFIG in root_finder.cu:19).
 *  - there is NO CPU fallback: without a HIP device sccd_create fails with SCCD_E_NO_DEVICE.
 */
#ifndef SCCD_H
#define SCCD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCCD_OK 0
#define SCCD_E_INVALID (-1)    /* bad argument (reference: assert / std::runtime_error)          */
#define SCCD_E_NO_DEVICE (-2)  /* no HIP device / device out of range                           */
#define SCCD_E_HIP (-3)        /* HIP runtime error (reference: gpuErrchk -> std::runtime_error) */
#define SCCD_E_NOMEM (-4)      /* out of device memory (memory_handler.cpp:66-68)               */
#define SCCD_E_NOT_BUILT (-5)  /* detect_overlaps before build (broad_phase.cu:123-126)         */
#define SCCD_E_OVERFLOW (-6)   /* internal work-queue capacity exhausted after retries          */

typedef struct sccd_ctx sccd_ctx;
typedef struct sccd_mesh sccd_mesh;
typedef struct sccd_boxes sccd_boxes;
typedef struct sccd_broad_phase sccd_broad_phase;

/* 64-byte box, bit-compatible with scalable_ccd::cuda::AABB
 * (src/scalable_ccd/cuda/broad_phase/aabb.cuh:82-92). */
typedef struct sccd_aabb {
    double min[3];
    double max[3];
    int32_t vertex_ids[3];
    int32_t element_id;
} sccd_aabb;

/* (aid, bid, toi) of SCALABLE_CCD_TOI_PER_QUERY builds (narrow_phase.cu:84-103). */
typedef struct sccd_collision {
    int32_t aid;
    int32_t bid;
    double toi;
} sccd_collision;

/* ------------------------------------------------------------------------------------------ */
/* context                                                                                    */

/* (Contexts are independent and may be used from different host threads.  One performance note: while more than two contexts are
 *  alive in the process -- a context and the helper context its ccd() makes are two -- the radix sort's passes take their tiles by
 *  atomic ticket instead of by block index, ~2 us per pass slower: the block-index form is only argued for two concurrent sorts.) */
int sccd_create(int device, sccd_ctx** out);
/* ABI CHECK.  sccd_stats and the SCCD_PROF_* arrays are written by the library into memory the caller sized at ITS compile time
 * (sccd_ccd_mesh*, sccd_get_profile): a caller built against another version of this header compares sizeof(sccd_stats) and
 * SCCD_PROF_COUNT with what the loaded library was built with before it hands such memory over (0.2 -> 0.3 grew both without a
 * way to ask; the Python binding and include/scalable_ccd/hip/ccd.hpp refuse a library that disagrees). */
size_t sccd_abi_sizeof_stats(void);
int sccd_abi_prof_count(void);
void sccd_destroy(sccd_ctx* ctx);
const char* sccd_last_error(const sccd_ctx* ctx);
const char* sccd_version(void);
/* hipStream_t to launch on (e.g. torch's current stream); NULL = the context's own stream. */
int sccd_set_stream(sccd_ctx* ctx, void* hip_stream);
int sccd_synchronize(sccd_ctx* ctx);
/* the hipStream_t the context launches on (its own, or the one handed to sccd_set_stream) */
void* sccd_get_stream(const sccd_ctx* ctx);
/* Self-test of the hand-written memory idioms of the narrow-phase kernel (no reference counterpart): n_waves
 * wavefronts each gather n_active (0..64) 48-byte vertex records into LDS by LDS-direct loads with a deliberately
 * late wait while other LDS traffic runs.  *n_bad = number of LDS / memory words that differ from the expected layout (0 = pass). */
int sccd_selftest_lds_gather(sccd_ctx* ctx, int n_waves, int n_active, int64_t* n_bad);

/* options (sccd_set_option) */
#define SCCD_OPT_ARITH 1            /* 1 (default since 0.2): a*b+c of root_finder.cu:137-155 fused -- the reference's CUDA build is
                                       compiled with --use_fast_math (CMakeLists.txt:219-225), i.e. -fmad=true; 0 strict: *, +/- rounded separately */
#define SCCD_OPT_NARROW_ALGO 2      /* 0 per-wave work queues (default); 1 level-synchronous BFS (root_finder.cu:431-447) */
#define SCCD_OPT_SWEEP_ALGO 3       /* 0 = 2 filter/queue/confirm STQ (default); 1 plain sweep-and-prune (sweep.cu:48-99); 3 direct exact sweep */
#define SCCD_OPT_SORT_AXIS 4        /* 0/1/2 = x/y/z (reference device path: x, aabb.cu:85-86); -1 = arg-max variance   */
#define SCCD_OPT_SHARD_RANK 5       /* multi-GPU: this rank's index ...                                                  */
#define SCCD_OPT_SHARD_COUNT 6      /* ... of this many ranks; a rank sorts and sweeps only its window of grid cells      */
#define SCCD_OPT_OVERLAP_CAPACITY 7 /* initial overlap buffer capacity in pairs (0 = automatic)                           */
#define SCCD_OPT_PROFILE 8          /* hipEvents around kernel classes (sccd_get_profile): 1 = all, (class mask) << 1 = some  */
#define SCCD_OPT_MAX_OVERLAP_CUTOFF 9 /* boxes swept per detect_overlaps_partial call (0 = all; memory_handler.hpp:9)      */
#define SCCD_OPT_MEMORY_LIMIT_MB 10 /* memory budget of the overlap list in MiB (0 = none; MemoryHandler::memory_limit_GB)  */
#define SCCD_OPT_SCALAR 11          /* 0 double (default, SCALABLE_CCD_USE_DOUBLE=ON); 1 float (=OFF, scalar.hpp:13-21): vertices are
                                       cast to float first, boxes / tolerances / inclusion function / TOI are float arithmetic
                                       (values travel widened in the same double-typed interfaces); narrow phase on a depth-first
                                       kernel of its own (np_walk_f32_k; check limits on the level-synchronous kernels), bit-equal to the
                                       oracle's float twin on the GPU (tests/test_gpu_parity.py) */
#define SCCD_OPT_PASSES_APART 13      /* ccd(): 1 = the vertex-face and the edge-edge pass one after the other on one stream, the host between
                                       * them: measurements of the passes' own durations (bench.py --passes-apart) */
/* id 12 is RETIRED (it was SCCD_OPT_MAX_ITER_FAST in 0.1 with the opposite sense: setting it now fails with SCCD_E_INVALID) */
#define SCCD_OPT_LIMIT_LEVEL_ORDER 14 /* check limits (max_iter >= 0, root_finder.cu:287-305): 0 (default) the fast kernel runs
                                       * without the limit and the library proves that the limit would not have changed the
                                       * answer (one query redone in the reference's level order on the host), falling back to the
                                       * level-synchronous kernels where the proof fails; 1: level-synchronous kernels always
                                       * (cross-check).  Either way the result is that of the reference's level order. */
#define SCCD_OPT_CELL_FACTOR_MILLI 17 /* grid cell size in thousandths of the mean box extent per minor axis (0 = default: 4000; < 0: one cell).
                                       * Any value is correct (grid.hpp); the SCCD_CELL_FACTOR environment variable of 0.1 is gone */
#define SCCD_OPT_BUILD_SCAN 18        /* 1: entries by count -> device-wide scan -> fill, in box order (reproducible entry order; was SCCD_BUILD=scan) */
#define SCCD_OPT_TOI_GUESS 19         /* 1 (default): sccd_ccd_mesh / sccd_ccd on a mesh whose previous call found an impact at T start from the bound
                                       * min(1, 1.125 T) instead of 1 and verify (a result below the bound is exact; a result at the bound: the step is
                                       * redone from 1); 0: always from 1, as ccd.cu:125.  The same history also settles SCCD_OPT_TWO_HALVES' bet. */
#define SCCD_OPT_TOI_GUESS_HITS 20    /* read: calls whose bound held / broke (redone); setting either resets both */
#define SCCD_OPT_TOI_GUESS_MISSES 21
#define SCCD_OPT_CULL 24              /* sccd_ccd / sccd_ccd_mesh / sccd_ccd_mesh_pass / sccd_ccd_collisions / sccd_ipc_ccd_strategy (since 0.4: both scalar
                                       * builds, with or without a check limit -- a culled pair has no acceptable domain under any traversal): 1 (default) the overlap pairs of a pass
                                       * go through the PROJECTION CULL before the bisection -- a pair is dropped if some direction d puts the eight
                                       * corner values of d . F (F: the collision function, affine in each of t, u, v) all beyond the reach of any
                                       * domain the reference's bisection could accept (csrc/narrow_cull.inc): the result is unchanged, the narrow
                                       * phase bisects a fraction of the pairs; 0: every pair is bisected, as root_finder.cu:372-457 does.
                                       * The cull looks at the slab of time the pass's narrow launch asks about ([0, b] for a start from b; the halves
                                       * of SCCD_OPT_TWO_HALVES each have their own).  1 means "where it pays": meshes of 50,000 edges + faces or more (a
                                       * launch per sweep costs a small step more than it saves); 2: always.
                                       * sccd_narrow_phase culls a caller's list of 100,000 pairs or more (2: any list) when no collision
                                       * records are asked for (their order is the list's): the same TOI, a fraction of the bisection. */
#define SCCD_OPT_TWO_HALVES 25        /* 1 (default): a narrow-phase launch of the plain walk kernel (double build, no check limit, no per-query output) that
                                       * starts from a TOI above 0.5 runs as two launches over the halves of time: the first from the bound 0.5; if it
                                       * accepts nothing, the second bisects what lies at or beyond 0.5, from the caller's TOI (csrc/narrow_walk.inc).
                                       * The same accepted domains as one launch (root_finder.cu:277-370), hence the same result; a call whose earliest
                                       * impact lies before 0.5 never explores the later halves of its first time splits.  0: one launch.
                                       * The two launches are a BET on an early impact: where the earliest impact lies at or beyond 0.5, or there is none, they
                                       * cost a second cull and a second launch per pass (+ 10-20 %).  Under 1, sccd_ccd / sccd_ccd_mesh take the bet only
                                       * for meshes of 600,000 edges + faces or more and -- with SCCD_OPT_TOI_GUESS on -- only if the last call on the
                                       * same mesh did not return a TOI >= 0.5; 2: always. */
#define SCCD_OPT_ALLOC_COUNT 23       /* read: device allocations the library's grow-only buffers have made since it was loaded (all contexts): a call
                                       * during which the count rises has grown a buffer -- hipFree + hipMalloc, milliseconds */
#define SCCD_OPT_DEVICE_SPAN_NS 26    /* read: what the DEVICE spent on the last sccd_ccd / sccd_ccd_mesh* call of this context, in nanoseconds of its own
                                       * real-time clock: from the start of the call's first kernel to the end of the kernel behind the last read-back
                                       * the host waited for (-1: no such call yet).  Beside the caller's own clock it separates a step the chip was
                                       * slow on from a step the host was late for (bench.py: device_span_ms) */
#define SCCD_OPT_HOST_WAITS 27        /* read: how often the host has waited for the device (read-backs of counters, early verdicts) since the context
                                       * was made, its helper context included: the difference across a call is the number of host round trips in it */
#define SCCD_OPT_READ_BACKS 28        /* read: of those waits, the read-back launches (a gather kernel + a polled word: ~10 us each on the call's critical
                                       * path).  A default sccd_ccd_mesh step on a mesh the context has stepped before needs NONE: both passes' verdicts
                                       * arrive in pinned memory behind their walk kernels, with everything the host has to look at (csrc/drivers.hip) */
#define SCCD_OPT_SPEC_HITS 15   /* read: speculative builds (sort, records and sweep enqueued for the previous build's entry counts) whose */
#define SCCD_OPT_SPEC_MISSES 16 /* guess held / broke and were redone, since the context was made; setting either resets both counters */
int sccd_set_option(sccd_ctx* ctx, int option, int64_t value);
int64_t sccd_get_option(const sccd_ctx* ctx, int option);

/* ------------------------------------------------------------------------------------------ */
/* device memory == thrust::device_vector<T> storage of DeviceMatrix<T> (utils/device_matrix.cuh:10-66)  */

/* bytes of device memory on the context's device (0 bytes gives a NULL pointer); blocking copies */
int sccd_dev_alloc(sccd_ctx* ctx, size_t bytes, void** d_ptr);
int sccd_dev_free(sccd_ctx* ctx, void* d_ptr);
int sccd_dev_upload(sccd_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int sccd_dev_download(sccd_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int sccd_dev_copy(sccd_ctx* ctx, void* d_dst, const void* d_src, size_t bytes); /* device to device (thrust::device_vector::resize keeps its contents) */

/* ------------------------------------------------------------------------------------------ */
/* mesh  == the four DeviceMatrix objects of ccd() (src/scalable_ccd/cuda/ccd.cu:103-106)     */

/* src_on_device: 0 = host pointers, 1 = device pointers (same column-major layout).
   Vertex indices of E and F are validated on the device while the matrices are packed: an index outside [0, nV) makes the
   call fail with SCCD_E_INVALID ("index out of range"; the reference asserts nothing and would fault). */
int sccd_mesh_create(sccd_ctx* ctx, const double* V0, const double* V1, int nV, const int32_t* E,
                     int nE, const int32_t* F, int nF, int src_on_device, sccd_mesh** out);
int sccd_mesh_update_vertices(sccd_mesh* mesh, const double* V0, const double* V1, int src_on_device);
/* all four matrices again, into the mesh's own buffers (grown where needed): what narrow_phase() with the reference's
   argument list -- four DeviceMatrix per call, narrow_phase.cuh:30-46 -- does without allocating per call */
int sccd_mesh_assign(sccd_mesh* mesh, const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                     const int32_t* F, int nF, int src_on_device);
void sccd_mesh_destroy(sccd_mesh* mesh);

/* ------------------------------------------------------------------------------------------ */
/* boxes                                                                                      */

/* build_vertex_boxes(V0, V1, boxes, inflation_radius): aabb.cuh:166-170 / aabb.cu:146-184.
 * Host in, host out; computed on the device. */
int sccd_build_vertex_boxes(sccd_ctx* ctx, const double* V0, const double* V1, int nV,
                            double inflation_radius, sccd_aabb* out);
/* build_edge_boxes / build_face_boxes: aabb.cuh:176-188 / aabb.cu:186-229. */
int sccd_build_edge_boxes(sccd_ctx* ctx, const sccd_aabb* vertex_boxes, int nV, const int32_t* E,
                          int nE, sccd_aabb* out);
int sccd_build_face_boxes(sccd_ctx* ctx, const sccd_aabb* vertex_boxes, int nV, const int32_t* F,
                          int nF, sccd_aabb* out);

/* DeviceAABBs(const std::vector<AABB>&): aabb.cuh:122-150 / aabb.cu:75-111 -- upload, split
 * into major-axis keys + payload, sort by min on the sort axis. */
int sccd_boxes_create(sccd_ctx* ctx, const sccd_aabb* boxes, int n, int src_on_device,
                      sccd_boxes** out);
/* Fused device path used by ccd(): vertex/edge/face boxes straight from the mesh, sorted
 * (ccd.cu:112-121 without the host round trip).  Any of the three outputs may be NULL. */
int sccd_boxes_from_mesh(sccd_ctx* ctx, const sccd_mesh* mesh, double inflation_radius,
                         sccd_boxes** vertex_boxes, sccd_boxes** edge_boxes,
                         sccd_boxes** face_boxes);
int sccd_boxes_size(const sccd_boxes* boxes);
/* copy the sorted boxes back (sorted order along the sort axis) */
int sccd_boxes_download(const sccd_boxes* boxes, sccd_aabb* out);
void sccd_boxes_destroy(sccd_boxes* boxes);

/* ------------------------------------------------------------------------------------------ */
/* broad phase == class BroadPhase (src/scalable_ccd/cuda/broad_phase/broad_phase.cuh:15-92)  */

int sccd_broad_phase_create(sccd_ctx* ctx, sccd_broad_phase** out);
void sccd_broad_phase_destroy(sccd_broad_phase* bp);
/* build(boxes) (B == NULL) and build(boxesA, boxesB): broad_phase.cu:29-101.  The boxes are
 * shared, not consumed (the reference clears them on the next build -- a quirk not kept). */
int sccd_broad_phase_build(sccd_broad_phase* bp, const sccd_boxes* A, const sccd_boxes* B);
/* detect_overlaps_partial(): broad_phase.cu:121-224.  *d_pairs is a DEVICE pointer to n rows
 * of int32[2], valid until the next call on bp.  One list: (min id, max id); two lists:
 * (A id, B id).  Order unspecified. */
int sccd_broad_phase_detect_overlaps_partial(sccd_broad_phase* bp, const int32_t** d_pairs,
                                             int64_t* n);
/* detect_overlaps(): broad_phase.cu:226-252.  *pairs is malloc'ed host memory (sccd_free). */
int sccd_broad_phase_detect_overlaps(sccd_broad_phase* bp, int32_t** pairs, int64_t* n);
int sccd_broad_phase_is_complete(const sccd_broad_phase* bp); /* broad_phase.cuh:57 */
int64_t sccd_broad_phase_num_boxes(const sccd_broad_phase* bp); /* broad_phase.cuh:63 */
/* number of sort-axis candidate tests of the last detect call (work metric) */
int64_t sccd_broad_phase_candidates(const sccd_broad_phase* bp);
/* The axis the reference's CPU sort_and_sweep() hands back for the NEXT call: arg-max over x, y, z
 * of sum(c^2) - sum(c)^2 / n of the box centres c = (min + max) / 2 of all boxes of A (and B),
 * ties to the lower axis (broad_phase/sort_and_sweep.cpp:176-195).  B may be NULL. */
int sccd_boxes_variance_axis(sccd_ctx* ctx, const sccd_boxes* A, const sccd_boxes* B, int* axis);
void sccd_free(void* host_ptr);

/* ------------------------------------------------------------------------------------------ */
/* narrow phase == narrow_phase<is_vf>() (src/scalable_ccd/cuda/narrow_phase/narrow_phase.cuh:30-46) */

/* pairs: n rows int32[2]; is_vf: (vertex, face) else (edge, edge).  *toi is in/out and must be
 * >= 0 (narrow_phase.cu:126).  collisions (may be NULL) receives a malloc'ed list of the
 * queries with toi < 1 as in SCALABLE_CCD_TOI_PER_QUERY builds.  Without `collisions` the list goes through the projection
 * cull first (SCCD_OPT_CULL): the result is that of the whole list. */
int sccd_narrow_phase(sccd_ctx* ctx, const sccd_mesh* mesh, const int32_t* pairs, int64_t n,
                      int pairs_on_device, int is_vf, int max_iter, double tol, double ms,
                      int allow_zero_toi, double* toi, sccd_collision** collisions,
                      int64_t* n_collisions);

/* ------------------------------------------------------------------------------------------ */
/* drivers                                                                                    */

typedef struct sccd_stats {
    int64_t n_vf_pairs, n_ee_pairs;           /* overlaps fed to the narrow phase             */
    int64_t n_vf_candidates, n_ee_candidates; /* sort-axis candidate tests                    */
    int64_t n_vf_checks, n_ee_checks;         /* inclusion-function evaluations               */
    double ms_boxes, ms_sort, ms_sweep, ms_narrow, ms_total; /* device time, SCCD_OPT_PROFILE */
    int64_t n_vf_culled, n_ee_culled;         /* (0.3) of those overlaps: dropped by the projection cull before the bisection
                                               * (SCCD_OPT_CULL: pairs whose collision function is provably never within reach of the
                                               * origin -- no domain of theirs can be accepted, so they cannot change the result) */
} sccd_stats;

/* The projection cull on its own (csrc/narrow_cull.inc; what sccd_ccd / sccd_ccd_mesh run between a pass's sweep and its bisection under
 * SCCD_OPT_CULL): of the n overlap pairs (host int32[2n]; vertex-face: (vertex, face), edge-edge: (edge, edge)) those that MAY have an
 * impact are copied to kept (host int32[2n], any order), *n_kept of them.  Every pair that is left out provably has no domain the
 * reference's bisection (root_finder.cu:277-370) could accept under the given minimum separation and tolerance.  No reference
 * counterpart; exported so that the claim can be tested against the oracle's per-query output (tests/test_gpu_parity.py). */
int sccd_query_cull(sccd_ctx* ctx, const sccd_mesh* mesh, const int32_t* pairs, int64_t n, int is_vf, double min_distance, double tolerance,
                    int32_t* kept, int64_t* n_kept);
/* ... for one SLAB OF TIME, as ccd() runs it (narrow_cull.inc, "slabs of time"): a narrow launch that starts from the bound b only
 * accepts domains that begin before b, the second launch of the "two halves of time" (SCCD_OPT_TWO_HALVES) only domains that end
 * after 0.5.  kept = the pairs that may have an accepted domain meeting t in [t_lo, t_hi], 0 <= t_lo < t_hi <= 1 ((0, 1):
 * sccd_query_cull).  Against the oracle's per-query output: a pair that is left out has no earliest impact in [t_lo, t_hi). */
int sccd_query_cull_slab(sccd_ctx* ctx, const sccd_mesh* mesh, const int32_t* pairs, int64_t n, int is_vf, double min_distance,
                         double tolerance, double t_lo, double t_hi, int32_t* kept, int64_t* n_kept);

/* ccd(V0,V1,E,F,min_distance,max_iterations,tolerance,allow_zero_toi,memory_limit_GB):
 * src/scalable_ccd/cuda/ccd.cuh:26-38 / ccd.cu:80-146.  Host matrices in, earliest TOI out.
 * STATEFUL BY DEFAULT (SCCD_OPT_TOI_GUESS = 1): the RESULT never depends on earlier calls, the LATENCY does -- a call on the mesh
 * object and sizes of the previous call of this context starts from 1.125 x that call's TOI (and is redone from 1 if nothing lies
 * below the bound: exact either way) and settles SCCD_OPT_TWO_HALVES' bet by that call's result; broad-phase buffers and the
 * speculative build's entry counts are kept from call to call as well (the first call on a context allocates and builds the slow
 * way).  SCCD_OPT_TOI_GUESS = 0 makes every call start from 1 as ccd.cu:125 does; sccd_ccd_mesh_from takes the caller's bound.
 * (The matrices go into a mesh the context owns and refills call after call; an index out of range is reported when the
 *  call ends -- the step has then run on indices clamped to 0, and its result is discarded.) */
int sccd_ccd(sccd_ctx* ctx, const double* V0, const double* V1, int nV, const int32_t* E, int nE,
             const int32_t* F, int nF, double min_distance, int max_iterations, double tolerance,
             int allow_zero_toi, int memory_limit_GB, double* toi);
/* ccd() of a SCALABLE_CCD_TOI_PER_QUERY build (ccd.cuh:26-38 with `collisions`, ccd.cu:14-78): additionally hands
 * back the (aid, bid, toi) records of every query with toi < 1 -- the vertex-face pass's, then the edge-edge pass's
 * (a malloc'ed list, sccd_free).  The overlap pairs never leave the device between the two phases. */
int sccd_ccd_collisions(sccd_ctx* ctx, const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                        const int32_t* F, int nF, double min_distance, int max_iterations, double tolerance,
                        int allow_zero_toi, int memory_limit_GB, double* toi, sccd_collision** collisions,
                        int64_t* n_collisions);
/* Same on a device-resident mesh; stats may be NULL.  This is what bench.py times. */
int sccd_ccd_mesh(sccd_ctx* ctx, const sccd_mesh* mesh, double min_distance, int max_iterations,
                  double tolerance, int allow_zero_toi, double* toi, sccd_stats* stats);
/* The same step with its result ALSO left in device memory: *d_toi (8 bytes of device memory, a double) receives the TOI by a
 * copy enqueued on the context's stream (sccd_get_stream), so that a multi-GPU caller can all-reduce(min) that word in place,
 * stream-ordered behind the step, without the host waiting for the collective (bench.py, sccd/dist.py allreduce_min_device).
 * toi (host) may be NULL.  No reference counterpart: the reference has no working multi-device path
 * (src/scalable_ccd/cuda/broad_phase/CMakeLists.txt:21 leaves _multigpu out of the build). */
int sccd_ccd_mesh_dev(sccd_ctx* ctx, const sccd_mesh* mesh, double min_distance, int max_iterations,
                      double tolerance, int allow_zero_toi, double* d_toi, double* toi, sccd_stats* stats);
/* ccd() on a resident mesh FROM A CALLER'S BOUND in (0, 1]: *toi = min(bound, the earliest impact below it) -- narrow_phase's toi is in / out
 * (narrow_phase.cu:126); a result below the bound is bit for bit what a start from 1 returns, a result AT the bound says only that nothing
 * lies below it.  The context's own history (SCCD_OPT_TOI_GUESS) is neither used nor updated.  For callers that hold a better bound than the
 * context can: the ranks of a multi-GPU job start from 1.125 x the REDUCED result of their last step (sccd/dist.py GlobalPrior).  Calls with
 * a check limit start from 1 whatever the bound (the limit's certificate is about the TOI a call started with). */
int sccd_ccd_mesh_from(sccd_ctx* ctx, const sccd_mesh* mesh, double min_distance, int max_iterations, double tolerance, int allow_zero_toi,
                       double bound, double* toi, sccd_stats* stats);
/* The two halves of ccd() for multi-GPU runs (one process per GPU).  A rank (SCCD_OPT_SHARD_RANK / SCCD_OPT_SHARD_COUNT) owns a
 * contiguous window of grid cells cut ON THE DEVICE from a sampled cell histogram; prepare builds the vertex boxes only (the
 * edge and face boxes of a sharded call are computed inside the fill, and stored only where they fall into the rank's
 * window); pass builds, sorts and sweeps the window and runs the narrow phase on the pairs it found.  A pair belongs to
 * exactly one cell, hence to one rank: the only exchange is an all-reduce(min) of *toi (RCCL), once after the EE pass
 * (optionally also after the VF pass: ccd.cu:125-143 threads toi from the VF pass into the EE pass, which only prunes). */
int sccd_ccd_mesh_prepare(sccd_ctx* ctx, const sccd_mesh* mesh, double min_distance);
int sccd_ccd_mesh_pass(sccd_ctx* ctx, const sccd_mesh* mesh, int is_vf, double min_distance,
                       int max_iterations, double tolerance, int allow_zero_toi, double* toi_inout,
                       sccd_stats* stats);
/* ipc_ccd_strategy(): src/scalable_ccd/cuda/ipc_ccd_strategy.hpp:17-24 / .cu:97-152. */
int sccd_ipc_ccd_strategy(sccd_ctx* ctx, const double* V0, const double* V1, int nV,
                          const int32_t* E, int nE, const int32_t* F, int nF, double min_distance,
                          int max_iterations, double tolerance, double* toi);

/* ------------------------------------------------------------------------------------------ */
/* profiling (replaces the reference's Profiler/ProfilePoint, utils/profiler.hpp:15-99)       */

#define SCCD_PROF_BOXES 0
#define SCCD_PROF_SORT 1
#define SCCD_PROF_CULL 2   /* (0.3) np_cull_k launches: the projection cull in front of a pass's bisection (0.2: SCCD_PROF_RANGES, a kernel that is gone) */
#define SCCD_PROF_RANGES 2 /* (the old name of slot 2) */
#define SCCD_PROF_SWEEP 3
#define SCCD_PROF_NARROW_VF 4 /* np_walk_k<true> / np_level_k<true> launches  */
#define SCCD_PROF_NARROW_EE 5 /* np_walk_k<false> / np_level_k<false> launches */
#define SCCD_PROF_SWEEP_EE 6  /* (0.3, late) the sweep of a mesh's EDGE list -- the longest kernel of a ccd() step -- apart from the other sweeps (slot 3) */
#define SCCD_PROF_COUNT 7
/* accumulated device milliseconds and launch counts per kernel class since the last reset */
int sccd_get_profile(sccd_ctx* ctx, double ms[SCCD_PROF_COUNT], int64_t launches[SCCD_PROF_COUNT]);
int sccd_reset_profile(sccd_ctx* ctx);

/* Host-side split behind SCCD_OPT_SHARD_*: cuts n weighted items (grid cells weighted by their
 * entry counts) into `parts` contiguous windows of nearly equal weight.  bounds has parts+1
 * entries, bounds[0] = 0, bounds[parts] = n; window r is [bounds[r], bounds[r+1]).  Needs no GPU. */
int sccd_shard_bounds(const uint32_t* weights, int n, int parts, int* bounds);

/* standalone kernels exposed for roofline measurements (bench.py --workload sort) */
int sccd_sort_pairs_u32(sccd_ctx* ctx, uint32_t* d_keys, uint32_t* d_vals, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* SCCD_H */
