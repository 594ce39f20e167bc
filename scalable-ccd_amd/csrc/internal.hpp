// internal.hpp -- object layouts in HBM and the launcher functions each .hip file exports.
#pragma once
#include "common.hpp"
#include <functional>
#include "grid.hpp"

// ------------------------------------------------------------------------------------------
// HBM layouts (DESIGN.md "Data layout in HBM")
//
// mesh.V   : double[nV][6]  = {x0,y0,z0, x1,y1,z1}  -- both frames of a vertex in ONE 48-byte
//            record of three 16-byte pieces (the unit of the narrow phase's LDS-direct loads), so
//            a query gathers 4 records (reference: two column-major matrices, 24 separate
//            8-byte gathers per query, narrow_phase.cu:44-66)
// mesh.E   : int2[nE], mesh.F : int4[nF] = {f0,f1,f2,0}   (row records instead of columns)
//
// sweep list (SortedList): one ENTRY per (box, overlapped cell), sorted by the composite key of grid.hpp.
//   An entry is an 80-byte record kept as five 16-byte pieces in five arrays (structure of arrays: the record
//   builder's stores, the sweep's row loads and its window staging are all consecutive 16-byte accesses):
//   ra   : double2[m] {min, max} on minor axis a           (reference: MiniBox min / max)
//   rb   : double2[m] {min, max} on minor axis b
//   rx   : double2[m] {min, max} on the sort axis         (reference: Scalar2 sorted_major_intervals)
//   rid  : int4[m]    {vertex ids, element id}             (reference: MiniBox vertex_ids, element_id)
//   raux : uint4[m]   {key, key of the max on the sort axis, lowest cell per minor axis (a | b << 16),
//                      first candidate column (two-list sweeps; one list: row + 1)}
struct sccd_mesh {
    sccd_ctx* ctx = nullptr;
    int nV = 0, nE = 0, nF = 0;
    DevBuf V, E, F;
};

// DeviceAABBs: the boxes of one list, resident in HBM in element order.  The split into keys and
// payload and the sort (aabb.cu:75-111 does them in the DeviceAABBs constructor) happen in
// BroadPhase::build, where the cell grid of BOTH lists is known.
// what is known about a list's vertex ids: boxes built HERE from a mesh carry the ids of aabb.cpp:57-58,107-109,128-130
enum BoxKind { BOX_UNKNOWN = 0, BOX_VERTEX = 1 /* {i, -i-1, -i-1} */, BOX_EDGE = 2 /* {e0, e1, -e0-1} */, BOX_FACE = 3 /* {f0, f1, f2} */ };
struct sccd_boxes {
    sccd_ctx* ctx = nullptr;
    int n = 0;
    int kind = BOX_UNKNOWN; // (uploaded boxes: unknown)
    DevBuf raw; // sccd_aabb[n]
    // LAZY list (ccd() of a multi-GPU rank): `raw` is allocated but only the boxes of the rank's window of cells get written,
    // by the fill pass, which computes every box from the vertex boxes and the element's vertex indices (boxes.hip BoxSrc)
    bool lazy = false;
    const sccd_aabb* lazy_vb = nullptr; // the vertex boxes
    const void* lazy_elems = nullptr;   // int2[n] edges / int4[n] faces
    // bounds + extent partials of the list ({GridStats, pad to 128 B, double[n_part][3]}): written by the
    // box builders themselves (fused) or, for uploaded boxes, by box_stats_k on first use
    mutable DevBuf stats;
    mutable int n_part = 0;
    mutable bool have_stats = false;
    GridStats* stats_head() const { return stats.as<GridStats>(); }
    double* stats_part() const { return reinterpret_cast<double*>(stats.as<char>() + 128); } // [n_part][9]: lo, hi, extent sums
};
constexpr size_t SCCD_STATS_BYTES = 128 + sizeof(double) * 9 * SCCD_STATS_BLOCKS;

// one sweep list: an entry per (box, overlapped cell), sorted by the composite key of grid.hpp
struct SortedList {
    int m = 0; // entries (>= number of boxes)
    int kind = BOX_UNKNOWN;    // of the boxes the entries were made from
    DevBuf key, idx;           // the (key, box index) pairs the radix sort orders
    DevBuf recs;               // the sorted records: five arrays of 16-byte pieces, `pstride` entries apart (see above)
    uint32_t pstride = 0;
    DevBuf offsets; // uint32[n boxes]: first entry of each box (scanned cell counts), build scratch
};
// what the sweep kernels take: the five piece arrays of one list -- ONE allocation, piece p of entry e at
// base[p * pstride + e], in the order the sweep stages them: ra, rb, rx, rid, raux (a wave-uniform base and a 32-bit
// offset per lane address any piece: no 64-bit address arithmetic in the kernels)
enum { REC_A = 0, REC_B = 1, REC_X = 2, REC_ID = 3, REC_AUX = 4 };
struct SweepRecs {
    const uint4* base;
    uint32_t pstride;
};
inline SweepRecs sweep_recs(const SortedList* L) { return SweepRecs { L->recs.as<uint4>(), L->pstride }; }
constexpr size_t SCCD_LIST_PAD = 64; // entries allocated past the last one: the sweep stages whole 32-column segments

// ccd() on small meshes: the sizes (edges + faces) from which the projection cull and the two halves of time are used under their
// default settings (drivers.hip: pass_cull_setup, ccd_on_mesh)
constexpr long long SCCD_CULL_MIN_ELEMENTS = 50000, SCCD_TWO_HALVES_MIN_ELEMENTS = 600000;
constexpr long long SCCD_CULL_MIN_PAIRS = 100000; // sccd_narrow_phase on a caller's list: the cull is a launch and a read-back, ~15 us
// ... and the size from which the edge list's records kernel is ordered behind the end of the vertex + face one (drivers.hip, the
// records gate): below it the cross-queue wait costs the 5-10 us a step gains above it (cloths of 100 k - 200 k triangles: +2 %)
constexpr long long SCCD_RECORDS_GATE_MIN_ELEMENTS = 600000;
// The slabs of the step a pass's projection cull is run for (narrow_cull.inc, "slabs of time"): [0, t_end], or -- for the two launches
// of the walk kernel's "two halves of time" -- [0, t_mid] behind the sweep and [t_mid, t_end] between the two launches.
struct CullSlabs {
    bool two = false;
    double t_mid = 0.5, t_end = 1.0;
};
// the grid parameters and, right behind them, the two list totals of a build: ONE copy brings both back (two copies in a row
// cost a 12 us bubble between them)
struct GridReadBack {
    GridParams gp;
    uint32_t total[2];
    uint32_t place; // (device only: the shared placement cursor of a merged two-list fill)
    uint32_t ext_q; // (device only: list A's largest extent along the sort axis, quantised -- the one-class two-list sweep)
};
struct ShardWindow;
struct sccd_broad_phase {
    sccd_ctx* ctx = nullptr;
    const sccd_boxes* A = nullptr;
    const sccd_boxes* B = nullptr; // nullptr: one list
    bool built = false;
    int64_t cursor = 0;     // thread_start_box_id of broad_phase.cuh:86 (in sorted rows)
    int64_t total_rows = 0; // rows of all sweep classes
    SortedList la, lb;         // sorted entry lists of A and B
    DevBuf grid;               // GridStats + GridParams
    DevBuf overlaps;           // int2[capacity]
    int64_t capacity = 0;
    int64_t n_overlaps = 0;
    int64_t candidates = 0;      // sort-axis candidate tests of the build so far (all chunks)
    int64_t candidates_done = 0; // ... of the chunks before the current one
    int cell_lo = 0, cell_hi = 1 << 30; // this rank's window of cells (multi-GPU shard)
    bool row_shard = false;             // too few cells to shard by: split the rows instead
    // The SPECULATIVE build (api.hip bp_build): sort, records and sweep are enqueued right behind the fill, for the entry
    // counts and the key width of the previous build of the same lists plus a margin; the kernels take the real counts from
    // device memory; the host looks at them when the sweep's counters come back and builds again, the slow way, if the guess
    // did not hold.  (Round 2 waited for the counts between the fill and the sort: 25-30 us of every build chain.)
    struct Guess {
        bool valid = false;
        int n_a = 0, n_b = 0, axis = 0, key_bits = 0;
        double cell_factor = 0;
        uint32_t total[2] = { 0, 0 };
    } guess;
    // THE PROJECTION CULL of ccd() (narrow_cull.inc): with cull.on every sweep bp_detect_partial launches is followed, on the same
    // stream, by a kernel that drops the pairs whose Tight-Inclusion bisection provably accepts nothing and compacts the others
    // into `kept`; the count comes back with the sweep's counters (n_kept).  `overlaps` / n_overlaps stay the full list.
    struct Cull {
        bool on = false;
        const sccd_mesh* mesh = nullptr;
        int is_vf = 0;
        double ms = 0, tol = 0;
        CullSlabs slabs; // what the lists are good for: narrow launches that start at or below slabs.t_end
    } cull;
    // called ONCE, between the launch of the next sweep and the launch of its cull (bp_detect_partial): ccd() puts the event that
    // releases the OTHER pass's sweep there -- that sweep waits for this one, not for this one's cull
    std::function<void()> after_sweep;
    int sweeps_in_call = 0; // sweeps launched since the last bp_detect_partial(bp, 0 / 1) started (1: the first attempt stands)
    // bp_detect_partial(bp, 2) reads the first attempt's counters back through ANOTHER context's stream and mailbox, behind this
    // event (recorded on this object's stream behind the sweep and its cull): ccd() has put the pass's walk kernel into this
    // object's stream already, and a read-back queued behind that kernel would come at the very end of the step.  Used once.
    sccd_ctx* rb_ctx = nullptr;
    hipEvent_t rb_after = nullptr;
    // ... or finds the first attempt's counters READ ALREADY (ccd(): they came with the pass's verdict -- narrow_walk.inc
    // np_verdict_k -- at the end of the step): {SweepCounters, GridReadBack, ShardWindow} in host memory, consumed by the next
    // bp_detect_partial(bp, 2); no read-back of its own for the first attempt then
    const struct SweepFirstRead* pre_read = nullptr;
    DevBuf kept, kept_b; // int2[capacity]: the cull's list; kept_b: the list of the second half of time (cull.slabs.two), made on
                         // the device only when that half is asked for at all (run_walk) -- its length stays there
    int64_t n_kept = 0;
    bool one_class = false;                // a two-list build whose sweep runs list B's rows only (api.hip bp_build)
    bool speculative = false;              // la.m / lb.m are BOUNDS until bp_detect_partial has checked the guess
    bool spec_window = false;              // ... of a rank's cell window, dealt out on the device
    uint32_t spec_bound[2] = { 0, 0 };     // what the records were sized for
    uint32_t spec_sorted = 0, spec_cap = 0; // pairs sorted (padded); room of the entry buffers per list
};

// ------------------------------------------------------------------------------------------
// boxes.hip
void launch_pack_vertices(sccd_ctx* c, const double* dV0, const double* dV1, int nV, double* dV);
void launch_pack_edges(sccd_ctx* c, const int32_t* dE, int nE, int nV, int2* out, unsigned* bad);
void launch_pack_faces(sccd_ctx* c, const int32_t* dF, int nF, int nV, int4* out, unsigned* bad);
// st / part may be null (no statistics); otherwise *st must be zeroed and the return value is the
// number of block partials written
int launch_vertex_boxes(sccd_ctx* c, const double* dV, int nV, double inflation, sccd_aabb* out, GridStats* st = nullptr,
                        double* part = nullptr);
int launch_edge_boxes(sccd_ctx* c, const sccd_aabb* vb, const int2* E, int nE, sccd_aabb* out, GridStats* st = nullptr,
                      double* part = nullptr);
int launch_face_boxes(sccd_ctx* c, const sccd_aabb* vb, const int4* F, int nF, sccd_aabb* out, GridStats* st = nullptr,
                      double* part = nullptr);
struct GridStats;
struct GridParams;
// edge and face boxes in one launch (nE, nF > 0); *n_part_e / *n_part_f: block partials written per list
void launch_edge_face_boxes(sccd_ctx* c, const sccd_aabb* vb, const int2* E, int nE, sccd_aabb* out_e, GridStats* st_e, double* part_e,
                            int* n_part_e, const int4* F, int nF, sccd_aabb* out_f, GridStats* st_f, double* part_f, int* n_part_f);
int launch_box_stats(sccd_ctx* c, const sccd_aabb* raw, int n, GridStats* st, double* part);
void launch_grid_setup(sccd_ctx* c, const GridStats* st_a, const double* part_a, int n_part_a, const GridStats* st_b,
                       const double* part_b, int n_part_b, int n_total, int axis, double cell_factor, int shrink,
                       GridParams* g, uint32_t* cursors, bool reserve_tag = false, uint32_t* zero_hist = nullptr);
void launch_cell_hist(sccd_ctx* c, const sccd_boxes* A, const sccd_boxes* B /* or null */, const GridParams* g, int stride, uint32_t* hist);
int launch_elem_stats(sccd_ctx* c, const sccd_boxes* b, int stride, GridStats* st, double* part);
void launch_elem_stats_two(sccd_ctx* c, const sccd_boxes* e, const sccd_boxes* f, int stride, int* n_part_e, int* n_part_f);
// a rank's window of cells, decided on the device (boxes.hip shard_window_k)
struct ShardWindow {
    int cell_lo, cell_hi; // this rank's cells
    int n_cells, pad;
    unsigned long long total_est;  // entries of the whole grid, estimated from the sampled histogram
    unsigned long long window_est; // ... of this rank's window
};
void launch_shard_window(sccd_ctx* c, const uint32_t* hist, const GridParams* g, int stride, int rank, int parts, ShardWindow* out);
// d_win != nullptr: the window is read from device memory (cell_lo / cell_hi are ignored)
void launch_cell_fill_append(sccd_ctx* c, const sccd_boxes* b, const GridParams* g, int cell_lo, int cell_hi,
                             uint32_t* cursor, uint32_t capacity, uint32_t* key, uint32_t* idx, bool tagged = false,
                             uint32_t* place = nullptr, const ShardWindow* d_win = nullptr);
// both lists of a merged two-list build in one launch: cursors[0] / [1] count list A's / B's entries, cursors[2] places both
void launch_cell_fill_append_two(sccd_ctx* c, const sccd_boxes* A, const sccd_boxes* B, const GridParams* g,
                                 int cell_lo, int cell_hi, uint32_t* cursors, uint32_t capacity, uint32_t* key, uint32_t* idx,
                                 const ShardWindow* d_win = nullptr);
void launch_cell_count(sccd_ctx* c, const sccd_aabb* raw, int n, const GridParams* g, int cell_lo, int cell_hi,
                       uint32_t* counts);
void launch_cell_fill(sccd_ctx* c, const sccd_aabb* raw, int n, const GridParams* g, int cell_lo, int cell_hi,
                      const uint32_t* offsets, uint32_t* key, uint32_t* idx);
// the sorted records of one list from its sorted (key, box index) pairs.  mode 0: one list; 1 / 2: this list is the
// row list A / B of a two-list sweep and `other` are the sorted keys of the column list (tagged like the merged sort left
// them): every record also gets its first candidate column (sweep.hip: the three sweep classes)
// both lists of a two-list build in one launch (b_tagged: list B's keys carry the list tag of a merged sort)
void launch_entry_records_two(sccd_ctx* c, const sccd_aabb* raw_a, const uint32_t* key_a, const uint32_t* idx_a, int ma,
                              const sccd_aabb* raw_b, const uint32_t* key_b, const uint32_t* idx_b, int mb, bool b_tagged,
                              const GridParams* g, SortedList* out_a, SortedList* out_b, const uint32_t* d_tot = nullptr, int expect_bits = 0,
                              const uint32_t* d_extq = nullptr);
// own_tagged / other_tagged: this list's / the column list's keys carry the list tag of a merged sort
void launch_entry_records(sccd_ctx* c, const sccd_aabb* raw, const uint32_t* key, const uint32_t* idx, int m,
                          const GridParams* g, int mode, const uint32_t* other, int n_other, bool own_tagged,
                          bool other_tagged, SortedList* out, const uint32_t* d_tot = nullptr, int expect_bits = 0);

// scan.hip
void exclusive_scan_u32(sccd_ctx* c, const uint32_t* in, uint32_t* out, int n, uint32_t* d_total);
// variance of box centres per axis -> arg-max axis (sort_and_sweep.cpp:176-195); blocking
int pick_sort_axis(sccd_ctx* c, const sccd_aabb* raw, int n, const sccd_aabb* raw_b = nullptr, int n_b = 0);

// sort.hip: in-place LSD radix sort of (key, value) pairs by key, ascending, stable
// d_n_real (device, may be null): only the first *d_n_real of the n pairs are there, the rest sort as key 0xFFFFFFFF (behind them)
bool radix_sort_pairs_u32(sccd_ctx* c, uint32_t* keys, uint32_t* vals, int64_t n, int key_bits = 32, const uint32_t* d_n_real = nullptr);

// sweep.hip
enum SweepEmit { EMIT_ONE_LIST = 0, EMIT_ROWS_A = 1, EMIT_ROWS_B = 2 };
struct SweepCounters { // lives in device memory (ctx->scalars)
    unsigned long long n_pairs;    // pairs found (may exceed capacity: overflow -> rerun)
    unsigned long long candidates; // (host side: sum of cand_parts)
    unsigned int tile_ticket;      // (unused: tiles are dealt statically)
    unsigned int pad;
    unsigned long long cand_parts[32]; // candidate columns tested (key range on the sort axis), summed over the rows; spread to avoid one hot word
    unsigned long long diag[4];        // SCCD_SWEEP_DIAG=1: filter blocks, filter groups of 8 steps, confirm rounds, segments staged (summed over waves)
    unsigned long long n_kept;         // ccd(): pairs the projection cull behind this sweep kept (narrow_cull.inc); cleared with the rest
    unsigned long long n_kept_b;       // ... and the cull for the second half of the pass's time, if that ran (run_walk)
    unsigned long long pad2[23];       // 512 bytes: hipMemsetAsync clears an aligned size with ONE fill kernel (280 B took two)
};
static_assert(sizeof(SweepCounters) == 512, "SweepCounters: keep the size a multiple of 256 bytes");
struct SweepFirstRead { // what bp_detect_partial(bp, 2) reads back of a first attempt (sccd_broad_phase::pre_read)
    SweepCounters h;
    GridReadBack built;
    ShardWindow hwin;
};
// rows [row_begin, row_end) of `rows` against the columns of `cols` (rows == cols: one list)
void launch_sweep(sccd_ctx* c, const SortedList* rows, const SortedList* cols, const GridParams* gp, int row_begin,
                  int row_end, int emit, int2* out, int64_t capacity, SweepCounters* d_cnt, const uint32_t* d_m_rows = nullptr,
                  const uint32_t* d_m_cols = nullptr, int expect_bits = 0); // (device-side entry counts and the key width they were sorted by: sweep_band_k)
// both classes of a two-list sweep: rows [a_begin, a_end) of A against B and rows [b_begin, b_end) of B against A
void launch_sweep_two(sccd_ctx* c, const SortedList* A, const SortedList* B, const GridParams* gp, int a_begin, int a_end,
                      int b_begin, int b_end, int2* out, int64_t capacity, SweepCounters* d_cnt, const uint32_t* d_tot = nullptr, int expect_bits = 0);

// narrow.hip
struct NarrowParams {
    const double* V;
    const int2* E;
    const int4* F;
    const int2* pairs;
    long long n_pairs;
    int is_vf;
    int max_iter;
    double tol;
    double ms;
    int allow_zero_toi;
    int arith;
    // the running TOI of this launch lives in ANOTHER launch's counters (ccd(): the edge-edge kernel starts beside the
    // vertex-face kernel and shares its word, so that each prunes with what the other finds); nullptr: its own
    unsigned long long* toi_word = nullptr;
    // The SECOND launch of the two halves of time has a list of its own where the pass culls per slab of time (src != nullptr):
    // made between the two launches, on the device, from the pass's whole overlap list -- and only if the first launch accepted
    // nothing (np_cull_k's go word).  nullptr: the second launch walks the first one's list.
    struct SecondHalf {
        const int2* src = nullptr;                 // the pass's overlaps
        const unsigned long long* d_n_src = nullptr; // ... how many (device)
        long long capacity = 0;                    // ... at most (the buffers' size)
        int2* kept = nullptr;                      // the list to make
        unsigned long long* d_n_kept = nullptr;    // ... its length (device; 0 at launch)
        double t_lo = 0.5, t_hi = 1.0;
    } second;
};
struct NarrowCounters;
void narrow_counters_upload(sccd_ctx* c, NarrowCounters* d_cnt, double toi);
void narrow_seed_word(sccd_ctx* c, NarrowCounters* dst, const NarrowCounters* src); // dst's TOI = min(dst's, src's), on c->stream
struct NarrowCounters {
    // The three hot words sit on separate 128-byte lines: sharing one line, the ticket atomics
    // queued behind every wave's TOI polls and cost tens of microseconds each.
    unsigned long long toi_bits; // running minimum (non-negative double as u64): polled + atomicMin
    unsigned long long pad0[15];
    unsigned long long ticket;   // (unused: np_walk_k's chunk tickets are checks_part[].ticket)
    unsigned long long pad1[15];
    unsigned long long n_checks; // inclusion-function evaluations (one atomicAdd per wave)
    unsigned int overflow;
    unsigned int n_ovf; // queries np_walk_k handed to the level-synchronous path (entries of its overflow list)
    unsigned long long toi_level; // level-synchronous kernels with a check limit: the TOI as of the start of the level
    unsigned int n_arg;           // np_walk_k with a check limit: (query, time) records of the lanes that lowered the TOI (narrow.hip: the certificate)
    unsigned int second_go;       // "two halves of time" (narrow_walk.inc): 1 = the second launch has work (np_verdict_k)
    // ccd(): the running TOI word of the OTHER pass's launch (device address, or 0) -- a wave that lowers this launch's TOI by an accepted
    // domain publishes the time there too, and np_verdict_k looks there before it sends a second half to work (narrow_walk.inc, "the peer")
    unsigned long long peer_word;
    unsigned long long pad3[11];
    // occupancy diagnostics of np_walk_k (SCCD_NP_DIAG=1 prints them)
    unsigned long long wave_steps;   // check steps executed by waves
    unsigned long long lane_steps;   // live lanes summed over those steps
    unsigned long long refill_execs; // hand-overs of a staging buffer
    unsigned long long steals;       // sub-domains moved between lanes
    unsigned long long stamp[8];     // SCCD_NP_DIAG=2: shader cycles per loop section, summed over waves
    unsigned long long max_wave_steps, waves_run; // longest wave (the kernel's critical path), waves that got work
    unsigned long long pops_reg, checked_ahead; // pops of deferred halves; nodes checked ahead by idle lanes in the tail
    unsigned long long wave_hist[16];      // waves by number of check steps, buckets of 16 (last: >= 240)
    unsigned long long xcd_steps[8], xcd_waves[8]; // check steps and waves per XCD (HW_REG_XCC_ID)
    unsigned long long tail_steps, max_tail_steps, max_tail_cycles, max_total_cycles, sum_tail_cycles; // after the wave's stream ran dry
    // check counts, striped over eight 128-byte lines: every wave adds its count when it ends, and they
    // all end together -- 2048 atomics on one word were a 20 us tail on every launch
    // (the same lines hold np_walk_k's chunk tickets, one per sub-list of the query list: hot during the launch, when
    // the check counts are not touched)
    struct alignas(128) Stripe {
        unsigned long long n;
        unsigned long long ticket;
    } checks_part[8];
};
static_assert(sizeof(NarrowCounters) <= 2048, "NarrowCounters must fit its slot of the scalars block");
constexpr int SCCD_QUEUE_MIN_MAX_ITER = 4096; // smaller check limits are served by the level-synchronous kernel
// runs the narrow phase on the stream; *toi in/out lives in d_cnt->toi_bits
// what else a pass's verdict carries behind its counters (np_verdict_k): up to three runs of device words, copied to the given word
// offsets of the pinned verdict buffer (sccd_ctx::verdict) -- ccd() puts the pass's sweep counters, grid and cell window there
struct VerdictExtras {
    const unsigned* src[4];
    unsigned off_words[4], n_words[4];
    int n;
};
constexpr unsigned VERDICT_SWEEP_AT = 2560, VERDICT_BUILT_AT = 3072, VERDICT_WINDOW_AT = 3200; // byte offsets of ccd()'s extras
// ... and of what a launch with a check limit adds itself (run_walk): {the query that holds the earliest accept, its record count, its
// 24 coordinates} -- the inputs of the certificate (narrow.hip np_cert_k), so that the host's proof needs no read-back either
constexpr unsigned VERDICT_CERT_AT = 3232, VERDICT_CERT_WORDS = 2 + 48;
void narrow_phase_begin(sccd_ctx* c, const NarrowParams& p, NarrowCounters* d_cnt, const double* h_toi_inout,
                        double* d_per_query_toi, const unsigned long long* d_n = nullptr, long long capacity = 0,
                        const VerdictExtras* vx = nullptr);
const char* narrow_verdict_wait(sccd_ctx* c);
void narrow_phase_end(sccd_ctx* c, const NarrowParams& p, NarrowCounters* d_cnt, double* h_toi_inout,
                      double* d_per_query_toi);
void narrow_phase_run(sccd_ctx* c, const NarrowParams& p, NarrowCounters* d_cnt, double* h_toi_inout,
                      double* d_per_query_toi);
bool narrow_uses_walk_kernel(const sccd_ctx* c, const NarrowParams& p, bool per_query); // (else: level-synchronous kernels)
// the TOI a pass's counters are STARTED from: the caller's, or 0.5 where the pass will be served by the walk kernel's two launches
// over the halves of time (narrow_walk.inc; plain launches of the double build from a TOI above 0.5, SCCD_OPT_TWO_HALVES)
double narrow_start_toi(const sccd_ctx* c, const NarrowParams& p, double toi, bool per_query);
// the projection cull (narrow_cull.inc): pairs[0 .. min(*d_n_pairs, capacity)) -> the pairs that may have an impact, compacted
// into d_kept[0 .. *d_n_kept) (any order; *d_n_kept must be 0); on c->stream.  Only p's mesh pointers, pairs, is_vf, ms, tol are used.
void narrow_cull_launch(sccd_ctx* c, const NarrowParams& p, const unsigned long long* d_n_pairs, long long capacity, int2* d_kept,
                        unsigned long long* d_n_kept, double t_lo = 0.0, double t_hi = 1.0, const unsigned* go = nullptr);
// the slabs for a pass whose narrow launches start from `toi` at most (narrow_start_toi decides about the two halves)
CullSlabs narrow_cull_slabs(const sccd_ctx* c, const NarrowParams& p, double toi);
// ti_census.cpp (host): ONE query bisected alone in the reference's level order with the check limit
double ti_census_level_order(const double v[8][3], int is_vf, int arith, double ms, double tol, int max_iter, int allow_zero_toi,
                             double toi_init, long long max_live, bool* gave_up);
int narrow_selftest_waves_per_block(); // (np_walk_k's block shape: NW_WAVES)
void narrow_selftest_lds_gather(sccd_ctx* c, const double* d_V, const int* d_perm, int n_waves, int n_active, double* d_out);
