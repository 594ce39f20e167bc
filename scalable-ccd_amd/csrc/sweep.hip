// sweep.hip -- the sweep (STQ for wave64) that emits overlap pairs from the sorted records.
//
// Replaces sweep_and_tiniest_queue<> / sweep_and_prune<> with Queue, add_overlap,
// RawDeviceBuffer::push (src/scalable_ccd/cuda/broad_phase/sweep.cu:48-182, queue.cuh:5-49,
// collision.cuh:11-54, utils/device_buffer.cuh:56-61) and the CPU twin batched_sweep
// (src/scalable_ccd/broad_phase/sort_and_sweep.cpp:77-125).
//
// Set semantics (identical to the reference): all (a,b) whose boxes intersect INCLUSIVELY on
// x, y and z, that share no vertex id, and (two lists) come from different lists.
//
// Sweep classes (the candidate columns of a row are a run of the column list that starts at the row's `start` and ends
// with the last column key <= the row's max-key; the exact tests decide):
//   one list : row i against columns j > i with K(min_j) <= K(max_i)                  start = i + 1
//   rows A   : columns B with K(min_a) <= K(min_b) <= K(max_a)                        start = lower_bound(keys B, K(min_a))
//   rows B   : columns A with K(min_b) <  K(min_a) <= K(max_b)                        start = upper_bound(keys A, K(min_b))
// The two two-list classes partition the intersecting cross pairs (exactly one of K(min_a) <= K(min_b),
// K(min_b) < K(min_a) holds), so no pair is emitted twice and none is lost.
//
// Structure of sweep_band_k, per wave of 64 lanes, over tiles of 64 sorted rows (lane = row):
//   STAGE    the rows' candidate columns lie in a band: row r starts at about start_0 + r.  The wave keeps a WINDOW of up
//            to 128 columns in LDS (circular, staged in 32-column segments with consecutive 16-byte loads: every column
//            record leaves HBM once per tile that needs it and is never gathered).
//   FILTER   SKEWED: at step c lane r tests ITS OWN column start_r + c -- the exact inclusive test on the two minor
//            axes and the key test on the sort axis, from LDS.  32 steps per block, a 32-bit hit mask per lane.  (Walking
//            the union of the 64 rows' runs with wave-uniform columns, as round 2 did, tests 64 + len columns per
//            row for a run of len.)
//   QUEUE    hit masks are expanded into a per-wave LDS queue of (row lane, column slot) candidates
//            (one wave prefix-sum per block) -- the "tiniest queue" of STQ, 64 lanes wide.
//   CONFIRM  whenever >= 64 candidates are queued, every lane confirms one: the column's sort-axis interval, ids and
//            lowest cells from the LDS window, the row's from the owning lane's registers (ds_bpermute): the exact
//            sort-axis test, the 3x3 vertex-id test (mesh-built lists: in the filter already), the owner-cell test.
//            No global memory access.
//   EMIT     survivors go to a per-wave LDS staging buffer; one global atomic per ~1000 pairs
//            reserves space, then the pairs are written coalesced (the reference: two global
//            atomics per pair, collision.cuh:45-54).
#include "internal.hpp"
#include "grid.hpp"

#include <algorithm>

namespace {

constexpr int SW_THREADS = 256;
constexpr int SW_WAVES = SW_THREADS / 64;
constexpr int SW_WIN = 128;   // column slots of a wave's window (slot = column & 127)
constexpr int SW_SEG = 32;    // columns per staging segment = steps per filter block
constexpr int SW_MIR = SW_SEG; // the filter's arrays repeat their first segment behind the last: a run of 32 steps never wraps
constexpr int SW_QCAP = 256;  // candidate queue entries per wave
constexpr int SW_OCAP = 1024; // staged output pairs per wave (smaller: the reservations on the one pair counter become the bottleneck -- 512: +40 %)

struct SweepLds { // per wave: 19,584 bytes; four waves per block, two blocks per CU
    double2 a[SW_WIN + SW_MIR]; // {min, max} on minor axis a
    double2 b[SW_WIN + SW_MIR];
    uint4 x[SW_WIN];          // double2 {min, max} on the sort axis
    uint4 id[SW_WIN];         // vertex ids, element id
    uint32_t key[SW_WIN + SW_MIR];
    uint32_t low[SW_WIN];       // lowest cells
    uint32_t q[SW_QCAP];        // candidates: row lane << 8 | column slot
    int2 o[SW_OCAP];
};

// output convention: sweep.cu:152-163 / sort_and_sweep.cpp:106-118
__device__ __forceinline__ int2 make_pair_out(int emit, int row_eid, int col_eid)
{
    if (emit == EMIT_ONE_LIST) return make_int2(min(row_eid, col_eid), max(row_eid, col_eid));
    if (emit == EMIT_ROWS_A) return make_int2(row_eid, col_eid);
    return make_int2(col_eid, row_eid);
}

// per-wave emission state (LDS staging + one global atomic per flush)
struct Emitter {
    int2* stage;     // LDS, SW_OCAP entries
    int ocount;      // wave-uniform
    int thr;         // wave-uniform: flush above this many staged pairs (the FIRST threshold differs from wave to wave: below)
    int2* out;       // global
    long long capacity;
    unsigned long long* n_pairs;

    __device__ __forceinline__ void flush()
    {
        if (ocount == 0) return;
        unsigned long long base = 0;
        if (lane_id() == 0) base = atomicAdd(n_pairs, (unsigned long long)ocount);
        base = __shfl(base, 0, 64);
        wave_lds_fence();
        for (int k = lane_id(); k < ocount; k += 64) {
            const unsigned long long dst = base + (unsigned long long)k;
            if ((long long)dst < capacity) out[dst] = stage[k];
        }
        wave_lds_fence();
        ocount = 0;
    }
    // The last flush of a kernel, by ALL waves of the block together: one atomic per block.  (Every
    // wave ends at about the same time, and the cursor is one hot word: ~90 atomics/us chip-wide --
    // a flush per wave made the tail of each launch a queue of 4096 atomics.)  The block's count of candidate tests
    // rides along (one more atomic per block, on one of 32 words).
    // `scratch`: 3 * SW_WAVES u64 of LDS no wave uses any more
    __device__ __forceinline__ void flush_block(unsigned long long* scratch, unsigned long long tests, unsigned long long* cand_word)
    {
        unsigned long long* blk_cnt = scratch;
        unsigned long long* blk_base = scratch + SW_WAVES;
        unsigned long long* blk_tests = scratch + 2 * SW_WAVES;
        const int w = (int)(threadIdx.x >> 6);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tests += __shfl_xor(tests, o, 64);
        __syncthreads(); // every wave is done with the scratch area
        if (lane_id() == 0) {
            blk_cnt[w] = (unsigned long long)ocount;
            blk_tests[w] = tests;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long sum = 0, tsum = 0;
            for (int k = 0; k < SW_WAVES; k++) {
                blk_base[k] = sum;
                sum += blk_cnt[k];
                tsum += blk_tests[k];
            }
            const unsigned long long b0 = sum ? atomicAdd(n_pairs, sum) : 0ull;
            for (int k = 0; k < SW_WAVES; k++) blk_base[k] += b0;
            if (tsum) atomicAdd(cand_word, tsum);
        }
        __syncthreads();
        const unsigned long long base = blk_base[w];
        for (int k = lane_id(); k < ocount; k += 64) {
            const unsigned long long dst = base + (unsigned long long)k;
            if ((long long)dst < capacity) out[dst] = stage[k];
        }
        ocount = 0;
    }
    __device__ __forceinline__ void push(bool ok, int2 pr) { push_mask(__ballot(ok), pr); }
    // mask: the lanes that emit (wave-uniform)
    __device__ __forceinline__ void push_mask(unsigned long long mask, int2 pr)
    {
        if (mask == 0) return;
        if ((mask >> lane_id()) & 1ull) stage[ocount + mbcnt64(mask)] = pr;
        ocount += popc64(mask);
        if (ocount > thr) {
            flush();
            thr = SW_OCAP - 64;
        }
    }
};

// one row's record in registers
struct RowRegs {
    double2 x, a, b;
    int4 id;
    uint4 aux; // key, max-key, lowest cells, first column
};
__device__ __forceinline__ double2 as_double2(const uint4& u)
{
    return make_double2(__hiloint2double((int)u.y, (int)u.x), __hiloint2double((int)u.w, (int)u.z));
}
__device__ __forceinline__ void load_row(RowRegs& r, const SweepRecs& R, int row)
{
    const uint4* p = R.base + (unsigned)row;
    const uint4 a = p[REC_A * (size_t)R.pstride], b = p[REC_B * (size_t)R.pstride], x = p[REC_X * (size_t)R.pstride];
    const uint4 id = p[REC_ID * (size_t)R.pstride];
    r.aux = p[REC_AUX * (size_t)R.pstride];
    r.a = as_double2(a);
    r.b = as_double2(b);
    r.x = as_double2(x);
    r.id = make_int4((int)id.x, (int)id.y, (int)id.z, (int)id.w);
}

// 32 columns [c0, c0 + 32) (c0 a multiple of 32) into their window slots: 160 pieces of 16 bytes.  Lane l moves piece
// (l >> 5) of column l & 31 (minor axis a / b), piece 2 + (l >> 5) (sort axis / ids) and, the lower half of the wave, the
// aux piece: consecutive lanes, consecutive 16 bytes.  Where each lane reads and writes is fixed for the kernel (Stager),
// so a segment costs three loads, two 16-byte LDS stores, two 4-byte ones and no address arithmetic to speak of.
// Columns past the end of the list read the arrays' padding (never tested).  In two halves, so that a segment can be
// asked for ahead and land in registers meanwhile.
struct SegRegs {
    uint4 v0, v1, v2;
};
struct Stager {
    const uint4* base; // the column list's pieces
    uint32_t g0, g1, g2; // per lane: offsets of its three pieces (units of 16 bytes) for column 0
    uint4 *d0, *d1;      // per lane: LDS slots of its first two pieces for window slot 0
    uint32_t *dk, *dl;   // ... of the key and the lowest cells
    bool low_half;
    __device__ __forceinline__ Stager(SweepLds& L, const SweepRecs& C)
    {
        const int lane = lane_id();
        const uint32_t s = (uint32_t)(lane & 31), h = (uint32_t)(lane >> 5);
        base = C.base;
        g0 = (REC_A + h) * C.pstride + s;
        g1 = (REC_X + h) * C.pstride + s;
        g2 = REC_AUX * C.pstride + s;
        d0 = reinterpret_cast<uint4*>(h ? L.b : L.a) + s;
        d1 = (h ? L.id : L.x) + s;
        dk = L.key + s;
        dl = L.low + s;
        low_half = h == 0;
    }
    __device__ __forceinline__ SegRegs load(unsigned c0) const
    {
        const uint4* b = base + c0; // wave-uniform
        SegRegs g;
        g.v0 = b[g0];
        g.v1 = b[g1];
        g.v2 = make_uint4(0u, 0u, 0u, 0u);
        if (low_half) g.v2 = b[g2];
        return g;
    }
    __device__ __forceinline__ void write(const SegRegs& g, unsigned c0) const
    {
        const unsigned slot0 = c0 & (unsigned)(SW_WIN - 1); // wave-uniform
        d0[slot0] = g.v0;
        d1[slot0] = g.v1;
        if (low_half) {
            dk[slot0] = g.v2.x;
            dl[slot0] = g.v2.z;
        }
        if (slot0 == 0) { // the filter's arrays repeat their first segment behind the last
            d0[SW_WIN] = g.v0;
            if (low_half) dk[SW_WIN] = g.v2.x;
        }
    }
};

// CONFIRM one hit of the lane's own row: AABB::intersects on the sort axis (aabb.cuh:68-73; the two minor axes passed the
// filter) && !share_a_vertex (collision.cuh:17-21; IDS == 0: the filter saw to that already) && this cell owns the pair
// (grid.hpp: a box listed in several cells meets the same partner in each of them; the pair is reported only from the
// cell (max of the two boxes' lowest cells per minor axis), which both boxes share iff they overlap on that axis: the
// packed maximum of the two lowest-cell words equals the row's own cell coordinates, packed alike).  The column comes
// from the LDS window, the row is in the lane's registers.  The comparisons narrow EXEC one after the other (v_cmpx, as
// in the filter); what is left of it is the mask of confirmed pairs.  Returns that mask (wave-uniform).
template <int IDS>
__device__ __forceinline__ unsigned long long confirm_mask(unsigned has, const double2& rx, const int4& rv, unsigned my_low,
                                                           unsigned my_cellpack, const double2& cx, const int4& cv, unsigned clow)
{
    unsigned long long save, ok;
    unsigned tmp;
    if (IDS)
        asm volatile("s_mov_b64 %[save], exec\n\t"
                     "v_cmpx_ne_u32_e32 vcc, 0, %[has]\n\t"
                     "v_pk_max_u16 %[tmp], %[mylow], %[clow]\n\t"
                     "v_cmpx_eq_u32_e32 vcc, %[tmp], %[mycell]\n\t"
                     "v_cmpx_ge_f64_e32 vcc, %[rxhi], %[cxlo]\n\t"
                     "v_cmpx_le_f64_e32 vcc, %[rxlo], %[cxhi]\n\t"
                     "v_cmpx_ne_u32_e32 vcc, %[r0], %[c0]\n\t"
                     "v_cmpx_ne_u32_e32 vcc, %[r0], %[c1]\n\t"
                     "v_cmpx_ne_u32_e32 vcc, %[r0], %[c2]\n\t"
                     "v_cmpx_ne_u32_e32 vcc, %[r1], %[c0]\n\t"
                     "v_cmpx_ne_u32_e32 vcc, %[r1], %[c1]\n\t"
                     "v_cmpx_ne_u32_e32 vcc, %[r1], %[c2]\n\t"
                     "v_cmpx_ne_u32_e32 vcc, %[r2], %[c0]\n\t"
                     "v_cmpx_ne_u32_e32 vcc, %[r2], %[c1]\n\t"
                     "v_cmpx_ne_u32_e32 vcc, %[r2], %[c2]\n\t"
                     "s_mov_b64 %[ok], exec\n\t"
                     "s_mov_b64 exec, %[save]"
                     : [save] "=&s"(save), [ok] "=&s"(ok), [tmp] "=&v"(tmp)
                     : [has] "v"(has), [mylow] "v"(my_low), [clow] "v"(clow), [mycell] "v"(my_cellpack), [rxlo] "v"(rx.x),
                       [rxhi] "v"(rx.y), [cxlo] "v"(cx.x), [cxhi] "v"(cx.y), [r0] "v"(rv.x), [r1] "v"(rv.y), [r2] "v"(rv.z),
                       [c0] "v"(cv.x), [c1] "v"(cv.y), [c2] "v"(cv.z)
                     : "vcc");
    else
        asm volatile("s_mov_b64 %[save], exec\n\t"
                     "v_cmpx_ne_u32_e32 vcc, 0, %[has]\n\t"
                     "v_pk_max_u16 %[tmp], %[mylow], %[clow]\n\t"
                     "v_cmpx_eq_u32_e32 vcc, %[tmp], %[mycell]\n\t"
                     "v_cmpx_ge_f64_e32 vcc, %[rxhi], %[cxlo]\n\t"
                     "v_cmpx_le_f64_e32 vcc, %[rxlo], %[cxhi]\n\t"
                     "s_mov_b64 %[ok], exec\n\t"
                     "s_mov_b64 exec, %[save]"
                     : [save] "=&s"(save), [ok] "=&s"(ok), [tmp] "=&v"(tmp)
                     : [has] "v"(has), [mylow] "v"(my_low), [clow] "v"(clow), [mycell] "v"(my_cellpack), [rxlo] "v"(rx.x),
                       [rxhi] "v"(rx.y), [cxlo] "v"(cx.x), [cxhi] "v"(cx.y)
                     : "vcc");
    return ok;
}

// One filter step of a lane: is column (key k, minor-axis intervals ca, cb) inside the row's key range (-> bit BIT of km) and
// does it overlap the row on both minor axes, inclusively (-> bit BIT of m)?  The five conditions narrow the EXEC mask one
// after the other (v_cmpx) and the two ORs run under it: 7 vector and 2 scalar instructions.  (Written as C the same step
// is 5 v_cmp + 4 s_and_b64 + 2 v_cndmask + 2 v_or and a v_mov per bit: 9-10 vector and 5 scalar instructions -- and this
// loop is what the sweep's time goes into.)  EXEC is restored before the statement ends; NaN bounds compare false like in C.
// KIND: what is known about the vertex ids of the two lists (boxes built HERE from a mesh: aabb.cpp:57-58,107-109,128-130 --
// vertex {i, -i-1, -i-1}, edge {e0, e1, -e0-1}, face {f0, f1, f2}, all indices >= 0).  A mesh's sweep lists are full of
// neighbours that overlap and share a vertex (an edge of a cloth has ten of them against three real partners): with the
// ids known, share_a_vertex (collision.cuh:17-21) is 4 comparisons (edge - edge) or 3 (vertex - face) instead of 9 --
// cheap enough for the filter, and the confirm loop (as many rounds as the busiest lane has hits) loses 60 % of its hits.
//   0  unknown (uploaded boxes): the nine comparisons stay in the confirm stage
//   1  edges x edges: {a0, a1} x {b0, b1}        (a negative filler equals nothing but the filler of the same a0)
//   2  row vertex, column face: v x {f0, f1, f2}
//   3  row face, column vertex: {f0, f1, f2} x v
template <unsigned BIT, int KIND>
__device__ __forceinline__ void filter_step(unsigned& m, unsigned& km, uint32_t k, uint32_t kmax, const double2& ca,
                                            const double2& cb, const double2& ra, const double2& rb, const int4& rv,
                                            const int4& cv)
{
    const double calo = ca.x, cahi = ca.y, cblo = cb.x, cbhi = cb.y;
    unsigned long long save;
#define SCCD_FILTER_HEAD                                   \
    "s_mov_b64 %[save], exec\n\t"                          \
    "v_cmpx_le_u32_e32 vcc, %[k], %[kmax]\n\t"             \
    "v_or_b32_e32 %[km], %[bit], %[km]\n\t"                \
    "v_cmpx_le_f64_e32 vcc, %[calo], %[rahi]\n\t"          \
    "v_cmpx_le_f64_e32 vcc, %[ralo], %[cahi]\n\t"          \
    "v_cmpx_le_f64_e32 vcc, %[cblo], %[rbhi]\n\t"          \
    "v_cmpx_le_f64_e32 vcc, %[rblo], %[cbhi]\n\t"
#define SCCD_FILTER_TAIL                                   \
    "v_or_b32_e32 %[m], %[bit], %[m]\n\t"                  \
    "s_mov_b64 exec, %[save]"
#define SCCD_FILTER_GEO_OPS                                                                                              \
    [k] "v"(k), [kmax] "v"(kmax), [calo] "v"(calo), [cahi] "v"(cahi), [cblo] "v"(cblo), [cbhi] "v"(cbhi), [ralo] "v"(ra.x), \
        [rahi] "v"(ra.y), [rblo] "v"(rb.x), [rbhi] "v"(rb.y), [bit] "i"(BIT)
    if (KIND == 1)
        asm volatile(SCCD_FILTER_HEAD "v_cmpx_ne_u32_e32 vcc, %[r0], %[c0]\n\t"
                                      "v_cmpx_ne_u32_e32 vcc, %[r0], %[c1]\n\t"
                                      "v_cmpx_ne_u32_e32 vcc, %[r1], %[c0]\n\t"
                                      "v_cmpx_ne_u32_e32 vcc, %[r1], %[c1]\n\t" SCCD_FILTER_TAIL
                     : [m] "+v"(m), [km] "+v"(km), [save] "=&s"(save)
                     : SCCD_FILTER_GEO_OPS, [r0] "v"(rv.x), [r1] "v"(rv.y), [c0] "v"(cv.x), [c1] "v"(cv.y)
                     : "vcc");
    else if (KIND == 2)
        asm volatile(SCCD_FILTER_HEAD "v_cmpx_ne_u32_e32 vcc, %[r0], %[c0]\n\t"
                                      "v_cmpx_ne_u32_e32 vcc, %[r0], %[c1]\n\t"
                                      "v_cmpx_ne_u32_e32 vcc, %[r0], %[c2]\n\t" SCCD_FILTER_TAIL
                     : [m] "+v"(m), [km] "+v"(km), [save] "=&s"(save)
                     : SCCD_FILTER_GEO_OPS, [r0] "v"(rv.x), [c0] "v"(cv.x), [c1] "v"(cv.y), [c2] "v"(cv.z)
                     : "vcc");
    else if (KIND == 3)
        asm volatile(SCCD_FILTER_HEAD "v_cmpx_ne_u32_e32 vcc, %[r0], %[c0]\n\t"
                                      "v_cmpx_ne_u32_e32 vcc, %[r1], %[c0]\n\t"
                                      "v_cmpx_ne_u32_e32 vcc, %[r2], %[c0]\n\t" SCCD_FILTER_TAIL
                     : [m] "+v"(m), [km] "+v"(km), [save] "=&s"(save)
                     : SCCD_FILTER_GEO_OPS, [r0] "v"(rv.x), [r1] "v"(rv.y), [r2] "v"(rv.z), [c0] "v"(cv.x)
                     : "vcc");
    else
        asm volatile(SCCD_FILTER_HEAD SCCD_FILTER_TAIL
                     : [m] "+v"(m), [km] "+v"(km), [save] "=&s"(save)
                     : SCCD_FILTER_GEO_OPS
                     : "vcc");
#undef SCCD_FILTER_HEAD
#undef SCCD_FILTER_TAIL
#undef SCCD_FILTER_GEO_OPS
}

// ONE: rows and columns are the same list and a row's first column is the row behind it
// chunk: consecutive tiles a wave sweeps in a row (inside a chunk the window just moves on).
// block / n_blocks: this block's place among the blocks that sweep THIS class (a launch can sweep two classes: below)
template <bool ONE, int KIND>
__device__ __forceinline__ void sweep_band_body(SweepLds* lds_s, unsigned long long* blk_scratch, const SweepRecs& R, int row_begin,
                                                int row_end, const SweepRecs& C, int n_cols, const GridParams* __restrict__ gp,
                                                int emit, int chunk, int2* __restrict__ out, long long capacity,
                                                SweepCounters* __restrict__ cnt, int diag, unsigned block, unsigned n_blocks)
{
    unsigned d_blocks = 0, d_groups = 0, d_rounds = 0, d_segs = 0; // wave-uniform (SCCD_SWEEP_DIAG)
    const int lane = lane_id(), w = threadIdx.x >> 6;
    SweepLds& L = lds_s[w];
    const int xb = gp->xb, Sb = max(gp->Sb, 1);
    // Tiles are dealt statically and cost about the same, so the waves of a launch fill their staging areas in step and would all
    // reserve space at the same moments: ~2,000 returning atomics on ONE word, served one after the other (~10 ns each).  The first
    // flush of a wave therefore comes after a share of the staging area that differs from wave to wave (1/15 ... 15/15): from then
    // on the waves' reservations are spread over the filling period (1M boxes: 0.2375 -> 0.2326 ms; with no reservation at all,
    // waves writing to private slices, 0.229: the emit's atomic is 2 % of the sweep -- profiles/r05_ab/sweep_experiments.txt).
    const int first_thr = 64 * (1 + (int)(((block * SW_WAVES + (unsigned)w) * 5u) % (unsigned)(SW_OCAP / 64 - 1)));
    Emitter em { L.o, 0, min(first_thr, SW_OCAP - 64), out, capacity, &cnt->n_pairs };
    const Stager st(L, C);
    int qcount = 0;               // queued candidates (wave-uniform)
    unsigned q_lo = 0;            // the window's first column when the oldest queued candidate was queued
    unsigned long long tests = 0; // candidate columns of this lane's rows (key range on the sort axis)
    const unsigned n_cols_up = ((unsigned)n_cols + (unsigned)(SW_SEG - 1)) & ~(unsigned)(SW_SEG - 1);
    const int num_tiles = (row_end - row_begin + 63) / 64;
    // Chunks of tiles are dealt statically (a ticket per tile made the sweep ticket-bound: one hot word serves ~90
    // atomics/us chip-wide), XCD-AWARE: blocks go to the eight XCDs in turn, and XCD x sweeps the x-th eighth of the
    // tiles with its blocks' waves on neighbouring chunks -- their column windows meet in that XCD's L2.  (Nothing
    // depends on which XCD a block lands on.)
    const int xcd = (int)(block & 7u), bi = (int)(block >> 3), nbx = (int)(n_blocks >> 3);
    const int tpx = (num_tiles + 7) >> 3;
    const int t_lo = xcd * tpx, t_hi = min(num_tiles, (xcd + 1) * tpx);
    const int ch_stride = nbx * SW_WAVES * chunk; // tiles between a wave's chunks
    int ch_first = t_lo + (bi * SW_WAVES + w) * chunk; // first tile of this wave's current chunk
    int tile = ch_first;
    // the rows of the NEXT tile are requested before the current one is processed
    RowRegs nx;
    nx.x = nx.a = nx.b = make_double2(0.0, 0.0);
    nx.id = make_int4(0, 0, 0, 0);
    nx.aux = make_uint4(0u, 0u, 0u, 0u);
    {
        const int row0 = row_begin + tile * 64 + lane;
        if (tile < t_hi && row0 < row_end) load_row(nx, R, row0);
    }
    // CONFIRM + EMIT a batch of queued candidates: the column from the LDS window, the row from the registers of the lane
    // that owns it (ds_bpermute).  EVERY lane takes part in the moves; lanes >= n carry no candidate.
    auto confirm_batch = [&](int n, const RowRegs& me, unsigned my_cellpack) {
        wave_lds_fence();
        const unsigned has = lane < n ? 1u : 0u;
        const uint32_t e = has ? L.q[qcount - n + lane] : 0u;
        qcount -= n;
        const int r = (int)((e >> 8) & 63u);
        const unsigned slot = e & (unsigned)(SW_WIN - 1);
        const uint4 cxu = L.x[slot], cvu = L.id[slot];
        const uint32_t clow = L.low[slot];
        const double2 cx = as_double2(cxu);
        const int4 cv = make_int4((int)cvu.x, (int)cvu.y, (int)cvu.z, (int)cvu.w);
        double2 rx;
        rx.x = __shfl(me.x.x, r, 64);
        rx.y = __shfl(me.x.y, r, 64);
        int4 rv = make_int4(0, 0, 0, 0);
        if (KIND == 0) {
            rv.x = __shfl(me.id.x, r, 64);
            rv.y = __shfl(me.id.y, r, 64);
            rv.z = __shfl(me.id.z, r, 64);
        }
        const int reid = __shfl(me.id.w, r, 64);
        const unsigned rlow = (unsigned)__shfl((int)me.aux.z, r, 64);
        const unsigned rcell = (unsigned)__shfl((int)my_cellpack, r, 64);
        const unsigned long long ok = confirm_mask<KIND == 0>(has, rx, rv, rlow, rcell, cx, cv, clow);
        em.push_mask(ok, make_pair_out(emit, reid, cv.w));
    };
    auto drain = [&](const RowRegs& me, unsigned my_cellpack) {
        while (qcount > 0) confirm_batch(min(qcount, 64), me, my_cellpack);
    };
    unsigned w_lo = 0, w_hi = 0; // columns [w_lo, w_hi) are staged (w_lo a multiple of 32, w_hi - w_lo <= 128)
    bool have = false;           // (the window survives from tile to tile inside a chunk)
    int ahead = 0;               // segments [ahead_c0, + 32 * ahead) were asked for ahead and sit in `seg`
    unsigned ahead_c0 = 0;
    SegRegs seg0, seg1, seg2, seg3;
    seg0.v0 = seg0.v1 = seg0.v2 = make_uint4(0u, 0u, 0u, 0u);
    seg1 = seg2 = seg3 = seg0;
    while (tile < t_hi) {
        const int row = row_begin + tile * 64 + lane;
        const bool valid = row < row_end;
        const RowRegs me = nx;
        // the tile after this one: the next of the chunk, or the first of this wave's next chunk
        int tile_next = tile + 1;
        bool next_follows = true;
        if (tile_next >= ch_first + chunk || tile_next >= t_hi) {
            ch_first += ch_stride;
            tile_next = ch_first;
            next_follows = false;
        }
        {
            const int row_next = row_begin + tile_next * 64 + lane;
            if (tile_next < t_hi && row_next < row_end) load_row(nx, R, row_next);
        }
        unsigned j = ONE ? (unsigned)row + 1u : me.aux.w; // next column of this lane's row
        bool live = valid && j < (unsigned)n_cols;
        const uint32_t kmax = me.aux.y;
        // the row's own cell, packed like the lowest-cell words (a | b << 16); a grid of one cell: all zero
        const unsigned my_cell = (unsigned)((unsigned long long)me.aux.x >> xb);
        const unsigned my_cellpack = (my_cell / (unsigned)Sb) | ((my_cell % (unsigned)Sb) << 16);
        for (;;) {
            const unsigned base = wave_min_u32_dpp(live ? j : 0xFFFFFFFFu);
            if (base == 0xFFFFFFFFu) break;
            const unsigned nlo = base & ~(unsigned)(SW_SEG - 1);
            const unsigned need = min(n_cols_up, nlo + (unsigned)SW_WIN); // the whole window (a lane can be 96 columns ahead of the first)
            unsigned from = (have && w_hi > nlo && w_lo <= nlo) ? w_hi : nlo;
            if (from < need) {
                // a segment [c0, c0 + 32) takes the slots of columns [c0 - 128, c0 - 96): candidates queued since the window
                // started at q_lo may still name those
                if (qcount > 0 && need > q_lo + (unsigned)SW_WIN) drain(me, my_cellpack);
                // (asked for ahead: they have landed, or are about to)
                if (ahead >= 1 && ahead_c0 == from && from < need) {
                    st.write(seg0, from);
                    from += SW_SEG;
                    if (ahead >= 2 && from < need) {
                        st.write(seg1, from);
                        from += SW_SEG;
                        if (ahead >= 3 && from < need) {
                            st.write(seg2, from);
                            from += SW_SEG;
                            if (ahead >= 4 && from < need) {
                                st.write(seg3, from);
                                from += SW_SEG;
                            }
                        }
                    }
                }
                for (; from < need; from += SW_SEG) {
                    seg0 = st.load(from);
                    st.write(seg0, from);
                    ++d_segs;
                }
                wave_lds_fence();
            }
            ahead = 0;
            w_lo = nlo;
            w_hi = from;
            have = true;

            // ---- FILTER: lane r tests its own columns j, j + 1, ... (at most 32, inside the window)
            const unsigned lim = min(w_hi, (unsigned)n_cols);
            int avail = 0;
            if (live && j < lim) avail = (int)min(lim - j, (unsigned)SW_SEG);
            const unsigned s0 = j & (unsigned)(SW_WIN - 1);
            unsigned m = 0, km = 0;
            ++d_blocks;
            // (eight steps at a time: their 72 dwords of LDS reads are in flight together; the whole run unrolled asks for
            // more registers than a lane has.  A block ends with the longest lane's run.)
            bool inside = true; // the lane's last tested column was still inside its key range (keys ascend: once out, out)
#pragma unroll 1
            for (int c8 = 0; c8 < SW_SEG; c8 += 8) {
                if (__ballot(inside && avail > c8) == 0) break; // the block ends with the longest run
                ++d_groups;
                const unsigned at = s0 + (unsigned)c8;
                uint32_t k[8];
                double2 ca[8], cb[8];
                int4 cv[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    k[i] = L.key[at + i];
                    ca[i] = L.a[at + i];
                    cb[i] = L.b[at + i];
                    cv[i] = make_int4(0, 0, 0, 0);
                    const unsigned slot = (at + (unsigned)i) & (unsigned)(SW_WIN - 1); // (the ids are not mirrored)
                    if (KIND == 1) {
                        const uint2 t = *reinterpret_cast<const uint2*>(&L.id[slot]);
                        cv[i].x = (int)t.x;
                        cv[i].y = (int)t.y;
                    } else if (KIND == 2) {
                        const uint4 t = L.id[slot];
                        cv[i] = make_int4((int)t.x, (int)t.y, (int)t.z, 0);
                    } else if (KIND == 3) {
                        cv[i].x = (int)L.id[slot].x;
                    }
                }
                unsigned m8 = 0, km8 = 0;
                filter_step<1u, KIND>(m8, km8, k[0], kmax, ca[0], cb[0], me.a, me.b, me.id, cv[0]);
                filter_step<2u, KIND>(m8, km8, k[1], kmax, ca[1], cb[1], me.a, me.b, me.id, cv[1]);
                filter_step<4u, KIND>(m8, km8, k[2], kmax, ca[2], cb[2], me.a, me.b, me.id, cv[2]);
                filter_step<8u, KIND>(m8, km8, k[3], kmax, ca[3], cb[3], me.a, me.b, me.id, cv[3]);
                filter_step<16u, KIND>(m8, km8, k[4], kmax, ca[4], cb[4], me.a, me.b, me.id, cv[4]);
                filter_step<32u, KIND>(m8, km8, k[5], kmax, ca[5], cb[5], me.a, me.b, me.id, cv[5]);
                filter_step<64u, KIND>(m8, km8, k[6], kmax, ca[6], cb[6], me.a, me.b, me.id, cv[6]);
                filter_step<128u, KIND>(m8, km8, k[7], kmax, ca[7], cb[7], me.a, me.b, me.id, cv[7]);
                m |= m8 << c8;
                km |= km8 << c8;
                inside = (km8 & 0x80u) != 0;
            }
            const unsigned amask = avail >= 32 ? 0xFFFFFFFFu : ((1u << avail) - 1u);
            m &= amask;
            km &= amask;
            bool goes_on = false; // the lane's last column of this block is still inside its key range (keys ascend)
            if (avail > 0) {
                goes_on = (km >> (avail - 1)) & 1u;
                tests += (unsigned)__popc(km); // (the columns inside the row's key range are a prefix of the run)
                j += (unsigned)avail;
                live = goes_on && j < (unsigned)n_cols;
            }
            // ask for what comes behind the window now: it lands while this block's candidates are confirmed
            if (w_hi < n_cols_up) {
                if (__ballot(goes_on) != 0) { // a lane goes on in the next block
                    ahead = 1;
                    ahead_c0 = w_hi;
                    seg0 = st.load(ahead_c0);
                }
            }
            if (__ballot(live) == 0 && tile_next < t_hi) {
                // The tile's LAST block: ask for the NEXT tile's window now -- up to four segments, in registers while this
                // block's candidates are queued and confirmed (a tile used to start with a bare wait for its first loads:
                // at two waves per SIMD that wait was a third of the sweep).  Inside a chunk the window moves on and only
                // what lies behind it is asked for.
                const int row_n = row_begin + tile_next * 64 + lane;
                const unsigned j_n = ONE ? (unsigned)row_n + 1u : nx.aux.w;
                const unsigned base_n = wave_min_u32_dpp((row_n < row_end && j_n < (unsigned)n_cols) ? j_n : 0xFFFFFFFFu);
                if (base_n != 0xFFFFFFFFu) {
                    const unsigned nlo_n = base_n & ~(unsigned)(SW_SEG - 1);
                    const unsigned need_n = min(n_cols_up, nlo_n + (unsigned)SW_WIN);
                    const unsigned from_n = (next_follows && w_hi > nlo_n && w_lo <= nlo_n) ? w_hi : nlo_n;
                    ahead_c0 = from_n;
                    ahead = 0;
                    if (from_n < need_n) {
                        seg0 = st.load(from_n);
                        ahead = 1;
                    }
                    if (from_n + 1u * SW_SEG < need_n) {
                        seg1 = st.load(from_n + 1u * SW_SEG);
                        ahead = 2;
                    }
                    if (from_n + 2u * SW_SEG < need_n) {
                        seg2 = st.load(from_n + 2u * SW_SEG);
                        ahead = 3;
                    }
                    if (from_n + 3u * SW_SEG < need_n) {
                        seg3 = st.load(from_n + 3u * SW_SEG);
                        ahead = 4;
                    }
                }
            }
            if (__ballot(m != 0) == 0) continue;

            // ---- QUEUE: expand the hit masks into (row lane, column slot) candidates -- STQ's queue, 64 lanes wide.
            // (Every lane walking the hits of its own row instead, with the row in its registers, was measured: as many
            // rounds as the busiest lane has hits, 8 per block where the average lane has 1.4 -- the queue's full lanes win.)
            const int mine = __popc(m);
            unsigned total_u;
            const int incl = (int)wave_incl_scan_dpp((unsigned)mine, &total_u);
            const int total = (int)total_u;
            if (qcount == 0) q_lo = w_lo;
            if (qcount + total <= SW_QCAP) {
                int pos = qcount + incl - mine;
                while (m) {
                    const int b = __ffs((int)m) - 1;
                    m &= m - 1;
                    L.q[pos++] = ((uint32_t)lane << 8) | ((s0 + (unsigned)b) & (unsigned)(SW_WIN - 1));
                }
                qcount += total;
            } else {
                // crowded block: one candidate per lane per round
                for (;;) {
                    const bool hasm = m != 0;
                    const unsigned long long mask = __ballot(hasm);
                    if (mask == 0) break;
                    while (qcount > SW_QCAP - 64) confirm_batch(64, me, my_cellpack);
                    if (hasm) {
                        const int b = __ffs((int)m) - 1;
                        m &= m - 1;
                        L.q[qcount + mbcnt64(mask)] = ((uint32_t)lane << 8) | ((s0 + (unsigned)b) & (unsigned)(SW_WIN - 1));
                    }
                    qcount += popc64(mask);
                }
            }
            // ---- CONFIRM + EMIT full batches
            while (qcount >= 64) {
                ++d_rounds;
                confirm_batch(64, me, my_cellpack);
            }
        }
        drain(me, my_cellpack); // (the rows leave the registers)
        if (!next_follows) have = false; // the next tile is somewhere else: its window starts from nothing (what was asked for ahead)
        tile = tile_next;
    }
    wave_lds_fence();
    if (diag && lane == 0) {
        atomicAdd(&cnt->diag[0], (unsigned long long)d_blocks);
        atomicAdd(&cnt->diag[1], (unsigned long long)d_groups);
        atomicAdd(&cnt->diag[2], (unsigned long long)d_rounds);
        atomicAdd(&cnt->diag[3], (unsigned long long)d_segs);
    }
    em.flush_block(blk_scratch, tests, &cnt->cand_parts[blockIdx.x & 31u]);
}

template <bool ONE, int KIND>
__global__ __launch_bounds__(SW_THREADS, 2) void sweep_band_k(SweepRecs R, int row_begin, int row_end, SweepRecs C, int n_cols,
                                                              const GridParams* __restrict__ gp, int emit, int chunk,
                                                              int2* __restrict__ out, long long capacity,
                                                              SweepCounters* __restrict__ cnt, int diag,
                                                              const uint32_t* __restrict__ d_m_rows, const uint32_t* __restrict__ d_m_cols,
                                                              int expect_bits)
{
    // d_m_rows / d_m_cols (may be null): the lists' entry counts in DEVICE memory -- a sweep enqueued before the host knows
    // them (api.hip: the speculative build) is launched for the counts it expects and takes the real ones from here
    __shared__ SweepLds lds_s[SW_WAVES];
    __shared__ unsigned long long blk_scratch[3 * SW_WAVES];
    // (a count beyond what the record arrays were sized for is a failed guess: nothing is swept, the host finds out)
    // (so is another key width than the lists were sorted by: their records were never made)
    if ((d_m_rows && *d_m_rows > (uint32_t)row_end) || (d_m_cols && *d_m_cols > (uint32_t)n_cols) || (d_m_rows && gp->key_bits != expect_bits)) return;
    if (d_m_rows) row_end = min(row_end, (int)*d_m_rows);
    if (d_m_cols) n_cols = (int)*d_m_cols;
    sweep_band_body<ONE, KIND>(lds_s, blk_scratch, R, row_begin, row_end, C, n_cols, gp, emit, chunk, out, capacity, cnt, diag,
                               blockIdx.x, gridDim.x);
}

// Both classes of a two-list sweep in ONE launch: every block sweeps its share of rows A against columns B, then its share
// of rows B against columns A -- two launches in a row cost a launch gap and a second ramp-up and tail on the critical
// path of every vertex-face pass.  (Splitting the BLOCKS between the classes by their numbers of tiles was measured:
// a vertex row and a face row do not cost the same, the lighter class's blocks idle: 135 us against 111 for two launches.)
template <int KIND_A, int KIND_B>
__global__ __launch_bounds__(SW_THREADS, 2) void sweep_band2_k(SweepRecs A, int a_begin, int a_end, int n_a, SweepRecs B, int b_begin,
                                                               int b_end, int n_b, const GridParams* __restrict__ gp, int chunk_a,
                                                               int chunk_b, int2* __restrict__ out, long long capacity,
                                                               SweepCounters* __restrict__ cnt, int diag,
                                                               const uint32_t* __restrict__ d_tot /* [A, B] or null: sweep_band_k */, int expect_bits)
{
    __shared__ SweepLds lds_s[SW_WAVES];
    __shared__ unsigned long long blk_scratch[3 * SW_WAVES];
    if (d_tot) {
        if (d_tot[0] > (uint32_t)n_a || d_tot[1] > (uint32_t)n_b || gp->key_bits != expect_bits) return; // (a failed guess: sweep_band_k)
        n_a = (int)d_tot[0];
        n_b = (int)d_tot[1];
        a_end = min(a_end, n_a);
        b_end = min(b_end, n_b);
    }
    sweep_band_body<false, KIND_A>(lds_s, blk_scratch, A, a_begin, a_end, B, n_b, gp, EMIT_ROWS_A, chunk_a, out, capacity, cnt, diag,
                                   blockIdx.x, gridDim.x);
    __syncthreads(); // (the scratch words of the class's last flush)
    sweep_band_body<false, KIND_B>(lds_s, blk_scratch, B, b_begin, b_end, A, n_a, gp, EMIT_ROWS_B, chunk_b, out, capacity, cnt, diag,
                                   blockIdx.x, gridDim.x);
}

// Plain sweep-and-prune, one thread per row, exact boxes only (the reference's baseline
// variant sweep_and_prune<>, sweep.cu:48-99).  Kept as an in-library cross-check of the STQ
// kernel (SCCD_OPT_SWEEP_ALGO = 1); not tuned.
__global__ void sweep_sap_k(SweepRecs R, int row_begin, int row_end, SweepRecs C, int n_cols, int one,
                            const GridParams* __restrict__ gp, int emit, int2* __restrict__ out, long long capacity,
                            SweepCounters* __restrict__ cnt)
{
    const int row = row_begin + blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long tests = 0;
    if (row < row_end) {
        const int xb = gp->xb, Sb = gp->Sb;
        const bool one_cell = gp->n_cells <= 1;
        RowRegs me;
        load_row(me, R, row);
        const int cell = (int)((unsigned long long)me.aux.x >> xb);
        for (unsigned j = one ? (unsigned)row + 1u : me.aux.w; j < (unsigned)n_cols; j++) {
            RowRegs col;
            load_row(col, C, (int)j);
            const uint4 caux = col.aux;
            if (caux.x > me.aux.y) break;
            ++tests;
            const double2 cx = col.x, ca = col.a, cb = col.b;
            const int4 cv = col.id;
            const bool geo = (me.x.y >= cx.x) && (me.x.x <= cx.y) && (me.a.y >= ca.x) && (me.a.x <= ca.y) && (me.b.y >= cb.x)
                && (me.b.x <= cb.y);
            const bool share = me.id.x == cv.x || me.id.x == cv.y || me.id.x == cv.z || me.id.y == cv.x || me.id.y == cv.y
                || me.id.y == cv.z || me.id.z == cv.x || me.id.z == cv.y || me.id.z == cv.z;
            const int ma = (int)max(me.aux.z & 0xFFFFu, caux.z & 0xFFFFu), mb = (int)max(me.aux.z >> 16, caux.z >> 16);
            if (geo && !share && (one_cell || ma * Sb + mb == cell)) {
                const unsigned long long dst = atomicAdd(&cnt->n_pairs, 1ull);
                if ((long long)dst < capacity) out[dst] = make_pair_out(emit, me.id.w, cv.w);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tests += __shfl_xor(tests, o, 64);
    if (lane_id() == 0 && tests) atomicAdd(&cnt->cand_parts[blockIdx.x & 31u], tests);
}

} // namespace

void launch_sweep(sccd_ctx* c, const SortedList* rows, const SortedList* cols, const GridParams* gp, int row_begin,
                  int row_end, int emit, int2* out, int64_t capacity, SweepCounters* d_cnt, const uint32_t* d_m_rows,
                  const uint32_t* d_m_cols, int expect_bits)
{
    if (row_end <= row_begin || cols->m == 0) return;
    SCCD_REQUIRE(!(d_m_rows && c->sweep_algo == 1), "sweep: the plain sweep takes exact counts");
    const bool one = rows == cols;
    const SweepRecs R = sweep_recs(rows), C = sweep_recs(cols);
    if (c->sweep_algo == 1) {
        const int n = row_end - row_begin;
        hipLaunchKernelGGL(sweep_sap_k, dim3((n + 255) / 256), dim3(256), 0, c->stream, R, row_begin, row_end, C, cols->m,
                           one ? 1 : 0, gp, emit, out, (long long)capacity, d_cnt);
    } else {
        const int num_tiles = (row_end - row_begin + 63) / 64;
        // RESIDENT blocks only: tiles are dealt statically over the grid, so a block that has to wait
        // for a slot doubles the tail.  78 KB of LDS per block -> 2 blocks per CU; a multiple of 8 blocks (the XCD-aware deal).
        const int per_cu = c->sweep_blocks_per_cu > 0 ? c->sweep_blocks_per_cu : 2;
        int grid = std::min((num_tiles + SW_WAVES - 1) / SW_WAVES, c->num_cus * per_cu);
        grid = std::max(8, (grid + 7) / 8 * 8);
        // consecutive tiles per wave and deal (measured on 1M random boxes and the 1M-triangle cloth: a wave has only ~12
        // tiles per launch, and with 2 / 4 / 8 of them in a row the deal's quantisation costs more than the shared
        // windows save: 0.27 / 0.28 / 0.29 / 0.32 ms for chunks of 1 / 2 / 4 / 8)
        const int chunk = 1;
        // what the lists' vertex ids are known to be (filter_step): mesh-built lists test shared vertices in the filter
        int kind = 0;
        {
            if (one && rows->kind == BOX_EDGE) kind = 1;
            else if (!one && rows->kind == BOX_VERTEX && cols->kind == BOX_FACE) kind = 2;
            else if (!one && rows->kind == BOX_FACE && cols->kind == BOX_VERTEX) kind = 3;
        }
        auto go = [&](auto kernel) {
            const int diag = lab_env().sweep_diag;
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(SW_THREADS), 0, c->stream, R, row_begin, row_end, C, cols->m, gp, emit,
                               chunk, out, (long long)capacity, d_cnt, diag, d_m_rows, d_m_cols, expect_bits);
        };
        if (kind == 1) go(sweep_band_k<true, 1>);
        else if (kind == 2) go(sweep_band_k<false, 2>);
        else if (kind == 3) go(sweep_band_k<false, 3>);
        else if (one) go(sweep_band_k<true, 0>);
        else go(sweep_band_k<false, 0>);
    }
    SCCD_HIP(hipGetLastError());
}

// Both classes of a two-list sweep (rows A x columns B, rows B x columns A) in one launch.
void launch_sweep_two(sccd_ctx* c, const SortedList* A, const SortedList* B, const GridParams* gp, int a_begin, int a_end,
                      int b_begin, int b_end, int2* out, int64_t capacity, SweepCounters* d_cnt, const uint32_t* d_tot, int expect_bits)
{
    const int tiles_a = std::max(0, (a_end - a_begin + 63) / 64), tiles_b = std::max(0, (b_end - b_begin + 63) / 64);
    if (c->sweep_algo == 1 || tiles_a == 0 || tiles_b == 0 || A->m == 0 || B->m == 0) {
        launch_sweep(c, A, B, gp, a_begin, a_end, EMIT_ROWS_A, out, capacity, d_cnt, d_tot, d_tot ? d_tot + 1 : nullptr, expect_bits);
        launch_sweep(c, B, A, gp, b_begin, b_end, EMIT_ROWS_B, out, capacity, d_cnt, d_tot ? d_tot + 1 : nullptr, d_tot, expect_bits);
        return;
    }
    const int per_cu = c->sweep_blocks_per_cu > 0 ? c->sweep_blocks_per_cu : 2;
    const int tiles = std::max(tiles_a, tiles_b);
    int grid = std::min((tiles + SW_WAVES - 1) / SW_WAVES, c->num_cus * per_cu);
    grid = std::max(8, (grid + 7) / 8 * 8);
    const bool vf = A->kind == BOX_VERTEX && B->kind == BOX_FACE;
    const int diag = lab_env().sweep_diag;
    const SweepRecs RA = sweep_recs(A), RB = sweep_recs(B);
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(SW_THREADS), 0, c->stream, RA, a_begin, a_end, A->m, RB, b_begin, b_end, B->m, gp,
                           1, 1, out, (long long)capacity, d_cnt, diag, d_tot, expect_bits);
    };
    if (vf) go(sweep_band2_k<2, 3>);
    else go(sweep_band2_k<0, 0>);
    SCCD_HIP(hipGetLastError());
}
