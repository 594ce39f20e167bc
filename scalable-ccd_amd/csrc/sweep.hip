// sweep.hip -- candidate ranges + the sweep (STQ for wave64) that emits overlap pairs.
//
// Replaces sweep_and_tiniest_queue<> / sweep_and_prune<> with Queue, add_overlap,
// RawDeviceBuffer::push (src/scalable_ccd/cuda/broad_phase/sweep.cu:48-182, queue.cuh:5-49,
// collision.cuh:11-54, utils/device_buffer.cuh:56-61) and the CPU twin batched_sweep
// (src/scalable_ccd/broad_phase/sort_and_sweep.cpp:77-125).
//
// Set semantics (identical to the reference): all (a,b) whose boxes intersect INCLUSIVELY on
// x, y and z, that share no vertex id, and (two lists) come from different lists.
//
// Structure of sweep_stq_k, per wave of 64 lanes, persistent over tiles of 64 sorted rows:
//   FILTER   lane = row (its conservative float bounds in registers); the column records are
//            wave-uniform 16-byte loads.  32 columns per block; each lane collects a 32-bit
//            hit mask.  No cross-lane traffic in the inner loop.
//   QUEUE    hit masks are expanded into a per-wave LDS queue of (row, col) candidates
//            (one wave prefix-sum per block) -- the "tiniest queue" of STQ, 64 lanes wide.
//   CONFIRM  whenever >= 64 candidates are queued, every lane confirms one with the exact
//            double boxes (6 inclusive compares) and the 3x3 vertex-id test.
//   EMIT     survivors go to a per-wave LDS staging buffer; one global atomic per ~1000 pairs
//            reserves space, then the pairs are written coalesced (the reference: two global
//            atomics per pair, collision.cuh:45-54).
#include "internal.hpp"
#include "grid.hpp"

#include <algorithm>

namespace {

constexpr int SW_THREADS = 256;
constexpr int SW_WAVES = SW_THREADS / 64;
constexpr int SW_QCAP = 256;  // candidate queue entries per wave
constexpr int SW_OCAP = 1024; // staged output pairs per wave
constexpr int SW_BLOCK = 32;  // columns per filter block

// ------------------------------------------------------------------------------------------
// candidate ranges (prefix information of the sorted keys)
//   mode 0 one list : cols (i, ub(key, kmax_i))                     pairs i<j, K(min_j) <= K(max_i)
//   mode 1 rows A   : cols B with K(min_a) <= K(min_b) <= K(max_a)  [lb(keyB,key_a), ub(keyB,kmax_a))
//   mode 2 rows B   : cols A with K(min_b) <  K(min_a) <= K(max_b)  [ub(keyA,key_b), ub(keyA,kmax_b))
// Modes 1 and 2 partition the intersecting cross pairs (exactly one of K(min_a) <= K(min_b),
// K(min_b) < K(min_a) holds), so no pair is emitted twice and none is lost.
__device__ __forceinline__ unsigned lower_bound_u32(const uint32_t* __restrict__ a, unsigned n, uint32_t v)
{
    unsigned lo = 0, hi = n;
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ unsigned upper_bound_u32(const uint32_t* __restrict__ a, unsigned n, uint32_t v)
{
    unsigned lo = 0, hi = n;
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// One block = 1024 consecutive sorted rows.  Their candidate ranges all lie inside one window of
// the column list ([lb(first row's key), ub(largest max-key of the block))), found with two
// full binary searches by one lane; every row then searches only inside the window, which is a
// few hundred entries that stay in this CU's L1.
__device__ __forceinline__ unsigned lower_bound_in(const uint32_t* __restrict__ a, unsigned lo, unsigned hi, uint32_t v)
{
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ unsigned upper_bound_in(const uint32_t* __restrict__ a, unsigned lo, unsigned hi, uint32_t v)
{
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// lower_bound / upper_bound over a sorted array by ONE WAVE: 64 probes per round instead of one
// (a 5M-entry list takes 4 dependent rounds of loads instead of 23).  UPPER: first index with
// a[i] > v, else first index with a[i] >= v.  All 64 lanes must call it with the same arguments.
template <bool UPPER> __device__ __forceinline__ unsigned wave_bound_u32(const uint32_t* __restrict__ a, unsigned n, uint32_t v)
{
    unsigned lo = 0, hi = n; // the answer lies in [lo, hi]
    const unsigned lane = (unsigned)lane_id();
    while (hi - lo > 64u) {
        // 64 probes cut [lo, hi) into 65 pieces
        const unsigned long long span = (unsigned long long)(hi - lo);
        const unsigned pos = lo + (unsigned)(span * (lane + 1u) / 65ull);
        const uint32_t x = a[pos < hi ? pos : hi - 1u];
        const bool before = (pos < hi) && (UPPER ? (x <= v) : (x < v)); // the answer is beyond pos
        const unsigned long long m = __ballot(before);
        // probes are increasing and the predicate is monotone: m is a run of low bits
        const int k = popc64(m);
        const unsigned new_lo = k == 0 ? lo : lo + (unsigned)(span * (unsigned long long)k / 65ull) + 1u;
        const unsigned new_hi = k == 64 ? hi : lo + (unsigned)(span * (unsigned long long)(k + 1) / 65ull);
        lo = new_lo;
        hi = new_hi < new_lo ? new_lo : new_hi;
    }
    // at most 64 candidates left: one probe each
    const unsigned pos = lo + lane;
    const bool before = pos < hi && (UPPER ? (a[pos] <= v) : (a[pos] < v));
    return lo + (unsigned)popc64(__ballot(before));
}

constexpr int RG_THREADS = 1024; // rows per block: one window search and one statistics atomic per block
constexpr int RG_WAVES = RG_THREADS / 64;
__global__ __launch_bounds__(RG_THREADS) void ranges_k(const uint32_t* __restrict__ key_r, const uint32_t* __restrict__ kmax_r,
                                                int n_rows, const uint32_t* __restrict__ key_c, int n_cols, int mode,
                                                uint2* __restrict__ ranges, unsigned long long* __restrict__ candidates)
{
    __shared__ uint32_t s_kmax[RG_WAVES];
    __shared__ unsigned s_win[2];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n_rows;
    const uint32_t k_lo = valid ? key_r[i] : 0xFFFFFFFFu;
    const uint32_t k_hi = valid ? kmax_r[i] : 0u;
    const uint32_t wmax = wave_max_u32(k_hi);
    if (lane_id() == 0) s_kmax[threadIdx.x >> 6] = wmax;
    __syncthreads();
    if (threadIdx.x < 64) { // the first wave finds the window, 64 probes per round
        uint32_t bmax = s_kmax[0];
#pragma unroll
        for (int k = 1; k < RG_WAVES; k++) bmax = max(bmax, s_kmax[k]);
        // rows are sorted: the first row of the block has the smallest key
        const uint32_t k_first = (uint32_t)__shfl((int)k_lo, 0, 64);
        const unsigned i_first = (unsigned)(blockIdx.x * blockDim.x);
        const unsigned w_lo = (mode == 0) ? i_first + 1u : wave_bound_u32<false>(key_c, (unsigned)n_cols, k_first);
        unsigned w_hi = wave_bound_u32<true>(key_c, (unsigned)n_cols, bmax);
        if (w_hi < w_lo) w_hi = w_lo;
        if (threadIdx.x == 0) {
            s_win[0] = w_lo;
            s_win[1] = w_hi;
        }
    }
    __syncthreads();
    const unsigned w0 = s_win[0], w1 = s_win[1];
    unsigned long long cnt = 0;
    if (valid) {
        unsigned s, e;
        if (mode == 0) {
            s = (unsigned)i + 1;
            e = upper_bound_in(key_c, max(s, w0), w1, k_hi);
        } else if (mode == 1) {
            s = lower_bound_in(key_c, w0, w1, k_lo);
            e = upper_bound_in(key_c, s, w1, k_hi);
        } else {
            s = upper_bound_in(key_c, w0, w1, k_lo);
            e = upper_bound_in(key_c, s, w1, k_hi);
        }
        if (e < s) e = s;
        ranges[i] = make_uint2(s, e);
        cnt = e - s;
    }
    // work metric only: one atomic per BLOCK, spread over 32 words (a single word serialises
    // at ~90 atomics/us chip-wide, which used to cost more than the searches)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    __shared__ unsigned long long s_cnt[RG_WAVES];
    if (lane_id() == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
#pragma unroll
        for (int k = 0; k < RG_WAVES; k++) tot += s_cnt[k];
        if (tot) atomicAdd(candidates + (blockIdx.x & 31), tot);
    }
}

// ------------------------------------------------------------------------------------------
struct ExactBox {
    double lo[3], hi[3];
    int v[3];
    int eid;
};
__device__ __forceinline__ ExactBox load_exact(const sccd_aabb* __restrict__ b)
{
    const double4 q0 = reinterpret_cast<const double4*>(b)[0];
    const double2 q1 = reinterpret_cast<const double2*>(b)[2];
    const int4 q2 = reinterpret_cast<const int4*>(b)[3];
    ExactBox r;
    r.lo[0] = q0.x;
    r.lo[1] = q0.y;
    r.lo[2] = q0.z;
    r.hi[0] = q0.w;
    r.hi[1] = q1.x;
    r.hi[2] = q1.y;
    r.v[0] = q2.x;
    r.v[1] = q2.y;
    r.v[2] = q2.z;
    r.eid = q2.w;
    return r;
}
// AABB::intersects (aabb.cuh:68-73) && !share_a_vertex (collision.cuh:17-21)
__device__ __forceinline__ bool exact_pair_ok(const ExactBox& a, const ExactBox& b)
{
    // (no short-circuits: a wave confirms 64 candidates in lockstep, a skipped comparison saves nothing)
    const bool geo = (a.hi[0] >= b.lo[0]) & (a.lo[0] <= b.hi[0]) & (a.hi[1] >= b.lo[1]) & (a.lo[1] <= b.hi[1])
        & (a.hi[2] >= b.lo[2]) & (a.lo[2] <= b.hi[2]);
    const bool share = (a.v[0] == b.v[0]) | (a.v[0] == b.v[1]) | (a.v[0] == b.v[2]) | (a.v[1] == b.v[0])
        | (a.v[1] == b.v[1]) | (a.v[1] == b.v[2]) | (a.v[2] == b.v[0]) | (a.v[2] == b.v[1]) | (a.v[2] == b.v[2]);
    return geo & !share;
}
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
    // a flush per wave made the tail of each launch a queue of 4096 atomics.)
    // `scratch`: 2 * SW_WAVES u64 of LDS no wave uses any more (the kernel's LDS budget is exactly
    // 4 blocks per CU: not one byte may be added)
    __device__ __forceinline__ void flush_block(unsigned long long* scratch)
    {
        unsigned long long* blk_cnt = scratch;
        unsigned long long* blk_base = scratch + SW_WAVES;
        const int w = (int)(threadIdx.x >> 6);
        __syncthreads(); // every wave is done with the scratch area
        if (lane_id() == 0) blk_cnt[w] = (unsigned long long)ocount;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long sum = 0;
            for (int k = 0; k < SW_WAVES; k++) {
                blk_base[k] = sum;
                sum += blk_cnt[k];
            }
            const unsigned long long b0 = sum ? atomicAdd(n_pairs, sum) : 0ull;
            for (int k = 0; k < SW_WAVES; k++) blk_base[k] += b0;
        }
        __syncthreads();
        const unsigned long long base = blk_base[w];
        for (int k = lane_id(); k < ocount; k += 64) {
            const unsigned long long dst = base + (unsigned long long)k;
            if ((long long)dst < capacity) out[dst] = stage[k];
        }
        ocount = 0;
    }
    __device__ __forceinline__ void push(bool ok, int2 pr)
    {
        const unsigned long long mask = __ballot(ok);
        if (mask == 0) return;
        if (ok) stage[ocount + mbcnt64(mask)] = pr;
        ocount += popc64(mask);
        if (ocount > SW_OCAP - 64) flush();
    }
};

__device__ __forceinline__ double sel3(const double v[3], int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); }

// A box listed in several cells meets the same partner in each of them; the pair is reported
// only from the cell (max of the two boxes' lowest cells per minor axis), which both boxes
// share iff they overlap on that axis (grid.hpp).
__device__ __forceinline__ bool owns_pair(const GridParams& g, uint32_t row_key, const ExactBox& a, const ExactBox& b)
{
    if (g.n_cells <= 1) return true;
    const int cell = (int)((unsigned long long)row_key >> g.xb);
    const int ma = max(grid_cell_a(g, sel3(a.lo, g.aa)), grid_cell_a(g, sel3(b.lo, g.aa)));
    const int mb = max(grid_cell_b(g, sel3(a.lo, g.ab)), grid_cell_b(g, sel3(b.lo, g.ab)));
    return ma * g.Sb + mb == cell; // (= the row's cell (ca, cb): 0 <= mb < Sb, so the two coordinates need no division to compare)
}

// low_r / low_c: the lowest cells of the boxes per minor axis as entry_gather_k stored them (the values owns_pair would
// compute again from the exact boxes: twenty FP64 instructions per candidate against two 4-byte loads of lines that the
// filter stage has just touched)
__device__ __forceinline__ void confirm(bool active, uint2 cand, const sccd_aabb* __restrict__ box_r,
                                        const uint32_t* __restrict__ key_r, const uint32_t* __restrict__ low_r,
                                        const sccd_aabb* __restrict__ box_c, const uint32_t* __restrict__ low_c,
                                        const GridParams& g, int emit, Emitter& em)
{
    bool ok = false;
    int2 pr = make_int2(0, 0);
    if (active) {
        const ExactBox a = load_exact(box_r + cand.x);
        const ExactBox b = load_exact(box_c + cand.y);
        const uint32_t la = low_r[cand.x], lb = low_c[cand.y];
        const int cell = (int)((unsigned long long)key_r[cand.x] >> g.xb);
        const int ma = (int)max(la & 0xFFFFu, lb & 0xFFFFu), mb = (int)max(la >> 16, lb >> 16);
        const bool owner = (g.n_cells <= 1) | (ma * g.Sb + mb == cell); // owns_pair on the stored cells
        ok = (int)exact_pair_ok(a, b) & (int)owner;
        pr = make_pair_out(emit, a.eid, b.eid);
    }
    em.push(ok, pr);
}

__global__ __launch_bounds__(SW_THREADS, 4) void sweep_stq_k(
    const float4* __restrict__ filt_r, const sccd_aabb* __restrict__ box_r, const uint32_t* __restrict__ key_r,
    const uint32_t* __restrict__ low_r, const uint2* __restrict__ ranges, int row_begin, int row_end,
    const float4* __restrict__ filt_c, const sccd_aabb* __restrict__ box_c, const uint32_t* __restrict__ low_c,
    const GridParams* __restrict__ gp, int emit, int2* __restrict__ out, long long capacity, SweepCounters* __restrict__ cnt)
{
    const GridParams g = *gp;
    __shared__ uint2 q_s[SW_WAVES][SW_QCAP];
    __shared__ int2 o_s[SW_WAVES][SW_OCAP];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    uint2* q = q_s[w];
    Emitter em { o_s[w], 0, out, capacity, &cnt->n_pairs };
    int qcount = 0;
    const int num_tiles = (row_end - row_begin + 63) / 64;

    // Tiles are dealt round-robin to the waves of the grid.  (A ticket per tile made the whole
    // sweep ticket-bound: one hot word serves ~90 atomics/us chip-wide, and a launch has ~27k
    // tiles that each take only a few microseconds.)
    const int n_waves = (int)gridDim.x * SW_WAVES;
    // the rows of the NEXT tile are requested before the current one is processed (their
    // latency hides behind the filter / confirm work)
    uint2 rg_n = make_uint2(0u, 0u);
    float4 fr_n = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const int row0 = row_begin + ((int)blockIdx.x * SW_WAVES + w) * 64 + lane;
        if (row0 < row_end) {
            rg_n = ranges[row0];
            fr_n = filt_r[row0];
        }
    }
    for (int tile = (int)blockIdx.x * SW_WAVES + w; tile < num_tiles; tile += n_waves) {
        const int row = row_begin + tile * 64 + lane;
        const bool valid = row < row_end;
        const uint2 rg = rg_n;
        const float4 fr = fr_n;
        {
            const int row_next = row + n_waves * 64;
            rg_n = make_uint2(0u, 0u);
            fr_n = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tile + n_waves < num_tiles && row_next < row_end) {
                rg_n = ranges[row_next];
                fr_n = filt_r[row_next];
            }
        }
        const bool nonempty = valid && rg.y > rg.x;
        const unsigned jmin = readfirst_u32(wave_min_u32(nonempty ? rg.x : 0xFFFFFFFFu));
        const unsigned jmax = readfirst_u32(wave_max_u32(nonempty ? rg.y : 0u));
        if (jmin >= jmax) continue;

        for (unsigned j0 = jmin & ~(unsigned)(SW_BLOCK - 1); j0 < jmax; j0 += SW_BLOCK) {
            // this lane's live columns inside [j0, j0+32)
            const unsigned lo = max(rg.x, j0), hi = min(rg.y, j0 + SW_BLOCK);
            unsigned live = 0;
            if (hi > lo) {
                const unsigned len = hi - lo;
                live = (len >= 32u ? 0xFFFFFFFFu : ((1u << len) - 1u)) << (lo - j0);
            }
            unsigned m = 0;
            const float4* __restrict__ cb = filt_c + j0; // wave-uniform address
#pragma unroll
            for (int b = 0; b < SW_BLOCK; b++) {
                const float4 cc = cb[b];
                const bool hit = (cc.x <= fr.y) & (fr.x <= cc.y) & (cc.z <= fr.w) & (fr.z <= cc.w);
                m |= hit ? (1u << b) : 0u;
            }
            m &= live;
            if (__ballot(m != 0) == 0) continue;

            // ---- expand the hit masks into the candidate queue
            const int mine = __popc(m);
            const int incl = wave_incl_scan(mine);
            const int total = (int)readfirst_u32((unsigned)__shfl(incl, 63, 64));
            if (qcount + total <= SW_QCAP) {
                int pos = qcount + incl - mine;
                while (m) {
                    const int b = __ffs((int)m) - 1;
                    m &= m - 1;
                    q[pos++] = make_uint2((unsigned)row, j0 + (unsigned)b);
                }
                qcount += total;
            } else {
                // crowded block: one candidate per lane per round
                for (;;) {
                    const bool has = m != 0;
                    const unsigned long long mask = __ballot(has);
                    if (mask == 0) break;
                    while (qcount > SW_QCAP - 64) {
                        wave_lds_fence();
                        const uint2 cand = q[qcount - 64 + lane];
                        qcount -= 64;
                        confirm(true, cand, box_r, key_r, low_r, box_c, low_c, g, emit, em);
                    }
                    if (has) {
                        const int b = __ffs((int)m) - 1;
                        m &= m - 1;
                        q[qcount + mbcnt64(mask)] = make_uint2((unsigned)row, j0 + (unsigned)b);
                    }
                    qcount += popc64(mask);
                }
            }
            // ---- confirm full batches
            while (qcount >= 64) {
                wave_lds_fence();
                const uint2 cand = q[qcount - 64 + lane];
                qcount -= 64;
                confirm(true, cand, box_r, key_r, low_r, box_c, low_c, g, emit, em);
            }
        }
    }
    // drain
    if (qcount > 0) {
        wave_lds_fence();
        const bool act = lane < qcount;
        const uint2 cand = act ? q[lane] : make_uint2(0u, 0u);
        confirm(act, cand, box_r, key_r, low_r, box_c, low_c, g, emit, em);
    }
    wave_lds_fence();
    em.flush_block(reinterpret_cast<unsigned long long*>(&q_s[0][0])); // (the candidate queues are empty now)
}

// Direct exact sweep: the kernel of choice once the cell grid has cut the candidates down to a
// few per emitted pair.  Lane = row with its EXACT box, ids and lowest cell in registers; the
// column records (64-byte box + lowest cell) are wave-uniform, i.e. scalar loads; the exact
// inclusive test, the vertex-id test and the owner-cell test run in the loop and survivors go
// straight to the staged emitter.  No candidate queue, no dependent gathers.
__global__ __launch_bounds__(SW_THREADS) void sweep_direct_k(
    const sccd_aabb* __restrict__ box_r, const uint32_t* __restrict__ key_r, const uint32_t* __restrict__ low_r,
    const uint2* __restrict__ ranges, int row_begin, int row_end, const sccd_aabb* __restrict__ box_c,
    const uint32_t* __restrict__ low_c, const GridParams* __restrict__ gp, int emit, int2* __restrict__ out,
    long long capacity, SweepCounters* __restrict__ cnt)
{
    __shared__ int2 o_s[SW_WAVES][SW_OCAP];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    Emitter em { o_s[w], 0, out, capacity, &cnt->n_pairs };
    const int xb = gp->xb, Sb = gp->Sb;
    const bool one_cell = gp->n_cells <= 1;
    const int num_tiles = (row_end - row_begin + 63) / 64;
    // Tiles are dealt round-robin to the waves of the grid.  (A ticket per tile made the whole
    // sweep ticket-bound: one hot word serves ~90 atomics/us chip-wide, and a launch has ~27k
    // tiles that each take only a few microseconds.)
    const int n_waves = (int)gridDim.x * SW_WAVES;
    for (int tile = (int)blockIdx.x * SW_WAVES + w; tile < num_tiles; tile += n_waves) {
        const int row = row_begin + tile * 64 + lane;
        const bool valid = row < row_end;
        uint2 rg = make_uint2(0u, 0u);
        ExactBox a;
        int ca = 0, cb = 0, la = 0, lb = 0;
        if (valid) {
            rg = ranges[row];
            a = load_exact(box_r + row);
            const int cell = (int)((unsigned long long)key_r[row] >> xb);
            ca = cell / Sb;
            cb = cell - ca * Sb;
            const uint32_t lw = low_r[row];
            la = (int)(lw & 0xFFFFu);
            lb = (int)(lw >> 16);
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                a.lo[k] = 1.0; // empty box: never intersects
                a.hi[k] = -1.0;
                a.v[k] = 0;
            }
            a.eid = 0;
        }
        const bool nonempty = valid && rg.y > rg.x;
        const unsigned jmin = readfirst_u32(wave_min_u32(nonempty ? rg.x : 0xFFFFFFFFu));
        const unsigned jmax = readfirst_u32(wave_max_u32(nonempty ? rg.y : 0u));
        for (unsigned j0 = jmin; j0 < jmax; j0 += 4) { // wave-uniform columns, 4 scalar records per wait
            ExactBox b[4];
            uint32_t lw[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { // the arrays are padded: reading past the last column is harmless
                b[u] = load_exact(box_c + j0 + u);
                lw[u] = low_c[j0 + u];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const unsigned j = j0 + (unsigned)u;
                bool ok = j >= rg.x && j < rg.y && exact_pair_ok(a, b[u]);
                if (!one_cell) ok = ok && max(la, (int)(lw[u] & 0xFFFFu)) == ca && max(lb, (int)(lw[u] >> 16)) == cb;
                em.push(ok, make_pair_out(emit, a.eid, b[u].eid));
            }
        }
    }
    em.flush();
}

// Plain sweep-and-prune, one thread per row, exact boxes only (the reference's baseline
// variant sweep_and_prune<>, sweep.cu:48-99).  Kept as an in-library cross-check of the STQ
// kernel (SCCD_OPT_SWEEP_ALGO = 1); not tuned.
__global__ void sweep_sap_k(const sccd_aabb* __restrict__ box_r, const uint32_t* __restrict__ key_r,
                            const uint2* __restrict__ ranges, int row_begin, int row_end,
                            const sccd_aabb* __restrict__ box_c, const GridParams* __restrict__ gp, int emit,
                            int2* __restrict__ out, long long capacity, SweepCounters* __restrict__ cnt)
{
    const int row = row_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= row_end) return;
    const GridParams g = *gp;
    const uint2 rg = ranges[row];
    const ExactBox a = load_exact(box_r + row);
    const uint32_t rk = key_r[row];
    for (unsigned j = rg.x; j < rg.y; j++) {
        const ExactBox b = load_exact(box_c + j);
        if (exact_pair_ok(a, b) && owns_pair(g, rk, a, b)) {
            const unsigned long long dst = atomicAdd(&cnt->n_pairs, 1ull);
            if ((long long)dst < capacity) out[dst] = make_pair_out(emit, a.eid, b.eid);
        }
    }
}

} // namespace

void launch_ranges(sccd_ctx* c, const SortedList* rows, const SortedList* cols, int mode, uint2* ranges,
                   unsigned long long* d_candidates)
{
    if (rows->m == 0) return;
    const int grid = (rows->m + RG_THREADS - 1) / RG_THREADS;
    hipLaunchKernelGGL(ranges_k, dim3(grid), dim3(RG_THREADS), 0, c->stream, rows->key.as<uint32_t>(),
                       rows->kmax.as<uint32_t>(), rows->m, cols->key.as<uint32_t>(), cols->m, mode, ranges,
                       d_candidates);
    SCCD_HIP(hipGetLastError());
}

void launch_sweep(sccd_ctx* c, const SortedList* rows, const SortedList* cols, const GridParams* gp,
                  const uint2* ranges, int row_begin, int row_end, int emit, int2* out, int64_t capacity,
                  SweepCounters* d_cnt, bool direct)
{
    if (row_end <= row_begin || cols->m == 0) return;
    if (c->sweep_algo == 1) {
        const int n = row_end - row_begin;
        hipLaunchKernelGGL(sweep_sap_k, dim3((n + 255) / 256), dim3(256), 0, c->stream, rows->box.as<sccd_aabb>(),
                           rows->key.as<uint32_t>(), ranges, row_begin, row_end, cols->box.as<sccd_aabb>(), gp, emit,
                           out, (long long)capacity, d_cnt);
    } else if (direct) {
        const int num_tiles = (row_end - row_begin + 63) / 64;
        const int grid = std::max(1, std::min((num_tiles + SW_WAVES - 1) / SW_WAVES, c->num_cus * 4));
        hipLaunchKernelGGL(sweep_direct_k, dim3(grid), dim3(SW_THREADS), 0, c->stream, rows->box.as<sccd_aabb>(),
                           rows->key.as<uint32_t>(), rows->lowcell.as<uint32_t>(), ranges, row_begin, row_end,
                           cols->box.as<sccd_aabb>(), cols->lowcell.as<uint32_t>(), gp, emit, out,
                           (long long)capacity, d_cnt);
    } else {
        const int num_tiles = (row_end - row_begin + 63) / 64;
        // RESIDENT blocks only: tiles are dealt statically over the grid, so a block that has to wait
        // for a slot doubles the tail.  128 VGPRs (launch bounds) and 40 KB of LDS -> 4 blocks per CU.
        // (With 156 VGPRs only 3 of the 4 blocks per CU were resident: sweep 0.61 -> 0.47 ms on C4.)
        static const int per_cu = std::getenv("SCCD_SWEEP_BLOCKS") ? std::atoi(std::getenv("SCCD_SWEEP_BLOCKS")) : 4;
        const int grid = std::max(1, std::min((num_tiles + SW_WAVES - 1) / SW_WAVES,
                                              c->num_cus * (c->sweep_blocks_per_cu > 0 ? c->sweep_blocks_per_cu : per_cu)));
        hipLaunchKernelGGL(sweep_stq_k, dim3(grid), dim3(SW_THREADS), 0, c->stream, rows->filt.as<float4>(),
                           rows->box.as<sccd_aabb>(), rows->key.as<uint32_t>(), rows->lowcell.as<uint32_t>(), ranges, row_begin,
                           row_end, cols->filt.as<float4>(), cols->box.as<sccd_aabb>(), cols->lowcell.as<uint32_t>(), gp, emit,
                           out, (long long)capacity, d_cnt);
    }
    SCCD_HIP(hipGetLastError());
}
