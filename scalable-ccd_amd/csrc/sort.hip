// sort.hip -- hand-written LSD radix sort of (uint32 key, uint32 value) pairs for gfx950.
//
// Replaces thrust::sort_by_key with a comparator on Scalar2 keys moving 48-byte MiniBox payloads
// (src/scalable_ccd/cuda/broad_phase/aabb.cu:107-109): here only an 8-byte (key, index) pair
// moves through the passes and the payload is gathered once afterwards (boxes.hip).
//
// Default: ONESWEEP -- 4 passes x 8-bit digits, 6 launches in total:
//   os_hist_k   digit histograms of all four passes in one read of the keys (4 B/key)
//   os_bases_k  reduction of the per-block histograms + exclusive scan over the digits
//   os_pass_k   x4: each 4096-key tile ranks its keys, publishes its per-digit counts and
//               obtains its global offsets by DECOUPLED LOOK-BACK over the preceding tiles
//               (one relaxed agent-scope 32-bit word per (tile, digit): flag + count in the same
//               word, so no separate payload needs ordering), then scatters through LDS so that
//               each digit's run leaves the CU as consecutive addresses.
//               Per pass: 8 B read + 8 B written per pair -- the algorithmic minimum for LSD.
// Ranking is wave64-native: the peers of a lane (same digit) are found with 8 ballots, the rank
// inside the 64-key row is the popcount of the peers below the lane (mbcnt), one leader lane
// per digit bumps the wave's LDS counter.  Order inside a tile is (wave, row, lane) = index
// order, so the sort is stable.  Tiles take their index from an atomic ticket, so a tile only
// ever waits for tiles that already run (no dependence on the dispatch order).
//
// (The simple count / scan / scatter pipeline of round 1 -- SCCD_SORT=classic -- is gone: tests/test_gpu_parity.py checks the
// sort against torch.sort instead.)
#include "internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

// A tile of 4,096 keys is ranked by EIGHT waves of 8 rows (round 5, late; rounds 1-4: four waves of 16): at the sizes the broad phase sorts
// a pass is a chain of latencies, not of bytes -- load, rank, stage and write all run over a wave's rows one after the other -- and twice
// the waves per tile shorten that part of it: os_pass_k 23.8 -> 21.6 us at 1.9 M pairs alone (29 -> 26 in the step), the step's sort class 0.226 -> 0.197 ms; 16 M keys: the
// same (0.534 / 0.536 ms).  Sixteen waves of 4 rows: 66 KB of LDS per block, slower on both (0.25 ms; 0.64 ms).  103 registers: two blocks
// per CU by registers (three by LDS: 50 KB), 512 tiles resident at once -- enough for the broad phase's ~460.
// (RS_THREADS_ / RS_ITEMS_: tools/variants.sh)
#ifndef RS_THREADS_
#define RS_THREADS_ 512
#endif
#ifndef RS_ITEMS_
#define RS_ITEMS_ 8
#endif
constexpr int RS_THREADS = RS_THREADS_;        // >= 256: thread d < 256 owns digit d
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = RS_ITEMS_;            // rows of 64 keys per wave (256 threads x 8: -30 %, x 24: +7 % at 16M keys but -10 % at 2-5M)
constexpr int RS_TILE = RS_THREADS * RS_ITEMS; // 4096 keys per tile
constexpr int RS_WAVE_SPAN = RS_ITEMS * 64;    // keys owned by one wave

constexpr uint32_t OS_FLAG_AGG = 1u << 30;    // word holds the tile's own count
constexpr uint32_t OS_FLAG_PREFIX = 1u << 31; // word holds the inclusive prefix up to this tile
constexpr uint32_t OS_VALUE_MASK = (1u << 30) - 1u;

// stable rank of every key of this wave's rows among the keys of the wave with the same digit;
// wcnt[digit] (LDS, zeroed) ends up as the wave's digit counts
__device__ __forceinline__ void wave_rank_rows(const uint32_t (&key)[RS_ITEMS], const bool (&valid)[RS_ITEMS],
                                               int shift, uint32_t* wcnt, uint32_t (&rank)[RS_ITEMS])
{
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const uint32_t d = (key[r] >> shift) & 255u;
        unsigned long long peers = __ballot(valid[r]);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const int below = mbcnt64(peers);
        const int leader = (int)__builtin_ctzll(peers | (1ull << 63)); // lowest peer lane
        uint32_t old = 0;
        if (valid[r] && below == 0) { // the leader of this digit group
            old = wcnt[d];
            wcnt[d] = old + (uint32_t)popc64(peers);
        }
        wave_lds_fence();
        old = (uint32_t)__shfl((int)old, leader, 64);
        rank[r] = old + (uint32_t)below;
    }
}

// The same ranks from LDS "match" words instead of ballots (OS_RANK_MATCH=1; off): every lane ORs its lane bit into the
// 64-bit word of its digit, reads the word back -- its peers -- and the lowest peer adds the group's size to the wave's
// counter and clears the word.  One wave's LDS instructions execute in issue order, so the read-back follows every
// lane's OR and a row's counter reads precede its leaders' adds; with four match arrays the rows of a group of four
// are in flight together.  40 % fewer instructions per tile than the ballot form (1,907 against 3,136) and still
// SLOWER where it matters: 126 against 115 us per pass at 16M pairs, equal at 2-5M -- the 64-bit LDS atomics cost
// more than the ballots they replace.  Kept as the measured alternative.
// match: 4 x 256 words, zero on entry and zero again on return.
#ifndef OS_RANK_MATCH
#define OS_RANK_MATCH 0
#endif
__device__ __forceinline__ void wave_rank_rows_match(const uint32_t (&key)[RS_ITEMS], const bool (&valid)[RS_ITEMS], int shift,
                                                     uint32_t* wcnt, unsigned long long* match_a /* arrays 0, 1 */,
                                                     unsigned long long* match_b /* arrays 2, 3 */, uint32_t (&rank)[RS_ITEMS])
{
    const unsigned long long me = 1ull << lane_id();
#pragma unroll
    for (int g = 0; g < RS_ITEMS; g += 4) {
        unsigned long long* word[4];
        uint32_t d[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            d[j] = (key[g + j] >> shift) & 255u;
            word[j] = (j < 2 ? match_a : match_b) + (j & 1) * 256 + d[j];
            if (valid[g + j]) atomicOr(word[j], me);
        }
        wave_lds_fence();
        unsigned long long peers[4];
#pragma unroll
        for (int j = 0; j < 4; j++) peers[j] = *word[j];
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int below = mbcnt64(peers[j]);
            const uint32_t old = wcnt[d[j]];
            wave_lds_fence();
            if (valid[g + j] && below == 0) { // the lowest lane of the digit's group in this row
                atomicAdd(&wcnt[d[j]], (uint32_t)popc64(peers[j]));
                *word[j] = 0ull;
            }
            wave_lds_fence();
            rank[g + j] = old + (uint32_t)below;
        }
    }
}

// ---- onesweep ---------------------------------------------------------------------------------
constexpr int HS_THREADS = 1024; // one block per CU, sixteen waves: the loads of many waves in flight hide the HBM latency

// One more key in the digit histograms of the `passes` passes.  The low digits (quantised sort coordinate) are spread: plain
// LDS atomics.  The TOP digit of the broad phase's keys is the high bits of the box's cell, and the boxes a wave reads are
// neighbours: when all 64 keys share it (64 atomics on one LDS word are served one after the other) one lane adds 64.
// Digits past the key's width are not counted at all (a 24-bit key has 1.7 M zeros there: one word, fully serialised).
// Wave-wide: every lane of the wave must call it.
__device__ __forceinline__ void hist_add(uint32_t (*h)[256], uint32_t k, bool valid, int passes)
{
    for (int p = 0; p < passes - 1; p++)
        if (valid) atomicAdd(&h[p][(k >> (8 * p)) & 255u], 1u);
    const int p = passes - 1;
    const uint32_t d = (k >> (8 * p)) & 255u;
    const unsigned long long todo = __ballot(valid);
    if (todo == 0) return;
    const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)d, (int)__builtin_ctzll(todo));
    if (__ballot(valid && d == dl) == todo) {
        if (lane_id() == (int)__builtin_ctzll(todo)) atomicAdd(&h[p][dl], (uint32_t)popc64(todo));
    } else if (valid) {
        atomicAdd(&h[p][d], 1u);
    }
}

// d_n_real (may be null): the first *d_n_real of the n keys are there; the rest count as 0xFFFFFFFF -- a build that is sorted
// before the host knows its entry count sorts a padded number of keys (api.hip: the speculative build), and the padding must
// end up behind the keys, whatever the buffers hold there.
__global__ __launch_bounds__(HS_THREADS) void os_hist_k(const uint32_t* __restrict__ keys, long long n, uint32_t* __restrict__ partial,
                                                        uint4* __restrict__ zero, long long zero_n, int passes,
                                                        const uint32_t* __restrict__ d_n_real)
{
    const long long n_real = d_n_real ? min((long long)*d_n_real, n) : n;
    // every word the passes poll (tickets + look-back status) is zeroed here, not by memsets
    for (long long i = (long long)blockIdx.x * HS_THREADS + threadIdx.x; i < zero_n; i += (long long)gridDim.x * HS_THREADS)
        zero[i] = make_uint4(0u, 0u, 0u, 0u);
    __shared__ uint32_t h[4][256];
    h[threadIdx.x >> 8][threadIdx.x & 255] = 0;
    __syncthreads();
    // 16 keys per thread and round (four 16-byte loads in flight per thread), rounds strided over the grid
    const long long quads = (n + 3) / 4;
    const long long span = (long long)gridDim.x * HS_THREADS;
    // (wave-uniform trip count: hist_add is a wave-wide operation)
    for (long long q00 = (long long)blockIdx.x * HS_THREADS; q00 < quads; q00 += span * 4) {
        const long long q0 = q00 + threadIdx.x;
        uint4 k4[4];
        int cnt4[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const long long q = q0 + (long long)r * span;
            const long long i = q * 4;
            cnt4[r] = 0;
            k4[r] = make_uint4(0u, 0u, 0u, 0u);
            if (q < quads) {
                if (i + 3 < n_real) {
                    k4[r] = *reinterpret_cast<const uint4*>(keys + i);
                    cnt4[r] = 4;
                } else { // the end of the keys, or of the padded run behind them
                    k4[r] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                    if (i < n_real) k4[r].x = keys[i];
                    if (i + 1 < n_real) k4[r].y = keys[i + 1];
                    if (i + 2 < n_real) k4[r].z = keys[i + 2];
                    cnt4[r] = (int)min(4ll, n - i);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t kk[4] = { k4[r].x, k4[r].y, k4[r].z, k4[r].w };
#pragma unroll
            for (int j = 0; j < 4; j++) hist_add(h, kk[j], j < cnt4[r], passes);
        }
    }
    __syncthreads();
    partial[(size_t)blockIdx.x * 1024 + threadIdx.x] = h[threadIdx.x >> 8][threadIdx.x & 255];
}

// bases[p][d] = number of keys whose digit p is < d.  One block of 1024 threads per pass: four
// threads share a digit's column of the per-block partial histograms (coalesced across d, eight
// independent loads in flight each), LDS adds them up, then a block scan over d.
__global__ __launch_bounds__(1024) void os_bases_k(const uint32_t* __restrict__ partial, int n_blocks,
                                                   uint32_t* __restrict__ bases)
{
    __shared__ uint32_t part[4][256];
    __shared__ uint32_t wsum[4];
    const int p = blockIdx.x, d = threadIdx.x & 255, q = threadIdx.x >> 8;
    uint32_t s = 0;
#pragma unroll 8
    for (int b = q; b < n_blocks; b += 4) s += partial[(size_t)b * 1024 + p * 256 + d];
    part[q][d] = s;
    __syncthreads();
    s = part[0][d] + part[1][d] + part[2][d] + part[3][d];
    const int lane = lane_id(), w = d >> 6;
    const uint32_t incl = (uint32_t)wave_incl_scan((int)s);
    if (q == 0 && lane == 63) wsum[w] = incl;
    __syncthreads();
    if (q != 0) return;
    uint32_t base = 0;
    for (int k = 0; k < w; k++) base += wsum[k];
    bases[p * 256 + d] = base + incl - s;
}

// blocked 16-byte loads of one wave's share of a tile (4 x uint4 of keys, 4 x uint4 of values)
struct TileRegs {
    uint4 k[RS_ITEMS / 4], v[RS_ITEMS / 4];
};
__device__ __forceinline__ void tile_load(TileRegs& t, const uint32_t* __restrict__ keys_in,
                                          const uint32_t* __restrict__ vals_in, long long wave_base, int lane,
                                          long long n)
{
#pragma unroll
    for (int sgm = 0; sgm < RS_ITEMS / 4; sgm++) {
        const long long i = wave_base + sgm * 256 + lane * 4;
        uint4 k4 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu), v4 = make_uint4(0u, 0u, 0u, 0u);
        if (i + 3 < n) {
            k4 = *reinterpret_cast<const uint4*>(keys_in + i);
            v4 = *reinterpret_cast<const uint4*>(vals_in + i);
        } else {
            if (i < n) { k4.x = keys_in[i]; v4.x = vals_in[i]; }
            if (i + 1 < n) { k4.y = keys_in[i + 1]; v4.y = vals_in[i + 1]; }
            if (i + 2 < n) { k4.z = keys_in[i + 2]; v4.z = vals_in[i + 2]; }
        }
        t.k[sgm] = k4;
        t.v[sgm] = v4;
    }
}

// Persistent blocks; tiles are taken by atomic ticket (a tile only ever waits for tiles that already run -- two sorts
// of two streams can share the chip, so a static tile order could dead-lock their look-backs).  ONE ticket per block
// at the start (the ticket word is hot: 8 ns per atomic, and a launch of a few hundred tiles -- the broad phase's size --
// starts with every block queueing there; two tickets per block up front also left half of such a launch's blocks
// without a tile while the others did two).  The ticket of the next tile is requested at the top of a tile and used
// after its ranking: the next tile's loads are then in flight during the look-back, the staging and the writes.
__global__ __launch_bounds__(RS_THREADS) void os_pass_k(const uint32_t* __restrict__ keys_in,
                                                        const uint32_t* __restrict__ vals_in,
                                                        uint32_t* __restrict__ keys_out,
                                                        uint32_t* __restrict__ vals_out, long long n, int shift,
                                                        int num_tiles, const uint32_t* __restrict__ bases,
                                                        uint32_t* status, uint32_t* ticket, int dbg,
                                                        const uint32_t* __restrict__ d_n_real /* os_hist_k; first pass only */)
{
    const long long n_load = d_n_real ? min((long long)*d_n_real, n) : n; // keys beyond it are loaded as 0xFFFFFFFF
    __shared__ uint32_t wtot[RS_WAVES][256]; // per-wave digit totals -> offset of the wave inside the digit's run
    __shared__ uint32_t wrun[RS_WAVES][256]; // running per-wave counters of the ranking
    __shared__ uint32_t dig_excl[256];       // exclusive offset of the digit inside the tile
    __shared__ uint32_t dig_gbase[256];      // global position of the digit's first key of the tile
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[RS_TILE];
    __shared__ __attribute__((aligned(16))) uint32_t s_vals[RS_TILE];
    __shared__ uint32_t s_tk[2];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    uint32_t* wk = s_keys + w * RS_WAVE_SPAN;
    uint32_t* wv = s_vals + w * RS_WAVE_SPAN;

    // ticket == nullptr: a launch with a block per tile takes the tile from the block index -- no atomics at all on the one
    // ticket word (a few hundred of them at the start of every pass of a broad-phase sort were 7 of its 27 us).  A tile only
    // waits for LOWER tiles, and workgroups are handed to each XCD in increasing order: a lower tile is resident, or next in
    // line on its XCD, whenever a higher one spins -- also with a second sort of another stream on the chip.
    if (ticket) {
        if (threadIdx.x == 0) s_tk[0] = atomicAdd(ticket, 1u);
        __syncthreads();
    }
    uint32_t t0 = ticket ? s_tk[0] : blockIdx.x;
    TileRegs pre;
    if ((int)t0 < num_tiles) tile_load(pre, keys_in, vals_in, (long long)t0 * RS_TILE + (long long)w * RS_WAVE_SPAN, lane, n_load);

    while ((int)t0 < num_tiles) {
        const uint32_t tile = t0;
        const long long tile_base = (long long)tile * RS_TILE;
        const long long wave_base = tile_base + (long long)w * RS_WAVE_SPAN;
        const int tile_count = (int)min((long long)RS_TILE, n - tile_base);
        // 0. the prefetched blocked registers -> LDS -> striped registers (row r, lane l <-> index 64 r + l)
#pragma unroll
        for (int sgm = 0; sgm < RS_ITEMS / 4; sgm++) {
            *reinterpret_cast<uint4*>(wk + sgm * 256 + lane * 4) = pre.k[sgm];
            *reinterpret_cast<uint4*>(wv + sgm * 256 + lane * 4) = pre.v[sgm];
        }
#pragma unroll
        for (int k = 0; k < RS_WAVES * 256 / RS_THREADS; k++) { // (256 threads: thread d zeroes column d of every wave's row)
            (&wtot[0][0])[k * RS_THREADS + threadIdx.x] = 0;
            (&wrun[0][0])[k * RS_THREADS + threadIdx.x] = 0;
        }
        // request the next ticket (a launch with a block per tile has none to give: the broad phase's size, where the 400
        // wasted atomics on the one ticket word were a third of the 7 us a pass spends queueing there)
        uint32_t t1_req = (uint32_t)num_tiles;
        if (ticket && threadIdx.x == 0 && (int)gridDim.x < num_tiles) t1_req = atomicAdd(ticket, 1u);
        wave_lds_fence();
        uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
        bool valid[RS_ITEMS];
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) {
            valid[r] = wave_base + r * 64 + lane < n;
            key[r] = wk[r * 64 + lane];
            val[r] = wv[r * 64 + lane];
        }
        __syncthreads(); // counters zeroed by all, LDS tile buffers read by all
        if (OS_RANK_MATCH) { // the wave's slices of the tile buffers are free until the staging (4.): the match words of the ranking
#pragma unroll
            for (int i = 0; i < RS_WAVE_SPAN / 256; i++) {
                reinterpret_cast<uint4*>(wk)[i * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
                reinterpret_cast<uint4*>(wv)[i * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        // 1. digit counts first (LDS atomics) so the tile can publish them before the ranking
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++)
            if (valid[r]) atomicAdd(&wtot[w][(key[r] >> shift) & 255u], 1u);
        __syncthreads();
        uint32_t cnt = 0;
        const bool dig = RS_THREADS == 256 || threadIdx.x < 256; // thread d < 256 owns digit d
        uint32_t* my = status + (size_t)tile * 256 + (threadIdx.x & 255);
        {
            const int d = threadIdx.x & 255;
            uint32_t incl = 0;
            if (dig) {
#pragma unroll
                for (int k = 0; k < RS_WAVES; k++) {
                    const uint32_t t = wtot[k][d];
                    wtot[k][d] = cnt; // offset of wave k inside the digit's run
                    cnt += t;
                }
                __hip_atomic_store(my, cnt | OS_FLAG_AGG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // exclusive scan of cnt over the 256 digits -> position of the digit's run inside the tile
                incl = (uint32_t)wave_incl_scan((int)cnt);
                if (lane == 63) dig_gbase[w] = incl; // scratch
            }
            __syncthreads();
            if (dig) {
                uint32_t base = 0;
                for (int k = 0; k < w; k++) base += dig_gbase[k];
                dig_excl[d] = base + incl - cnt;
            }
        }
        // 2. ranking (the predecessors' words propagate meanwhile)
        if (dbg & 2) {
#pragma unroll
            for (int r = 0; r < RS_ITEMS; r++) rank[r] = 0;
        } else {
            if (OS_RANK_MATCH)
                wave_rank_rows_match(key, valid, shift, wrun[w], reinterpret_cast<unsigned long long*>(wk),
                                     reinterpret_cast<unsigned long long*>(wv), rank);
            else wave_rank_rows(key, valid, shift, wrun[w], rank);
        }
        if (threadIdx.x == 0) s_tk[1] = t1_req;
        __syncthreads(); // dig_gbase scratch consumed before it is overwritten below
        // the next tile's loads (consumed at the top of the next round)
        const uint32_t t1 = s_tk[1];
        if ((int)t1 < num_tiles) tile_load(pre, keys_in, vals_in, (long long)t1 * RS_TILE + (long long)w * RS_WAVE_SPAN, lane, n_load);
        // 3. decoupled look-back, thread d = digit d, eight predecessor tiles probed per round
        if (dig) {
            const int d = threadIdx.x;
            uint32_t excl = 0;
            long long t = (long long)tile - 1;
            bool done = t < 0 || (dbg & 4);
            while (!done) {
                uint32_t v[8];
#pragma unroll
                for (int j = 0; j < 8; j++)
                    v[j] = (t - j >= 0) ? __hip_atomic_load(status + (size_t)(t - j) * 256 + d, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT)
                                        : OS_FLAG_PREFIX;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (done) break;
                    if (v[j] == 0u) break; // not published yet: probe again from this tile
                    excl += v[j] & OS_VALUE_MASK;
                    if (v[j] & OS_FLAG_PREFIX) done = true;
                    --t;
                }
            }
            __hip_atomic_store(my, (excl + cnt) | OS_FLAG_PREFIX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dig_gbase[d] = bases[d] + excl;
        }
        // 4. stage the tile in LDS in sorted order (the striped copies were consumed before step 1)
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) {
            if (valid[r]) {
                const uint32_t d = (key[r] >> shift) & 255u;
                const uint32_t pos = dig_excl[d] + wtot[w][d] + rank[r];
                s_keys[pos] = key[r];
                s_vals[pos] = val[r];
            }
        }
        __syncthreads();
        // 5. consecutive threads write consecutive addresses of each digit's run
        for (int i = threadIdx.x; i < tile_count; i += RS_THREADS) {
            const uint32_t k = s_keys[i];
            const uint32_t d = (k >> shift) & 255u;
            const uint32_t dst = dig_gbase[d] + ((uint32_t)i - dig_excl[d]);
            if (dbg & 1) {
                if (dst == 0xFFFFFFFFu) keys_out[0] = k + s_vals[i];
            } else {
                keys_out[dst] = k;
                vals_out[dst] = s_vals[i];
            }
        }
        __syncthreads(); // LDS tile buffers and s_tk free for the next tile
        t0 = t1;
    }
}

} // namespace

// Sorts by the low `key_bits` bits (a multiple of 8).  Returns true when the result ended in the
// context's ping-pong buffers (odd number of passes) instead of (keys, vals).
bool radix_sort_pairs_u32(sccd_ctx* c, uint32_t* keys, uint32_t* vals, int64_t n, int key_bits, const uint32_t* d_n_real)
{
    if (n <= 1) return false;
    const int passes = std::max(1, std::min(4, (key_bits + 7) / 8));
    SCCD_REQUIRE(n < (1ll << 30), "radix sort: at most 2^30-1 elements");
    const int num_tiles = (int)((n + RS_TILE - 1) / RS_TILE);
    c->sort_tmp_keys.ensure(sizeof(uint32_t) * (size_t)n);
    c->sort_tmp_vals.ensure(sizeof(uint32_t) * (size_t)n);
    uint32_t* k_in = keys;
    uint32_t* v_in = vals;
    uint32_t* k_out = c->sort_tmp_keys.as<uint32_t>();
    uint32_t* v_out = c->sort_tmp_vals.as<uint32_t>();
    // onesweep: [4 tickets, one per 128-B line] [status passes x tiles x 256] [bases 4x256] [partial hist blocks x 1024]
    // (Per-segment tickets + bases were tried: the bases of a segment are only known for the FIRST pass,
    // later passes see permuted keys; ticket streams without a global order can deadlock the look-back; and a
    // ticket worth two consecutive tiles serialises it -- the second tile publishes its count only after the
    // first is finished, so every predecessor chain runs at one tile time per link: 70x slower.)
    const int hist_blocks = (int)std::min<long long>((n + 4 * HS_THREADS - 1) / (4 * HS_THREADS), c->num_cus); // (>= one 16-byte load per thread)
    const int dbg = 0; // (os_pass_k's timing ablations -- 1 no global writes, 2 no ranking, 4 no look-back -- are a rebuild away: tools/sortdbg.sh)
    const size_t status_bytes = (size_t)passes * num_tiles * 256 * sizeof(uint32_t);
    const size_t off_status = 512, off_bases = off_status + status_bytes, off_partial = off_bases + 4096;
    c->sort_hist.ensure(off_partial + (size_t)hist_blocks * 4096);
    char* base = c->sort_hist.as<char>();
    uint32_t* tickets = reinterpret_cast<uint32_t*>(base);
    uint32_t* bases = reinterpret_cast<uint32_t*>(base + off_bases);
    uint32_t* partial = reinterpret_cast<uint32_t*>(base + off_partial);
    uint32_t* status = reinterpret_cast<uint32_t*>(base + off_status);
    // every polled word (tickets and status, contiguous) is zeroed by os_hist_k before the passes
    hipLaunchKernelGGL(os_hist_k, dim3(hist_blocks), dim3(HS_THREADS), 0, c->stream, k_in, (long long)n, partial,
                       reinterpret_cast<uint4*>(base), (long long)((512 + status_bytes) / 16), passes, d_n_real);
    hipLaunchKernelGGL(os_bases_k, dim3(4), dim3(1024), 0, c->stream, partial, hist_blocks, bases);
    for (int pass = 0; pass < passes; pass++) {
        constexpr int pass_blocks = 2; // resident blocks per CU (103 registers x 8 waves: two per CU; 50 KB of LDS each)
        // (tiles by block index only while at most one ccd() call's two contexts are alive: the argument for it is about two
        // concurrent sorts -- common.hpp live_context_count)
        const bool force_tickets = lab_env().sort_tickets || __atomic_load_n(&live_context_count(), __ATOMIC_RELAXED) > 2;
        const int blocks = std::min(num_tiles, c->num_cus * pass_blocks);
        // (a block per tile: tiles by block index, no ticket word -- os_pass_k)
        uint32_t* const tk = (blocks == num_tiles && !force_tickets) ? nullptr : tickets + pass * 32;
        hipLaunchKernelGGL(os_pass_k, dim3(blocks), dim3(RS_THREADS), 0, c->stream, k_in,
                           v_in, k_out, v_out, (long long)n, 8 * pass, num_tiles, bases + 256 * pass,
                           status + (size_t)pass * num_tiles * 256, tk, dbg, pass == 0 ? d_n_real : nullptr);
        std::swap(k_in, k_out);
        std::swap(v_in, v_out);
    }
    SCCD_HIP(hipGetLastError());
    return (passes & 1) != 0;
}
