// sort.hip -- hand-written LSD radix sort of (uint32 key, uint32 value) pairs for gfx950.
//
// Replaces thrust::sort_by_key with a comparator on Scalar2 keys moving 48-byte MiniBox payloads
// (src/scalable_ccd/cuda/broad_phase/aabb.cu:107-109): here only an 8-byte (key, index) pair
// moves through the passes and the payload is gathered once afterwards (boxes.hip).
//
// 4 passes x 8-bit digits.  Per pass:
//   rs_count_k    per-tile digit histogram                     (read 4 B/key)
//   scan.hip      exclusive scan of the [digit][tile] table     (tiny)
//   rs_scatter_k  stable wave-level ranking + scatter           (read 8 B, write 8 B per pair)
// Ranking is wave64-native: peers of a lane = lanes with the same digit, found with 8 ballots;
// rank inside the row = popcount of the peers below the lane (mbcnt); one leader lane per
// digit bumps the wave's LDS counter.  Order inside a tile is (wave, row, lane) = index order,
// so the sort is stable.
#include "internal.hpp"

#include <algorithm>

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = 16;                           // rows of 64 keys per wave
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;         // 4096 keys per tile
constexpr int RS_WAVE_SPAN = RS_ITEMS * 64;            // keys owned by one wave

__global__ __launch_bounds__(RS_THREADS) void rs_count_k(const uint32_t* __restrict__ keys, long long n, int shift,
                                                         int num_tiles, uint32_t* __restrict__ counts)
{
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const long long i = base + (long long)r * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&hist[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    counts[(size_t)threadIdx.x * num_tiles + blockIdx.x] = hist[threadIdx.x];
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_k(const uint32_t* __restrict__ keys_in,
                                                           const uint32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ keys_out,
                                                           uint32_t* __restrict__ vals_out, long long n, int shift,
                                                           int num_tiles, const uint32_t* __restrict__ offsets)
{
    __shared__ uint32_t wcnt[RS_WAVES][256]; // per-wave digit counts, then per-wave bases
    const int lane = lane_id(), w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < RS_WAVES; k++) wcnt[k][threadIdx.x] = 0;
    __syncthreads();

    const long long wave_base = (long long)blockIdx.x * RS_TILE + (long long)w * RS_WAVE_SPAN;
    uint32_t key[RS_ITEMS];
    uint32_t rank[RS_ITEMS];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const long long i = wave_base + r * 64 + lane;
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : 0xFFFFFFFFu;
        const uint32_t d = (key[r] >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const int below = mbcnt64(peers);
        const int leader = (int)__builtin_ctzll(peers | (1ull << 63)); // lowest peer lane
        uint32_t old = 0;
        if (valid && below == 0) { // the leader of this digit group
            old = wcnt[w][d];
            wcnt[w][d] = old + (uint32_t)popc64(peers);
        }
        wave_lds_fence();
        old = (uint32_t)__shfl((int)old, leader, 64);
        rank[r] = old + (uint32_t)below;
    }
    __syncthreads();
    // thread t owns digit t: turn per-wave counts into per-wave global bases
    {
        const int d = threadIdx.x;
        uint32_t run = offsets[(size_t)d * num_tiles + blockIdx.x];
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) {
            const uint32_t t = wcnt[k][d];
            wcnt[k][d] = run;
            run += t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const long long i = wave_base + r * 64 + lane;
        if (i < n) {
            const uint32_t d = (key[r] >> shift) & 255u;
            const uint32_t pos = wcnt[w][d] + rank[r];
            keys_out[pos] = key[r];
            vals_out[pos] = vals_in[i];
        }
    }
}

} // namespace

void radix_sort_pairs_u32(sccd_ctx* c, uint32_t* keys, uint32_t* vals, int64_t n)
{
    if (n <= 1) return;
    SCCD_REQUIRE(n < (1ll << 31), "radix sort: at most 2^31-1 elements");
    const int num_tiles = (int)((n + RS_TILE - 1) / RS_TILE);
    c->sort_tmp_keys.ensure(sizeof(uint32_t) * (size_t)n);
    c->sort_tmp_vals.ensure(sizeof(uint32_t) * (size_t)n);
    c->sort_hist.ensure(sizeof(uint32_t) * 256 * (size_t)num_tiles);
    uint32_t* k_in = keys;
    uint32_t* v_in = vals;
    uint32_t* k_out = c->sort_tmp_keys.as<uint32_t>();
    uint32_t* v_out = c->sort_tmp_vals.as<uint32_t>();
    uint32_t* counts = c->sort_hist.as<uint32_t>();
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 8 * pass;
        hipLaunchKernelGGL(rs_count_k, dim3(num_tiles), dim3(RS_THREADS), 0, c->stream, k_in, (long long)n, shift,
                           num_tiles, counts);
        exclusive_scan_u32(c, counts, counts, 256 * num_tiles, reinterpret_cast<uint32_t*>(c->scalars.as<char>() + 1024));
        hipLaunchKernelGGL(rs_scatter_k, dim3(num_tiles), dim3(RS_THREADS), 0, c->stream, k_in, v_in, k_out, v_out,
                           (long long)n, shift, num_tiles, counts);
        std::swap(k_in, k_out);
        std::swap(v_in, v_out);
    }
    SCCD_HIP(hipGetLastError());
    // 4 passes: the result is back in (keys, vals)
}
