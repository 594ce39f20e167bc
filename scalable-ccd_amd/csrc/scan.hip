// scan.hip -- device-wide exclusive prefix sum of uint32 (the "prefix-scan of interval starts"
// that places every box's cell entries before the sort).
//
// Three small kernels (reduce per tile / scan of the tile sums / scan per tile + offset), all
// wave64 shuffles + one LDS exchange per block; the table is read twice and written once
// (12 B per element).  n < 2^31, sum < 2^32.
#include "internal.hpp"

namespace {

constexpr int SC_THREADS = 256;
constexpr int SC_ITEMS = 8;
constexpr int SC_TILE = SC_THREADS * SC_ITEMS; // 2048 elements per block

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds /*[5]*/, uint32_t& block_total)
{
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const uint32_t incl = (uint32_t)wave_incl_scan((int)v);
    if (lane == 63) lds[w] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < SC_THREADS / 64; k++) {
        const uint32_t t = lds[k];
        if (k < w) base += t;
        tot += t;
    }
    block_total = tot;
    __syncthreads();
    return base + incl - v;
}

__global__ __launch_bounds__(SC_THREADS) void sc_reduce_k(const uint32_t* __restrict__ in, int n,
                                                          uint32_t* __restrict__ tile_sums)
{
    __shared__ uint32_t lds[8];
    const int base = blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++)
        if (base + k < n) s += in[base + k];
    uint32_t tot;
    block_exclusive_scan(s, lds, tot);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// one block scans up to 2^20 tile sums (2^31 elements); also stores the grand total
__global__ __launch_bounds__(1024) void sc_tiles_k(uint32_t* __restrict__ tile_sums, int n_tiles,
                                                   uint32_t* __restrict__ total_out)
{
    __shared__ uint32_t wsum[16];
    const int per = (n_tiles + 1023) / 1024;
    const int beg = min(n_tiles, (int)threadIdx.x * per), end = min(n_tiles, beg + per);
    uint32_t s = 0;
    for (int i = beg; i < end; i++) s += tile_sums[i];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const uint32_t incl = (uint32_t)wave_incl_scan((int)s);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t t = wsum[k];
        if (k < w) base += t;
        tot += t;
    }
    uint32_t run = base + incl - s;
    for (int i = beg; i < end; i++) {
        const uint32_t t = tile_sums[i];
        tile_sums[i] = run;
        run += t;
    }
    if (threadIdx.x == 0) *total_out = tot;
}

__global__ __launch_bounds__(SC_THREADS) void sc_scan_k(const uint32_t* __restrict__ in, int n,
                                                        const uint32_t* __restrict__ tile_offsets,
                                                        uint32_t* __restrict__ out)
{
    __shared__ uint32_t lds[8];
    const int base = blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS;
    uint32_t v[SC_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        s += v[k];
    }
    uint32_t tot;
    uint32_t run = block_exclusive_scan(s, lds, tot) + tile_offsets[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}

} // namespace

// out[i] = sum_{j<i} in[j]; *d_total = sum of all (device pointer).  in == out allowed.
void exclusive_scan_u32(sccd_ctx* c, const uint32_t* in, uint32_t* out, int n, uint32_t* d_total)
{
    if (n <= 0) {
        SCCD_HIP(hipMemsetAsync(d_total, 0, sizeof(uint32_t), c->stream));
        return;
    }
    const int n_tiles = (n + SC_TILE - 1) / SC_TILE;
    c->sort_status.ensure(sizeof(uint32_t) * (size_t)n_tiles);
    uint32_t* tiles = c->sort_status.as<uint32_t>();
    hipLaunchKernelGGL(sc_reduce_k, dim3(n_tiles), dim3(SC_THREADS), 0, c->stream, in, n, tiles);
    hipLaunchKernelGGL(sc_tiles_k, dim3(1), dim3(1024), 0, c->stream, tiles, n_tiles, d_total);
    hipLaunchKernelGGL(sc_scan_k, dim3(n_tiles), dim3(SC_THREADS), 0, c->stream, in, n, tiles, out);
    SCCD_HIP(hipGetLastError());
}
