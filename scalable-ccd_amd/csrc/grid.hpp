// grid.hpp -- the composite sort key of the broad phase: (cell on the two minor axes, quantised
// coordinate on the sort axis) in ONE uint32, so the 4-pass radix sort orders all cells' sweep
// lists at once.
//
// Every map here is monotone non-decreasing in its coordinate; that is all correctness needs:
//   * boxes a, b overlap (inclusively) on a minor axis  =>  both contain the point
//     y* = max(a.min, b.min), hence both are listed in cell c(y*) = max(c(a.min), c(b.min));
//     a pair is reported only from that cell, so replication never duplicates a pair;
//   * inside one cell list sorted by q(min), a.min <= b.max  =>  q(a.min) <= q(b.max), so the
//     candidate range [.., upper_bound(q(max))) is a superset of the overlapping boxes and the
//     exact double test in the confirm stage decides.
#pragma once
#include "common.hpp"

struct GridStats { // device, filled by box_stats_k
    unsigned long long kmin[3]; // monotone u64 images of the global min (bitwise INVERTED) / max per axis
    unsigned long long kmax[3];
    double sumext[3];           // (unused: the extent sums travel as per-block partials, see box_stats_k)
};
constexpr int SCCD_MAX_CELLS = 16384;  // cells of the two-axis grid (14 of the 32 key bits at most)
constexpr int SCCD_DEFAULT_CELLS = SCCD_MAX_CELLS; // default cap (the SCCD_MAX_CELLS environment variable lowers it)
constexpr int SCCD_STATS_BLOCKS = 512; // blocks of box_stats_k per list (fixed: the partial sums must not depend on the device)

struct GridParams { // device, written by grid_setup_k
    int axis, aa, ab; // sort axis and the two minor axes
    int Sa, Sb;       // cells along aa and ab (>= 1)
    int xb;           // bits of the quantised sort coordinate (sized from range / mean extent)
    int n_cells;
    int key_bits;     // (tag bit +) cell bits + xb, a multiple of 8: the radix sort runs key_bits / 8 passes
    int tag_bit;      // two-list builds that sort both lists at once: bit key_bits - 1 marks list B; else -1
    int pad_;
    double x0, xscale, xqmax;
    double a0, inv_ha, b0, inv_hb;
};

#if defined(__HIPCC__)
__device__ __forceinline__ unsigned long long mono64(double x)
{
    x = x + 0.0;
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return b ^ ((b >> 63) ? 0xFFFFFFFFFFFFFFFFull : 0x8000000000000000ull);
}
__device__ __forceinline__ double mono64_inv(unsigned long long k)
{
    k ^= (k >> 63) ? 0x8000000000000000ull : 0xFFFFFFFFFFFFFFFFull;
    return __longlong_as_double((long long)k);
}

// quantised sort coordinate in [0, 2^xb)
__device__ __forceinline__ unsigned grid_qx(const GridParams& g, double x)
{
    double t = (x - g.x0) * g.xscale;
    if (!(t > 0.0)) t = 0.0; // also NaN
    if (t > g.xqmax) t = g.xqmax;
    return (unsigned)t;
}
__device__ __forceinline__ int grid_cell_1d(double y, double y0, double inv_h, int S)
{
    double t = (y - y0) * inv_h;
    if (!(t > 0.0)) t = 0.0;
    const double top = (double)(S - 1);
    if (t > top) t = top;
    return (int)t;
}
__device__ __forceinline__ int grid_cell_a(const GridParams& g, double y) { return grid_cell_1d(y, g.a0, g.inv_ha, g.Sa); }
__device__ __forceinline__ int grid_cell_b(const GridParams& g, double z) { return grid_cell_1d(z, g.b0, g.inv_hb, g.Sb); }
#endif
