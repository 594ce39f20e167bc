// search.hpp -- binary searches over sorted uint32 arrays in HBM: per lane inside a known window, and by one whole wave
// (64 probes per round) where the window is not known yet.  Used by the record builder (boxes.hip: the first candidate
// column of every row of a two-list sweep).
#pragma once
#include "common.hpp"

#if defined(__HIPCC__)
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
#endif
