// boxes.hip -- box construction and the split/gather around the sort, on the device.
//
// Replaces (reference paths relative to the reference root):
//   AABB::conservative_inflation / from_point      src/scalable_ccd/cuda/broad_phase/aabb.cu:19-37
//   build_vertex/edge/face_boxes (host, TBB)       aabb.cu:115-229   (CPU twin: broad_phase/aabb.cpp:38-133)
//   split_boxes + the payload movement of the sort aabb.cu:40-111
// The reference builds boxes on the host and copies 64 B/box to the device; here the mesh is
// already resident and the boxes never leave HBM.
#include "internal.hpp"
#include "ti_math_f32.hpp" // nextafter_up_f / nextafter_down_f of the float build
#include "grid.hpp"
#include "search.hpp"

#include <algorithm>

#define TI_INF __builtin_huge_val()

namespace {

constexpr int TPB = 256;
inline int grid_for(long long n) { return (int)((n + TPB - 1) / TPB); }

// column-major V0,V1 (Eigen) -> packed {x0,y0,z0,x1,y1,z1}
__global__ void pack_vertices_k(const double* __restrict__ V0, const double* __restrict__ V1, int nV,
                                double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nV) return;
    double2* o = reinterpret_cast<double2*>(out) + 3 * (size_t)i;
    o[0] = make_double2(V0[i], V0[i + (size_t)nV]);
    o[1] = make_double2(V0[i + 2 * (size_t)nV], V1[i]);
    o[2] = make_double2(V1[i + (size_t)nV], V1[i + 2 * (size_t)nV]);
}

// The index matrices are validated HERE, on the device, while they are packed (the reference asserts nothing and would
// fault): an index outside [0, nV) raises *bad and is stored CLAMPED, so no later gather can go wild even before the
// caller has seen the error.  (Round 2 scanned both matrices on the host first: 1.8 ms of a 5.6 ms ccd() from host
// matrices on the 1M-triangle cloth.)
__device__ __forceinline__ int checked_index(int v, int nV, unsigned* __restrict__ bad)
{
    if ((unsigned)v >= (unsigned)nV) {
        atomicOr(bad, 1u);
        return 0;
    }
    return v;
}
__global__ void pack_edges_k(const int* __restrict__ E, int nE, int nV, int2* __restrict__ out, unsigned* __restrict__ bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nE) return;
    out[i] = make_int2(checked_index(E[i], nV, bad), checked_index(E[i + (size_t)nE], nV, bad));
}

__global__ void pack_faces_k(const int* __restrict__ F, int nF, int nV, int4* __restrict__ out, unsigned* __restrict__ bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nF) return;
    out[i] = make_int4(checked_index(F[i], nV, bad), checked_index(F[i + (size_t)nF], nV, bad),
                       checked_index(F[i + 2 * (size_t)nF], nV, bad), 0);
}

__device__ __forceinline__ void store_box(sccd_aabb* out, const double lo[3], const double hi[3], int v0, int v1,
                                          int v2, int eid)
{
    double4* o = reinterpret_cast<double4*>(out);
    o[0] = make_double4(lo[0], lo[1], lo[2], hi[0]);
    double2* o2 = reinterpret_cast<double2*>(out) + 2;
    o2[0] = make_double2(hi[1], hi[2]);
    int4* oi = reinterpret_cast<int4*>(out) + 3;
    oi[0] = make_int4(v0, v1, v2, eid);
}

// bounds + summed extents of the boxes a thread has produced / read; finish() reduces over the
// block and publishes: exact integer atomics for the bounds, ONE partial per block for the sums
// (added up in a fixed order by grid_setup_k: every rank of a multi-GPU run must derive
// bit-identical grid parameters, which a floating-point atomicAdd would not guarantee)
struct StatsAcc {
    double lo[3], hi[3], se[3];
    __device__ __forceinline__ StatsAcc()
    {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            lo[k] = TI_INF;
            hi[k] = -TI_INF;
            se[k] = 0.0;
        }
    }
    __device__ __forceinline__ void add(const double (&l)[3], const double (&h)[3])
    {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            lo[k] = fmin(lo[k], l[k]);
            hi[k] = fmax(hi[k], h[k]);
            se[k] += h[k] - l[k];
        }
    }
    // every thread of the block must call this
    __device__ __forceinline__ void finish(GridStats* __restrict__ st, double* __restrict__ part) { finish(st, part, (int)blockIdx.x); }
    // block: this block's index among the blocks that build THIS list
    __device__ __forceinline__ void finish(GridStats* __restrict__ st, double* __restrict__ part, int block)
    {
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lo[k] = fmin(lo[k], __shfl_xor(lo[k], o, 64));
                hi[k] = fmax(hi[k], __shfl_xor(hi[k], o, 64));
                se[k] += __shfl_xor(se[k], o, 64);
            }
        }
        __shared__ double red[TPB / 64][9];
        const int w = threadIdx.x >> 6;
        if (lane_id() == 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                red[w][k] = lo[k];
                red[w][3 + k] = hi[k];
                red[w][6 + k] = se[k];
            }
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            const int k = threadIdx.x;
            double l = red[0][k], h = red[0][3 + k], s = red[0][6 + k];
#pragma unroll
            for (int j = 1; j < TPB / 64; j++) {
                l = fmin(l, red[j][k]);
                h = fmax(h, red[j][3 + k]);
                s += red[j][6 + k];
            }
            // one partial per block, no atomics (nine hot words shared by every block cost more than the
            // reduction grid_setup_k does instead)
            (void)st;
            part[block * 9 + k] = l;
            part[block * 9 + 3 + k] = h;
            part[block * 9 + 6 + k] = s;
        }
    }
};

// AABB::from_point(p_t0, p_t1, r): per coordinate
//   lo = min(nextafter_down(p0) - nextafter_up(r), nextafter_down(p1) - nextafter_up(r))
//   hi = max(nextafter_up(p0)   + nextafter_up(r), nextafter_up(p1)   + nextafter_up(r))
// (aabb.cu:19-37, aabb.cuh:55-63; ids aabb.cu:180-181)
// f32: the reference's float build -- vertices and radius cast to float FIRST (aabb.cpp:43-47, aabb.cu:124-128),
// nextafterf and float sums; the float results are stored widened (every later comparison is exact on them)
template <bool F32>
__global__ void vertex_boxes_k(const double* __restrict__ V, int nV, double r, sccd_aabb* __restrict__ out,
                               GridStats* __restrict__ st, double* __restrict__ part, unsigned long long* __restrict__ t_first)
{
    // t_first (may be null; host-coherent pinned memory): the device's real-time clock as the FIRST kernel of a ccd() step starts --
    // the step's other end is stamped by the kernel behind its last read-back (api.hip readback_gather_k): sccd_ctx::step_stamp
    if (t_first && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(t_first, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    StatsAcc acc;
    const double ru = nextafter_up(r);
    const float ru_f = nextafter_up_f((float)r);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nV; i += gridDim.x * blockDim.x) {
        const double2* v = reinterpret_cast<const double2*>(V) + 3 * (size_t)i;
        const double2 a = v[0], b = v[1], c2 = v[2];
        const double p0[3] = { a.x, a.y, b.x }, p1[3] = { b.y, c2.x, c2.y };
        double lo[3], hi[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (F32) {
                const float q0 = (float)p0[k], q1 = (float)p1[k];
                const float l0 = nextafter_down_f(q0) - ru_f, l1 = nextafter_down_f(q1) - ru_f;
                const float h0 = nextafter_up_f(q0) + ru_f, h1 = nextafter_up_f(q1) + ru_f;
                lo[k] = (l1 < l0) ? l1 : l0;
                hi[k] = (h0 < h1) ? h1 : h0;
                continue;
            }
            const double l0 = nextafter_down(p0[k]) - ru, l1 = nextafter_down(p1[k]) - ru;
            const double h0 = nextafter_up(p0[k]) + ru, h1 = nextafter_up(p1[k]) + ru;
            lo[k] = (l1 < l0) ? l1 : l0;
            hi[k] = (h0 < h1) ? h1 : h0;
        }
        store_box(out + i, lo, hi, i, -i - 1, -i - 1, i);
        acc.add(lo, hi);
    }
    if (st) acc.finish(st, part);
}

__device__ __forceinline__ double sel3d(const double (&v)[3], int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); }

struct BoxLoad {
    double lo[3], hi[3];
};
__device__ __forceinline__ BoxLoad load_box_geom(const sccd_aabb* b)
{
    const double4 q = reinterpret_cast<const double4*>(b)[0];
    const double2 q2 = reinterpret_cast<const double2*>(b)[2];
    BoxLoad r;
    r.lo[0] = q.x;
    r.lo[1] = q.y;
    r.lo[2] = q.z;
    r.hi[0] = q.w;
    r.hi[1] = q2.x;
    r.hi[2] = q2.y;
    return r;
}

// Where a list's boxes come from: the raw array (SRC 0), or -- multi-GPU: a rank needs the boxes of ITS window of cells only,
// and writing all 3 M boxes of a mesh on every rank was the largest item of an 8-rank step -- computed on the spot from the
// vertex boxes and the element's vertex indices (SRC 1 edges, SRC 2 faces: the arithmetic of edge_boxes_body /
// face_boxes_body, hence the same bits), and stored to `out` only where the caller says so.
struct BoxSrc {
    const sccd_aabb* raw; // SRC 0
    const sccd_aabb* vb;  // SRC 1, 2: the vertex boxes
    const void* elems;    // int2[] edges / int4[] faces
    sccd_aabb* out;       // where a computed box goes if it is kept (element order)
};
template <int SRC> __device__ __forceinline__ BoxLoad src_box(const BoxSrc& bs, int i, int4* ids)
{
    if (SRC == 0) return load_box_geom(bs.raw + i);
    BoxLoad r;
    if (SRC == 1) {
        const int2 e = reinterpret_cast<const int2*>(bs.elems)[i];
        const BoxLoad a = load_box_geom(bs.vb + e.x), b = load_box_geom(bs.vb + e.y);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            r.lo[k] = (b.lo[k] < a.lo[k]) ? b.lo[k] : a.lo[k];
            r.hi[k] = (a.hi[k] < b.hi[k]) ? b.hi[k] : a.hi[k];
        }
        *ids = make_int4(e.x, e.y, -e.x - 1, i);
    } else {
        const int4 f = reinterpret_cast<const int4*>(bs.elems)[i];
        const BoxLoad a = load_box_geom(bs.vb + f.x), b = load_box_geom(bs.vb + f.y), c = load_box_geom(bs.vb + f.z);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double l = (b.lo[k] < a.lo[k]) ? b.lo[k] : a.lo[k];
            l = (c.lo[k] < l) ? c.lo[k] : l;
            double h = (a.hi[k] < b.hi[k]) ? b.hi[k] : a.hi[k];
            h = (h < c.hi[k]) ? c.hi[k] : h;
            r.lo[k] = l;
            r.hi[k] = h;
        }
        *ids = make_int4(f.x, f.y, f.z, i);
    }
    return r;
}

// AABB(a, b): component-wise min/max (aabb.cuh:18-29); ids aabb.cu:200-203
__device__ __forceinline__ void edge_boxes_body(const sccd_aabb* __restrict__ vb, const int2* __restrict__ E, int nE,
                                                sccd_aabb* __restrict__ out, GridStats* __restrict__ st, double* __restrict__ part,
                                                int block, int n_blocks)
{
    StatsAcc acc;
    for (int i = block * (int)blockDim.x + (int)threadIdx.x; i < nE; i += n_blocks * (int)blockDim.x) {
        const int2 e = E[i];
        const BoxLoad a = load_box_geom(vb + e.x), b = load_box_geom(vb + e.y);
        double lo[3], hi[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            lo[k] = (b.lo[k] < a.lo[k]) ? b.lo[k] : a.lo[k];
            hi[k] = (a.hi[k] < b.hi[k]) ? b.hi[k] : a.hi[k];
        }
        store_box(out + i, lo, hi, e.x, e.y, -e.x - 1, i);
        acc.add(lo, hi);
    }
    if (st) acc.finish(st, part, block);
}
__global__ void edge_boxes_k(const sccd_aabb* __restrict__ vb, const int2* __restrict__ E, int nE,
                             sccd_aabb* __restrict__ out, GridStats* __restrict__ st, double* __restrict__ part)
{
    edge_boxes_body(vb, E, nE, out, st, part, (int)blockIdx.x, (int)gridDim.x);
}

// AABB(a, b, c) (aabb.cuh:31-42); ids aabb.cu:223-225
__device__ __forceinline__ void face_boxes_body(const sccd_aabb* __restrict__ vb, const int4* __restrict__ F, int nF,
                                                sccd_aabb* __restrict__ out, GridStats* __restrict__ st, double* __restrict__ part,
                                                int block, int n_blocks)
{
    StatsAcc acc;
    for (int i = block * (int)blockDim.x + (int)threadIdx.x; i < nF; i += n_blocks * (int)blockDim.x) {
        const int4 f = F[i];
        const BoxLoad a = load_box_geom(vb + f.x), b = load_box_geom(vb + f.y), c = load_box_geom(vb + f.z);
        double lo[3], hi[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double l = (b.lo[k] < a.lo[k]) ? b.lo[k] : a.lo[k];
            l = (c.lo[k] < l) ? c.lo[k] : l;
            double h = (a.hi[k] < b.hi[k]) ? b.hi[k] : a.hi[k];
            h = (h < c.hi[k]) ? c.hi[k] : h;
            lo[k] = l;
            hi[k] = h;
        }
        store_box(out + i, lo, hi, f.x, f.y, f.z, i);
        acc.add(lo, hi);
    }
    if (st) acc.finish(st, part, block);
}
__global__ void face_boxes_k(const sccd_aabb* __restrict__ vb, const int4* __restrict__ F, int nF,
                             sccd_aabb* __restrict__ out, GridStats* __restrict__ st, double* __restrict__ part)
{
    face_boxes_body(vb, F, nF, out, st, part, (int)blockIdx.x, (int)gridDim.x);
}
// the edge and the face boxes of a mesh in ONE launch (both depend on the vertex boxes only; the first blocks_e blocks build
// the edges': ccd() builds all three lists at the head of every step)
__global__ void edge_face_boxes_k(const sccd_aabb* __restrict__ vb, const int2* __restrict__ E, int nE, sccd_aabb* __restrict__ out_e,
                                  GridStats* __restrict__ st_e, double* __restrict__ part_e, int blocks_e,
                                  const int4* __restrict__ F, int nF, sccd_aabb* __restrict__ out_f, GridStats* __restrict__ st_f,
                                  double* __restrict__ part_f)
{
    if ((int)blockIdx.x < blocks_e) edge_boxes_body(vb, E, nE, out_e, st_e, part_e, (int)blockIdx.x, blocks_e);
    else face_boxes_body(vb, F, nF, out_f, st_f, part_f, (int)blockIdx.x - blocks_e, (int)gridDim.x - blocks_e);
}

// ---- composite key: cell on the minor axes + quantised sort coordinate (grid.hpp) ------------

// Bounds and summed extents of a box list that was NOT built here (uploaded boxes): same outputs
// as the builders' fused statistics.
__global__ void box_stats_k(const sccd_aabb* __restrict__ raw, int n, GridStats* __restrict__ st,
                            double* __restrict__ part /* [gridDim.x][3] */)
{
    StatsAcc acc;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const BoxLoad b = load_box_geom(raw + i);
        acc.add(b.lo, b.hi);
    }
    acc.finish(st, part);
}

// Bounds and summed extents of an element list from every `stride`-th element, WITHOUT building the list (multi-GPU: a
// rank builds the boxes of its window only; the grid needs one cell size everyone agrees on, and any cell size is correct --
// the same sample on every rank gives the same grid).  The sums are scaled by the stride: what grid_setup_k divides by the
// true number of boxes.
template <int SRC>
__global__ void elem_stats_k(BoxSrc bs, int n, int stride, GridStats* __restrict__ st, double* __restrict__ part)
{
    StatsAcc acc;
    for (long long i = (long long)(blockIdx.x * blockDim.x + threadIdx.x) * stride; i < n;
         i += (long long)gridDim.x * blockDim.x * stride) {
        int4 ids;
        const BoxLoad b = src_box<SRC>(bs, (int)i, &ids);
        acc.add(b.lo, b.hi);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) acc.se[k] *= (double)stride;
    acc.finish(st, part);
}

// both lazy lists of a mesh (edges: the first blocks_e blocks; faces) in ONE launch -- a rank of a multi-GPU job runs a build chain of
// 10-17 us kernels, and two of them in a row for the statistics were 12 us of it.  The same partials as two launches of elem_stats_k:
// a list's blocks stride over it by THEIR number, and write partials by their index among them.
__global__ void elem_stats2_k(BoxSrc be, int ne, int blocks_e, BoxSrc bf, int nf, int stride, GridStats* __restrict__ st_e, double* __restrict__ part_e,
                              GridStats* __restrict__ st_f, double* __restrict__ part_f)
{
    const bool edges = (int)blockIdx.x < blocks_e; // (block-uniform)
    const int block = edges ? (int)blockIdx.x : (int)blockIdx.x - blocks_e, blocks = edges ? blocks_e : (int)gridDim.x - blocks_e;
    const int n = edges ? ne : nf;
    StatsAcc acc;
    for (long long i = (long long)(block * (int)blockDim.x + (int)threadIdx.x) * stride; i < n; i += (long long)blocks * blockDim.x * stride) {
        int4 ids;
        const BoxLoad b = edges ? src_box<1>(be, (int)i, &ids) : src_box<2>(bf, (int)i, &ids);
        acc.add(b.lo, b.hi);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) acc.se[k] *= (double)stride;
    acc.finish(edges ? st_e : st_f, edges ? part_e : part_f, block);
}

// cell size = cell_factor x mean box extent on that axis; at most 2^10 cells in total
__global__ __launch_bounds__(SCCD_STATS_BLOCKS) void grid_setup_k(const GridStats* __restrict__ st_a, const double* __restrict__ part_a, int n_part_a,
                             const GridStats* __restrict__ st_b, const double* __restrict__ part_b, int n_part_b,
                             int n_total, int axis, double cell_factor, int shrink, GridParams* __restrict__ g,
                             uint32_t* __restrict__ cursors /* the two list cursors of the fill pass: zeroed here */,
                             int max_cells, int reserve_tag, uint32_t* __restrict__ zero_hist /* [SCCD_MAX_CELLS] or null */,
                             unsigned long long* __restrict__ sweep_cnt /* the context's SweepCounters: zeroed here */, int sweep_words,
                             unsigned long long* __restrict__ narrow_cnt /* its NarrowCounters or null: {toi, zeros} */, int narrow_words,
                             unsigned long long toi_bits, int peer_at, unsigned long long peer_bits)
{
    // The counters of the launches BEHIND this build start here: the sweep's (a fill kernel between record pass and sweep
    // was 8 us on the critical path of every chain) and, for ccd(), the narrow phase's (an upload there, another 7 us).
    for (int k = threadIdx.x; k < sweep_words; k += SCCD_STATS_BLOCKS) sweep_cnt[k] = 0ull;
    if (narrow_cnt)
        for (int k = threadIdx.x; k < narrow_words; k += SCCD_STATS_BLOCKS) narrow_cnt[k] = k == 0 ? toi_bits : (k == peer_at ? peer_bits : 0ull);
    if (threadIdx.x < 4) cursors[threadIdx.x] = 0u; // the two list totals, the placement cursor of a merged two-list fill, list A's extent
    if (zero_hist) // (multi-GPU: the sampled cell histogram the next launch adds to -- a memset of its own was a launch more)
        for (int k = threadIdx.x; k < SCCD_MAX_CELLS; k += SCCD_STATS_BLOCKS) zero_hist[k] = 0u;
    // Bounds and summed extents of both lists from the builders' block partials: thread j takes partial j of list A and of
    // list B (18 loads in flight, ONE memory round trip -- the kernel is on the critical path of every build and used to
    // make 72 of them in a row), then a tree in LDS that adds in a FIXED order: the same bits on every run and every rank
    // (a multi-GPU run needs identical grid parameters everywhere, which floating-point atomics would not give).
    static_assert(SCCD_STATS_BLOCKS == 512, "grid_setup_k: one thread per block partial");
    __shared__ double red[SCCD_STATS_BLOCKS][9];
    {
        const int j = threadIdx.x;
        double v[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const double dflt = k < 3 ? TI_INF : (k < 6 ? -TI_INF : 0.0);
            const double a = (j < n_part_a) ? part_a[j * 9 + k] : dflt;
            const double bb = (j < n_part_b) ? part_b[j * 9 + k] : dflt;
            v[k] = k < 3 ? fmin(a, bb) : (k < 6 ? fmax(a, bb) : a + bb);
        }
#pragma unroll
        for (int k = 0; k < 9; k++) red[j][k] = v[k];
        __syncthreads();
        for (int half = SCCD_STATS_BLOCKS / 2; half > 0; half >>= 1) {
            if (j < half) {
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const double x = red[j][k], y = red[j + half][k];
                    red[j][k] = k < 3 ? fmin(x, y) : (k < 6 ? fmax(x, y) : x + y);
                }
            }
            __syncthreads();
        }
    }
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double glo[3] = { red[0][0], red[0][1], red[0][2] }, ghi[3] = { red[0][3], red[0][4], red[0][5] };
    const double sumext[3] = { red[0][6], red[0][7], red[0][8] };
    const int aa = (axis == 0) ? 1 : 0, ab = (axis == 2) ? 1 : 2;
    double lo[3], hi[3];
    (void)st_a;
    (void)st_b;
    for (int k = 0; k < 3; k++) {
        lo[k] = glo[k] + 0.0; // (-0 -> +0, as the monotone integer image used to do)
        hi[k] = ghi[k] + 0.0;
    }
    int S[2];
    const int ax2[2] = { aa, ab };
    for (int t = 0; t < 2; t++) {
        const int k = ax2[t];
        const double range = hi[k] - lo[k];
        const double mean = n_total > 0 ? sumext[k] / (double)n_total : 0.0;
        const double h = cell_factor * mean;
        double s = (h > 0.0 && range > 0.0 && range < TI_INF) ? floor(range / h) : 1.0;
        if (!(s >= 1.0)) s = 1.0;
        if (s > (double)SCCD_MAX_CELLS) s = (double)SCCD_MAX_CELLS;
        S[t] = (int)s;
    }
    while ((long long)S[0] * S[1] > max_cells) { // cap the number of cells, shrinking the finer axis
        if (S[0] >= S[1]) S[0] = (S[0] + 1) / 2;
        else S[1] = (S[1] + 1) / 2;
    }
    for (int t = 0; t < shrink; t++) { // replication blew up: coarser cells
        S[0] = (S[0] + 1) / 2;
        S[1] = (S[1] + 1) / 2;
    }
    if (S[0] < 1) S[0] = 1;
    if (S[1] < 1) S[1] = 1;
    int cb = 0;
    while ((1 << cb) < S[0] * S[1]) cb++;
    g->axis = axis;
    g->aa = aa;
    g->ab = ab;
    g->Sa = S[0];
    g->Sb = S[1];
    // bits of the quantised sort coordinate: 16 quanta per mean box extent keep the candidate
    // ranges tight; the total key width is rounded up to whole 8-bit radix passes (<= 32 bits)
    const double xr = hi[axis] - lo[axis];
    const double xmean = n_total > 0 ? (axis == 0 ? sumext[0] : (axis == 1 ? sumext[1] : sumext[2])) / (double)n_total : 0.0;
    int need = 32;
    if (xr > 0.0 && xr < TI_INF && xmean > 0.0) {
        double ratio = 16.0 * xr / xmean;
        need = 1;
        while (need < 32 && (double)(1ull << need) < ratio) need++;
    }
    // (reserve_tag: one more bit on top marks the entries of list B, so that ONE sort orders both lists and
    // leaves them behind each other.  If that bit alone would cost a whole radix pass, the sort coordinate gives one up.)
    if (reserve_tag && need > 8 && (cb + need) % 8 == 0) need -= 1;
    int total = cb + need + reserve_tag;
    total = ((total + 7) / 8) * 8;
    if (total > 32) total = 32;
    if (total < 8) total = 8;
    const int xb = total - reserve_tag - cb;
    g->xb = xb;
    g->n_cells = S[0] * S[1];
    g->key_bits = total;
    g->tag_bit = reserve_tag ? total - 1 : -1;
    g->pad_ = 0;
    const double qmax = (double)((1ull << xb) - 1ull);
    g->x0 = lo[axis];
    g->xscale = (xr > 0.0 && xr < TI_INF) ? qmax / xr : 0.0;
    g->xqmax = qmax;
    const double ra = hi[aa] - lo[aa], rb = hi[ab] - lo[ab];
    g->a0 = lo[aa];
    g->inv_ha = (S[0] > 1 && ra > 0.0) ? (double)S[0] / ra : 0.0;
    g->b0 = lo[ab];
    g->inv_hb = (S[1] > 1 && rb > 0.0) ? (double)S[1] / rb : 0.0;
}

struct CellSpan {
    int a0, a1, b0, b1;
};
__device__ __forceinline__ CellSpan cell_span(const GridParams& g, const BoxLoad& b)
{
    CellSpan s;
    // (sel3d, not b.lo[g.aa]: an array indexed by a run-time value lives in scratch memory -- 56 bytes per thread of it,
    // stored and re-read for every box, in the two-list fill until round 3)
    s.a0 = grid_cell_a(g, sel3d(b.lo, g.aa));
    s.a1 = grid_cell_a(g, sel3d(b.hi, g.aa));
    s.b0 = grid_cell_b(g, sel3d(b.lo, g.ab));
    s.b1 = grid_cell_b(g, sel3d(b.hi, g.ab));
    return s;
}

// number of cells of [cell_lo, cell_hi) each box overlaps (the interval starts the prefix scan
// turns into offsets).  The cell window is this rank's shard (all cells on one GPU).
__global__ void cell_count_k(const sccd_aabb* __restrict__ raw, int n, const GridParams* __restrict__ gp,
                             int cell_lo, int cell_hi, uint32_t* __restrict__ counts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GridParams g = *gp;
    const CellSpan s = cell_span(g, load_box_geom(raw + i));
    uint32_t cnt = 0;
    if (cell_lo <= 0 && cell_hi >= g.n_cells) {
        cnt = (uint32_t)((s.a1 - s.a0 + 1) * (s.b1 - s.b0 + 1));
    } else {
        for (int ca = s.a0; ca <= s.a1; ca++) {
            const int c0 = max(ca * g.Sb + s.b0, cell_lo), c1 = min(ca * g.Sb + s.b1, cell_hi - 1);
            cnt += c1 >= c0 ? (uint32_t)(c1 - c0 + 1) : 0u;
        }
    }
    counts[i] = cnt;
}

// split_boxes, key part (aabb.cu:40-72): one (key, box index) entry per overlapped cell of the window
__global__ void cell_fill_k(const sccd_aabb* __restrict__ raw, int n, const GridParams* __restrict__ gp,
                            int cell_lo, int cell_hi, const uint32_t* __restrict__ offsets,
                            uint32_t* __restrict__ key, uint32_t* __restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GridParams g = *gp;
    const BoxLoad b = load_box_geom(raw + i);
    const CellSpan s = cell_span(g, b);
    const unsigned q = grid_qx(g, sel3d(b.lo, g.axis));
    uint32_t at = offsets[i];
    for (int ca = s.a0; ca <= s.a1; ca++)
        for (int cb = s.b0; cb <= s.b1; cb++) {
            const int cell = ca * g.Sb + cb;
            if (cell < cell_lo || cell >= cell_hi) continue;
            key[at] = (uint32_t)(((unsigned long long)cell << g.xb) | q); // xb may be 32
            idx[at] = (uint32_t)i;
            ++at;
        }
}

// Entries per cell, from every `stride`-th box (multi-GPU: the ranks take contiguous cell windows
// of about equal entry counts; ANY partition of the cells is correct, so a sample is enough --
// it only has to be the same sample on every rank).
// A block takes a CONTIGUOUS run of the samples (elements that follow each other in a mesh lie in a few cells: the flush at
// the end adds only the bins the block touched) and there are enough blocks for one sample per thread: round 2's 64 blocks
// with samples strided over the grid walked eight dependent gathers per thread -- 23-27 us of an 8-rank step's 530.
template <int SRC>
__device__ __forceinline__ void cell_hist_body(uint32_t* h, const BoxSrc& bs, int n, const GridParams& g, int stride,
                                               uint32_t* __restrict__ hist, int block, int n_blocks)
{
    for (int k = threadIdx.x; k < g.n_cells; k += blockDim.x) h[k] = 0;
    __syncthreads();
    const long long n_s = ((long long)n + stride - 1) / stride;
    const long long per = (n_s + n_blocks - 1) / n_blocks;
    const long long s0 = (long long)block * per, s1 = s0 + per < n_s ? s0 + per : n_s;
    for (long long sm = s0 + threadIdx.x; sm < s1; sm += blockDim.x) {
        int4 ids;
        const CellSpan s = cell_span(g, src_box<SRC>(bs, (int)(sm * stride), &ids));
        for (int ca = s.a0; ca <= s.a1; ca++)
            for (int cb = s.b0; cb <= s.b1; cb++) atomicAdd(&h[ca * g.Sb + cb], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < g.n_cells; k += blockDim.x)
        if (h[k]) atomicAdd(&hist[k], h[k]);
}
template <int SRC>
__global__ void cell_hist_k(BoxSrc bs, int n, const GridParams* __restrict__ gp, int stride, uint32_t* __restrict__ hist /*[n_cells]*/)
{
    __shared__ uint32_t h[SCCD_MAX_CELLS];
    cell_hist_body<SRC>(h, bs, n, *gp, stride, hist, (int)blockIdx.x, (int)gridDim.x);
}
// both lists of a two-list build in one launch (the first blocks_a blocks: list A)
template <int SRC_A, int SRC_B>
__global__ void cell_hist2_k(BoxSrc a, int na, BoxSrc b, int nb, int blocks_a, const GridParams* __restrict__ gp, int stride,
                             uint32_t* __restrict__ hist)
{
    __shared__ uint32_t h[SCCD_MAX_CELLS];
    if ((int)blockIdx.x < blocks_a) cell_hist_body<SRC_A>(h, a, na, *gp, stride, hist, (int)blockIdx.x, blocks_a);
    else cell_hist_body<SRC_B>(h, b, nb, *gp, stride, hist, (int)blockIdx.x - blocks_a, (int)gridDim.x - blocks_a);
}

// Count + fill in ONE pass for a cell window (multi-GPU): a block of 1024 threads adds up the
// entries of its boxes, reserves room with ONE atomic (the cursor is a single hot word: ~90
// atomics/us chip-wide, so one per wave made the pass atomic-bound) and writes them.  The order of the entries depends on the
// order the waves arrive in -- the radix sort that follows is stable, so only the order of EQUAL
// keys (and with it the order, not the set, of the emitted pairs) varies.  Entries beyond
// `capacity` are counted but not written (the host grows the buffers and runs the pass again).
// place != nullptr: the entries go where a SHARED cursor says (both lists of a merged two-list sort fill one buffer, in any
// order: the sort that follows separates them by the tag bit) while `cursor` only counts this list's entries.
constexpr int FILL_PER = 4;                // boxes per thread of the fill pass
constexpr int FILL_BOXES = 1024 * FILL_PER; // ... per block of 1024 threads
// d_win != nullptr: the cell window comes from device memory (shard_window_k wrote it: no host round trip in between)
template <int SRC>
__device__ __forceinline__ void cell_fill_append_body(const BoxSrc& bs, int n, const GridParams* __restrict__ gp,
                                                      int cell_lo, int cell_hi, uint32_t* __restrict__ cursor, uint32_t capacity,
                                                      uint32_t* __restrict__ key, uint32_t* __restrict__ idx, int tagged,
                                                      uint32_t* __restrict__ place, int block, const ShardWindow* __restrict__ d_win = nullptr,
                                                      uint32_t* __restrict__ extq_out = nullptr)
{
    // extq_out (may be null): receives the list's largest extent along the sort axis in QUANTISED units, max over the boxes of
    // xq(hi) - xq(lo) -- what a one-class two-list sweep extends a row's window backwards by (entry_record_body)
    if (d_win) {
        cell_lo = d_win->cell_lo;
        cell_hi = d_win->cell_hi;
    }
    // FILL_PER boxes per thread (box j of the block's 4096 at thread j % 1024: consecutive threads, consecutive boxes): the
    // list cursor is ONE hot word -- ~90 atomics/us chip-wide -- and a block per 1024 boxes made the pass atomic-bound
    // (1,465 blocks x 2 atomics for the 1M-triangle cloth's faces and vertices: 33 us of a 75 us kernel)
    const GridParams g = *gp;
    const uint32_t tag = (tagged && g.tag_bit >= 0) ? (1u << g.tag_bit) : 0u; // list B of a merged two-list sort
    CellSpan s[FILL_PER];
    unsigned q[FILL_PER];
    uint32_t cnt[FILL_PER], mine = 0, extq = 0;
#pragma unroll
    for (int k = 0; k < FILL_PER; k++) {
        const int i = block * FILL_BOXES + k * (int)blockDim.x + (int)threadIdx.x;
        s[k] = CellSpan { 0, -1, 0, -1 };
        q[k] = 0;
        cnt[k] = 0;
        if (i < n) {
            int4 ids = make_int4(0, 0, 0, 0);
            const BoxLoad b = src_box<SRC>(bs, i, &ids);
            s[k] = cell_span(g, b);
            q[k] = grid_qx(g, sel3d(b.lo, g.axis));
            if (extq_out) extq = max(extq, grid_qx(g, sel3d(b.hi, g.axis)) - q[k]);
            for (int ca = s[k].a0; ca <= s[k].a1; ca++) {
                const int c0 = max(ca * g.Sb + s[k].b0, cell_lo), c1 = min(ca * g.Sb + s[k].b1, cell_hi - 1);
                cnt[k] += c1 >= c0 ? (uint32_t)(c1 - c0 + 1) : 0u;
            }
            // (a box computed here is kept where the window lists it: the record builder gathers it by element number)
            if (SRC != 0 && cnt[k] > 0) store_box(bs.out + i, b.lo, b.hi, ids.x, ids.y, ids.z, ids.w);
        }
        mine += cnt[k];
    }
    const uint32_t incl = (uint32_t)wave_incl_scan((int)mine);
    const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
    __shared__ uint32_t wave_tot[16], wave_base[16], wave_ext[16];
    const int w = threadIdx.x >> 6;
    if (lane_id() == 63) wave_tot[w] = total;
    if (extq_out) {
        const uint32_t wmax = wave_max_u32_dpp(extq);
        if (lane_id() == 0) wave_ext[w] = wmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (extq_out) {
            uint32_t m = 0;
            for (int k = 0; k < (int)(blockDim.x >> 6); k++) m = max(m, wave_ext[k]);
            if (m) atomicMax(extq_out, m);
        }
        uint32_t sum = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) {
            wave_base[k] = sum;
            sum += wave_tot[k];
        }
        uint32_t b0 = sum ? atomicAdd(cursor, sum) : 0u;
        if (place && sum) b0 = atomicAdd(place, sum);
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) wave_base[k] += b0;
    }
    __syncthreads();
    uint32_t at = wave_base[w] + incl - mine;
    if (mine == 0 || (unsigned long long)at + mine > capacity) return;
#pragma unroll
    for (int k = 0; k < FILL_PER; k++) {
        const int i = block * FILL_BOXES + k * (int)blockDim.x + (int)threadIdx.x;
        for (int ca = s[k].a0; ca <= s[k].a1; ca++)
            for (int cb = s[k].b0; cb <= s[k].b1; cb++) {
                const int cell = ca * g.Sb + cb;
                if (cell < cell_lo || cell >= cell_hi) continue;
                key[at] = (uint32_t)(((unsigned long long)cell << g.xb) | q[k]) | tag; // xb may be 32
                idx[at] = (uint32_t)i;
                ++at;
            }
    }
}

template <int SRC>
__global__ void cell_fill_append_k(BoxSrc bs, int n, const GridParams* __restrict__ gp, int cell_lo, int cell_hi,
                                   uint32_t* __restrict__ cursor, uint32_t capacity, uint32_t* __restrict__ key,
                                   uint32_t* __restrict__ idx, int tagged, uint32_t* __restrict__ place,
                                   const ShardWindow* __restrict__ d_win)
{
    cell_fill_append_body<SRC>(bs, n, gp, cell_lo, cell_hi, cursor, capacity, key, idx, tagged, place, (int)blockIdx.x, d_win);
}
// both lists of a merged two-list build in ONE launch (the first blocks_a blocks: list A): they fill the same buffers by the
// same placement cursor and only count apart
template <int SRC_A, int SRC_B>
__global__ void cell_fill_append2_k(BoxSrc a, int na, BoxSrc b, int nb, int blocks_a, const GridParams* __restrict__ gp, int cell_lo,
                                    int cell_hi, uint32_t* __restrict__ cursors /* [0] A, [1] B, [2] placement, [3] list A's extent (quantised) */, uint32_t capacity,
                                    uint32_t* __restrict__ key, uint32_t* __restrict__ idx, const ShardWindow* __restrict__ d_win)
{
    if ((int)blockIdx.x < blocks_a)
        cell_fill_append_body<SRC_A>(a, na, gp, cell_lo, cell_hi, cursors, capacity, key, idx, 0, cursors + 2, (int)blockIdx.x, d_win,
                                     cursors + 3);
    else
        cell_fill_append_body<SRC_B>(b, nb, gp, cell_lo, cell_hi, cursors + 1, capacity, key, idx, 1, cursors + 2,
                                     (int)blockIdx.x - blocks_a, d_win);
}

// Multi-GPU: this rank's window of cells from the sampled histogram, ON THE DEVICE (round 2 read the 64 KB histogram back
// and dealt the cells out on the host: an 80 us round trip in every rank's build, and a rank's build is all latency).
// The same rule as shard_bounds() (api.hip; mirrored in sccd/dist.py): window r starts at the first cell k whose midpoint in
// running weight, prefix[k] + w[k] / 2, reaches total * r / parts -- that expression never decreases with k, so the first
// such cell is found by bisection over the scanned histogram.  One block of 1024 threads, 16 cells each.
__global__ __launch_bounds__(1024) void shard_window_k(const uint32_t* __restrict__ hist, const GridParams* __restrict__ gp, int stride,
                                                       int rank, int parts, ShardWindow* __restrict__ out)
{
    static_assert(SCCD_MAX_CELLS == 16384, "shard_window_k: 1024 threads x 16 cells");
    __shared__ uint32_t pre[SCCD_MAX_CELLS]; // exclusive prefix sums
    __shared__ uint32_t wsum[16];
    const int n = gp->n_cells;
    const int t = threadIdx.x;
    uint32_t w[16], sum = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int k = t * 16 + i;
        w[i] = k < n ? hist[k] : 0u;
        sum += w[i];
    }
    const uint32_t incl = (uint32_t)wave_incl_scan((int)sum);
    if (lane_id() == 63) wsum[t >> 6] = incl;
    __syncthreads();
    uint32_t base = incl - sum;
    for (int k = 0; k < (t >> 6); k++) base += wsum[k];
    uint32_t total = 0;
    for (int k = 0; k < 16; k++) total += wsum[k];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        pre[t * 16 + i] = base;
        base += w[i];
    }
    __syncthreads();
    __shared__ int bound[2];
    if (t < 2) {
        const int r = rank + t; // bounds[rank], bounds[rank + 1]
        int at;
        if (r <= 0) at = 0;
        else if (r >= parts) at = n;
        else {
            const unsigned long long target = (unsigned long long)total * (unsigned long long)r / (unsigned long long)parts;
            int lo = 0, hi = n; // first k with pre[k] + hist[k] / 2 >= target, else n
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if ((unsigned long long)pre[mid] + hist[mid] / 2u < target) lo = mid + 1;
                else hi = mid;
            }
            at = lo;
        }
        bound[t] = at;
    }
    __syncthreads();
    if (t == 0) {
        const int lo = bound[0], hi = max(bound[0], bound[1]);
        out->cell_lo = lo;
        out->cell_hi = hi;
        out->n_cells = n;
        out->pad = 0;
        out->total_est = (unsigned long long)total * (unsigned long long)stride;
        const uint32_t p_hi = hi < n ? pre[hi] : total, p_lo = lo < n ? pre[lo] : total;
        out->window_est = (unsigned long long)(p_hi - p_lo) * (unsigned long long)stride;
    }
}

// payload movement of thrust::sort_by_key (aabb.cu:107-109) as ONE gather after the index sort, fused with what
// split_boxes derives per box (aabb.cu:40-72): the sorted 80-byte records of the sweep, five 16-byte pieces in five
// arrays (internal.hpp).  MODE 1 / 2: this list is the row list A / B of a two-list sweep and `other` holds the sorted
// keys of the column list; every row also gets its FIRST CANDIDATE COLUMN
//     rows A: lower_bound(keys B, K(min_a))      -- the columns with K(min_a) <= K(min_b) <= K(max_a) start there
//     rows B: upper_bound(keys A, K(min_b))      -- ... with K(min_b) <  K(min_a) <= K(max_b)
// (the reference walks j = i + 1, ... of ONE merged list, sweep.cu:125-131; two lists swept as two classes never test
// a vertex against a vertex or a face against a face).  A block's consecutive rows all start inside one window of
// the column keys: two waves find its ends with 64 probes per round, every row then searches inside the window (a few
// hundred keys in this CU's L1).  The sweep finds the END of a row's columns itself (the first key beyond K(max)).
// own_tagged / other_tagged: this list's / the column list's keys carry the list tag of a merged sort (grid tag_bit).
// Rows per block: 512.  As kernels the two record launches of a step do not care (1,024 / 512 / 256: noise) -- but ccd() runs the edge
// list's beside the vertex-face SWEEP (drivers.hip: the records gate), whose two 216-register waves per SIMD leave 80 registers: eight
// waves of 32 fit beside them, sixteen -- a block of 1,024 rows -- wait for a CU the sweep has left.
#ifndef ER_THREADS_
#define ER_THREADS_ 512
#endif
constexpr int ER_THREADS = ER_THREADS_;
constexpr int ER_WINDOW = ER_THREADS_ * 4; // column keys of a block's window staged in LDS (8 KB per 512 threads)
struct RecordArgs { // one list's share of a record launch
    const sccd_aabb* raw;
    const uint32_t *key, *idx;
    int m;
    int own_tagged;
    const uint32_t* other;
    int n_other, other_tagged;
    uint4* recs;
    uint32_t pstride;
};
template <int MODE>
__device__ __forceinline__ void entry_record_body(const RecordArgs& a, const GridParams* __restrict__ gp, unsigned* s_win, uint32_t* s_keys, int block,
                                                  uint32_t ext_q = 0xFFFFFFFFu)
{
    const sccd_aabb* __restrict__ raw = a.raw;
    const uint32_t* __restrict__ key = a.key;
    const uint32_t* __restrict__ idx = a.idx;
    const uint32_t* __restrict__ other = a.other;
    uint4* __restrict__ recs = a.recs;
    const int m = a.m, own_tagged = a.own_tagged, n_other = a.n_other, other_tagged = a.other_tagged;
    const uint32_t pstride = a.pstride;
    const int e = block * (int)blockDim.x + (int)threadIdx.x;
    const bool valid = e < m;
    const GridParams g = *gp;
    const uint32_t tag = g.tag_bit >= 0 ? (1u << g.tag_bit) : 0u;
    const uint32_t own_strip = own_tagged ? tag : 0u, other_or = other_tagged ? tag : 0u;
    const uint32_t k_e = valid ? (key[e] & ~own_strip) : 0u;
    // the box is asked for FIRST: its two dependent loads (index, then the 64-byte record) are in flight while the block's window and
    // the row's first column are searched for (behind the searches they were another ~3 us of every block's latency chain)
    double4 q0 = make_double4(0.0, 0.0, 0.0, 0.0);
    double2 q1 = make_double2(0.0, 0.0);
    int4 ids = make_int4(0, 0, 0, 0);
    if (valid) {
        const sccd_aabb* src = raw + idx[e];
        q0 = reinterpret_cast<const double4*>(src)[0];
        q1 = reinterpret_cast<const double2*>(src)[2];
        ids = reinterpret_cast<const int4*>(src)[3];
    }
    uint32_t start = (uint32_t)e + 1u;
    if (MODE == 1 && ext_q == 0xFFFFFFFEu) {
        start = 0u; // (list A of a one-class sweep: its entries are columns only, no row of it is ever swept)
    } else if (MODE != 0) {
        // ext_q != NO_EXT (list B's rows, ONE-CLASS sweep: sweep.hip launch_sweep_two): the row's window also reaches BACK, over
        // the columns that start before the row and may still overlap it -- those whose quantised start is at most ext_q (the
        // other list's largest quantised extent, from the fill) below the row's.  The key with the coordinate lowered keeps
        // the rows' order, so the block's window search works as before.
        const uint32_t xmask = g.xb >= 32 ? 0xFFFFFFFFu : ((1u << g.xb) - 1u);
        auto back = [&](uint32_t k) -> uint32_t {
            const uint32_t xq = k & xmask;
            return (k & ~xmask) | (xq > ext_q ? xq - ext_q : 0u);
        };
        const bool one_class = MODE == 2 && ext_q != 0xFFFFFFFFu;
        const int w = threadIdx.x >> 6;
        if (w < 2) { // wave 0: the window's lower end from the block's first key; wave 1: its upper end from the last one
            const int e_first = block * (int)blockDim.x, e_last = min(e_first + (int)blockDim.x, m) - 1;
            uint32_t v = key[w == 0 ? e_first : e_last] & ~own_strip;
            if (one_class && w == 0) v = back(v);
            v |= other_or;
            const unsigned at = (MODE == 1 || (one_class && w == 0)) ? wave_bound_u32<false>(other, (unsigned)n_other, v)
                                                                     : wave_bound_u32<true>(other, (unsigned)n_other, v);
            if (lane_id() == 0) s_win[w] = at;
        }
        __syncthreads();
        const unsigned w0 = s_win[0], w1 = max(s_win[0], s_win[1]);
        // the window's keys go through LDS when they fit (they do, by a wide margin, unless one list dwarfs the other): a row's
        // binary search is then ~10 LDS reads instead of ~10 dependent trips to the vector cache
        const uint32_t needle = (MODE == 1 || !one_class) ? (k_e | other_or) : (back(k_e) | other_or);
        if (w1 - w0 <= (unsigned)ER_WINDOW) { // (block-uniform)
            for (unsigned i = threadIdx.x; i < w1 - w0; i += blockDim.x) s_keys[i] = other[w0 + i];
            __syncthreads();
            if (valid) start = w0 + ((MODE == 1 || one_class) ? lower_bound_in(s_keys, 0u, w1 - w0, needle) : upper_bound_in(s_keys, 0u, w1 - w0, needle));
        } else if (valid) {
            start = (MODE == 1 || one_class) ? lower_bound_in(other, w0, w1, needle) : upper_bound_in(other, w0, w1, needle);
        }
    }
    if (!valid) return;
    const double lo[3] = { q0.x, q0.y, q0.z };
    const double hi[3] = { q0.w, q1.x, q1.y };
    reinterpret_cast<double2*>(recs + (size_t)REC_X * pstride)[e] = make_double2(sel3d(lo, g.axis), sel3d(hi, g.axis));
    reinterpret_cast<double2*>(recs + (size_t)REC_A * pstride)[e] = make_double2(sel3d(lo, g.aa), sel3d(hi, g.aa));
    reinterpret_cast<double2*>(recs + (size_t)REC_B * pstride)[e] = make_double2(sel3d(lo, g.ab), sel3d(hi, g.ab));
    reinterpret_cast<int4*>(recs + (size_t)REC_ID * pstride)[e] = ids;
    const uint32_t cellbits = (uint32_t)((((unsigned long long)k_e) >> g.xb) << g.xb);
    const uint32_t kmax = cellbits | grid_qx(g, sel3d(hi, g.axis));
    const uint32_t lowcell = (uint32_t)grid_cell_a(g, sel3d(lo, g.aa)) | ((uint32_t)grid_cell_b(g, sel3d(lo, g.ab)) << 16);
    recs[(size_t)REC_AUX * pstride + e] = make_uint4(k_e, kmax, lowcell, start);
}
template <int MODE>
__global__ __launch_bounds__(ER_THREADS) void entry_record_k(RecordArgs a, const GridParams* __restrict__ gp, const uint32_t* __restrict__ d_tot,
                                                             int expect_bits)
{
    // d_tot (may be null): the list's entry count in DEVICE memory -- a build whose records are made before the host knows
    // the count (api.hip: the speculative build) launches for the count it expects; the real one is taken from here
    __shared__ unsigned s_win[2];
    __shared__ uint32_t s_keys[MODE != 0 ? ER_WINDOW : 1];
    if (d_tot) { // (a count beyond the bound the launch was sized for is a failed guess: nothing is done here, the host finds out)
        // (so is another key width than the sort was run for: the pairs are not in order then)
        if (d_tot[0] > (uint32_t)a.m || gp->key_bits != expect_bits) return;
        a.m = (int)d_tot[0];
    }
    if ((long long)blockIdx.x * ER_THREADS >= a.m) return;
    entry_record_body<MODE>(a, gp, s_win, s_keys, (int)blockIdx.x);
}
// both lists of a two-list build in ONE launch (the first blocks_a blocks: list A's rows; two launches in a row sat on the
// critical path of every vertex-face pass)
__global__ __launch_bounds__(ER_THREADS) void entry_record2_k(RecordArgs a, RecordArgs b, int blocks_a, const GridParams* __restrict__ gp,
                                                              const uint32_t* __restrict__ d_tot, int expect_bits,
                                                              const uint32_t* __restrict__ d_extq /* one-class sweep: list A's extent, or null */)
{
    __shared__ unsigned s_win[2];
    __shared__ uint32_t s_keys[ER_WINDOW];
    if (d_tot) { // device-side counts (entry_record_k): ONE merged, sorted array -- list A's pairs, then list B's
        if (d_tot[0] > (uint32_t)a.m || d_tot[1] > (uint32_t)b.m || gp->key_bits != expect_bits) return; // (a failed guess: entry_record_k)
        const int ma = (int)d_tot[0], mb = (int)d_tot[1];
        a.m = b.n_other = ma;
        b.m = a.n_other = mb;
        b.key = a.key + ma;
        b.idx = a.idx + ma;
        a.other = b.key;
        b.other = a.key;
        blocks_a = (ma + ER_THREADS - 1) / ER_THREADS;
        if ((int)blockIdx.x >= blocks_a + (mb + ER_THREADS - 1) / ER_THREADS) return;
    }
    if ((int)blockIdx.x < blocks_a) entry_record_body<1>(a, gp, s_win, s_keys, (int)blockIdx.x, d_extq ? 0xFFFFFFFEu : 0xFFFFFFFFu);
    else entry_record_body<2>(b, gp, s_win, s_keys, (int)blockIdx.x - blocks_a, d_extq ? min(*d_extq, 0xFFFFFFF0u) : 0xFFFFFFFFu);
}

// sum and sum of squares of the box centres per axis (sort_and_sweep.cpp:176-186): per-block
// partials, added up on the host in index order (deterministic, like the extent sums)
__global__ void centre_moments_k(const sccd_aabb* __restrict__ raw, int n, double* __restrict__ part /* [gridDim.x][6] */)
{
    double s[3] = { 0, 0, 0 }, s2[3] = { 0, 0, 0 };
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const BoxLoad b = load_box_geom(raw + i);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double cc = (b.lo[k] + b.hi[k]) / 2;
            s[k] += cc;
            s2[k] += cc * cc;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s[k] += __shfl_xor(s[k], o, 64);
            s2[k] += __shfl_xor(s2[k], o, 64);
        }
    }
    __shared__ double red[TPB / 64][6];
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            red[w][k] = s[k];
            red[w][3 + k] = s2[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double t = red[0][threadIdx.x];
#pragma unroll
        for (int j = 1; j < TPB / 64; j++) t += red[j][threadIdx.x];
        part[blockIdx.x * 6 + threadIdx.x] = t;
    }
}

} // namespace

void launch_pack_vertices(sccd_ctx* c, const double* dV0, const double* dV1, int nV, double* dV)
{
    if (nV == 0) return;
    hipLaunchKernelGGL(pack_vertices_k, dim3(grid_for(nV)), dim3(TPB), 0, c->stream, dV0, dV1, nV, dV);
    SCCD_HIP(hipGetLastError());
}
void launch_pack_edges(sccd_ctx* c, const int32_t* dE, int nE, int nV, int2* out, unsigned* bad)
{
    if (nE == 0) return;
    hipLaunchKernelGGL(pack_edges_k, dim3(grid_for(nE)), dim3(TPB), 0, c->stream, dE, nE, nV, out, bad);
    SCCD_HIP(hipGetLastError());
}
void launch_pack_faces(sccd_ctx* c, const int32_t* dF, int nF, int nV, int4* out, unsigned* bad)
{
    if (nF == 0) return;
    hipLaunchKernelGGL(pack_faces_k, dim3(grid_for(nF)), dim3(TPB), 0, c->stream, dF, nF, nV, out, bad);
    SCCD_HIP(hipGetLastError());
}
// builders: grid-stride over at most SCCD_STATS_BLOCKS blocks (one extent partial per block)
int launch_vertex_boxes(sccd_ctx* c, const double* dV, int nV, double inflation, sccd_aabb* out, GridStats* st, double* part)
{
    if (nV == 0) return 0;
    const int grid = std::min(grid_for(nV), SCCD_STATS_BLOCKS);
    unsigned long long* const t_first = c->step_stamp_armed ? reinterpret_cast<unsigned long long*>(c->mailbox_dev + SCCD_MAILBOX_BYTES + 64) : nullptr;
    c->step_stamp_armed = false; // (one stamp per step: its first kernel)
    if (c->scalar_f32) hipLaunchKernelGGL(vertex_boxes_k<true>, dim3(grid), dim3(TPB), 0, c->stream, dV, nV, inflation, out, st, part, t_first);
    else hipLaunchKernelGGL(vertex_boxes_k<false>, dim3(grid), dim3(TPB), 0, c->stream, dV, nV, inflation, out, st, part, t_first);
    SCCD_HIP(hipGetLastError());
    return grid;
}
int launch_edge_boxes(sccd_ctx* c, const sccd_aabb* vb, const int2* E, int nE, sccd_aabb* out, GridStats* st, double* part)
{
    if (nE == 0) return 0;
    const int grid = std::min(grid_for(nE), SCCD_STATS_BLOCKS);
    hipLaunchKernelGGL(edge_boxes_k, dim3(grid), dim3(TPB), 0, c->stream, vb, E, nE, out, st, part);
    SCCD_HIP(hipGetLastError());
    return grid;
}
int launch_face_boxes(sccd_ctx* c, const sccd_aabb* vb, const int4* F, int nF, sccd_aabb* out, GridStats* st, double* part)
{
    if (nF == 0) return 0;
    const int grid = std::min(grid_for(nF), SCCD_STATS_BLOCKS);
    hipLaunchKernelGGL(face_boxes_k, dim3(grid), dim3(TPB), 0, c->stream, vb, F, nF, out, st, part);
    SCCD_HIP(hipGetLastError());
    return grid;
}
// edge and face boxes in one launch; *n_part_e / *n_part_f: block partials written per list
void launch_edge_face_boxes(sccd_ctx* c, const sccd_aabb* vb, const int2* E, int nE, sccd_aabb* out_e, GridStats* st_e, double* part_e,
                            int* n_part_e, const int4* F, int nF, sccd_aabb* out_f, GridStats* st_f, double* part_f, int* n_part_f)
{
    const int ge = std::min(grid_for(nE), SCCD_STATS_BLOCKS), gf = std::min(grid_for(nF), SCCD_STATS_BLOCKS);
    *n_part_e = ge;
    *n_part_f = gf;
    hipLaunchKernelGGL(edge_face_boxes_k, dim3((unsigned)(ge + gf)), dim3(TPB), 0, c->stream, vb, E, nE, out_e, st_e, part_e, ge, F, nF,
                       out_f, st_f, part_f);
    SCCD_HIP(hipGetLastError());
}
// returns the number of block partials written to `part` (at most SCCD_STATS_BLOCKS)
int launch_box_stats(sccd_ctx* c, const sccd_aabb* raw, int n, GridStats* st, double* part)
{
    if (n == 0) return 0;
    const int grid = std::min(grid_for(n), SCCD_STATS_BLOCKS);
    hipLaunchKernelGGL(box_stats_k, dim3(grid), dim3(TPB), 0, c->stream, raw, n, st, part);
    SCCD_HIP(hipGetLastError());
    return grid;
}
void launch_grid_setup(sccd_ctx* c, const GridStats* st_a, const double* part_a, int n_part_a, const GridStats* st_b,
                       const double* part_b, int n_part_b, int n_total, int axis, double cell_factor, int shrink,
                       GridParams* g, uint32_t* cursors, bool reserve_tag, uint32_t* zero_hist)
{
    const int max_cells = SCCD_DEFAULT_CELLS;
    // (the context's counter block: SweepCounters at 0, NarrowCounters at 2048 -- build.hip / drivers.hip)
    static_assert(offsetof(NarrowCounters, toi_bits) == 0, "grid_setup_k writes the TOI into word 0");
    unsigned long long* const sweep_cnt = c->scalars.as<unsigned long long>();
    const bool np_init = c->np_init_pending;
    unsigned long long toi_bits = 0;
    std::memcpy(&toi_bits, &c->np_init_toi, 8);
    hipLaunchKernelGGL(grid_setup_k, dim3(1), dim3(SCCD_STATS_BLOCKS), 0, c->stream, st_a, part_a, n_part_a, st_b, part_b, n_part_b,
                       n_total, axis, cell_factor, shrink, g, cursors, max_cells, reserve_tag ? 1 : 0, zero_hist, sweep_cnt,
                       (int)(sizeof(SweepCounters) / 8), np_init ? sweep_cnt + 2048 / 8 : nullptr, (int)(sizeof(NarrowCounters) / 8), toi_bits,
                       (int)(offsetof(NarrowCounters, peer_word) / 8), (unsigned long long)(uintptr_t)c->np_init_peer);
    SCCD_HIP(hipGetLastError());
    c->sweep_cnt_cleared = true; // (consumed by the next sweep of this context: bp_detect_partial)
    if (np_init) {
        c->np_init_pending = false;
        c->np_init_peer = nullptr; // (this build's launch only)
        c->np_uploaded = true; // (narrow_phase_begin: nothing to upload for a launch that starts from this TOI)
        c->np_uploaded_toi = c->np_init_toi;
    }
}
// the boxes of a list for the kernels above: the raw array, or (lazy lists) the recipe to compute them
static BoxSrc box_src(const sccd_boxes* b)
{
    return BoxSrc { b->raw.as<sccd_aabb>(), b->lazy_vb, b->lazy_elems, b->raw.as<sccd_aabb>() };
}
static int src_kind(const sccd_boxes* b) { return b->lazy ? (b->kind == BOX_EDGE ? 1 : 2) : 0; }
static int hist_blocks_for(int n, int stride)
{
    return n > 0 ? std::max(1, std::min(grid_for((n + stride - 1) / stride), 512)) : 0; // (64 KB of LDS each: two per CU)
}
// the sampled cell histogram of one list, or of both lists of a two-list build in one launch (B may be null)
void launch_cell_hist(sccd_ctx* c, const sccd_boxes* A, const sccd_boxes* B, const GridParams* g, int stride, uint32_t* hist)
{
    const int na = A->n, nb = B ? B->n : 0;
    const int blocks_a = hist_blocks_for(na, stride), blocks_b = hist_blocks_for(nb, stride);
    if (blocks_a + blocks_b == 0) return;
    const BoxSrc a = box_src(A), b = B ? box_src(B) : box_src(A);
    const dim3 block(TPB);
    if (blocks_b == 0 || blocks_a == 0) {
        const sccd_boxes* L = blocks_a ? A : B;
        const BoxSrc bs = blocks_a ? a : b;
        const dim3 grid((unsigned)(blocks_a + blocks_b));
        switch (src_kind(L)) {
        case 0: hipLaunchKernelGGL(cell_hist_k<0>, grid, block, 0, c->stream, bs, L->n, g, stride, hist); break;
        case 1: hipLaunchKernelGGL(cell_hist_k<1>, grid, block, 0, c->stream, bs, L->n, g, stride, hist); break;
        default: hipLaunchKernelGGL(cell_hist_k<2>, grid, block, 0, c->stream, bs, L->n, g, stride, hist); break;
        }
    } else {
        auto go = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3((unsigned)(blocks_a + blocks_b)), block, 0, c->stream, a, na, b, nb, blocks_a, g, stride, hist);
        };
        // (list A of a two-list build is the vertices, always there; list B may be lazy faces or edges)
        SCCD_REQUIRE(src_kind(A) == 0, "broad phase: a lazy list A");
        if (src_kind(B) == 0) go(cell_hist2_k<0, 0>);
        else if (src_kind(B) == 2) go(cell_hist2_k<0, 2>);
        else go(cell_hist2_k<0, 1>);
    }
    SCCD_HIP(hipGetLastError());
}
// statistics of a lazy list from every stride-th element; returns the number of block partials written
int launch_elem_stats(sccd_ctx* c, const sccd_boxes* b, int stride, GridStats* st, double* part)
{
    const int n = b->n;
    if (n == 0) return 0;
    const int grid = std::min(grid_for((n + stride - 1) / stride), SCCD_STATS_BLOCKS);
    const BoxSrc bs = box_src(b);
    if (src_kind(b) == 1) hipLaunchKernelGGL(elem_stats_k<1>, dim3(grid), dim3(TPB), 0, c->stream, bs, n, stride, st, part);
    else hipLaunchKernelGGL(elem_stats_k<2>, dim3(grid), dim3(TPB), 0, c->stream, bs, n, stride, st, part);
    SCCD_HIP(hipGetLastError());
    return grid;
}
// the statistics of a mesh's lazy edge and face lists in one launch; *n_part_e / *n_part_f: block partials written per list
void launch_elem_stats_two(sccd_ctx* c, const sccd_boxes* e, const sccd_boxes* f, int stride, int* n_part_e, int* n_part_f)
{
    const int ge = std::min(grid_for((e->n + stride - 1) / stride), SCCD_STATS_BLOCKS), gf = std::min(grid_for((f->n + stride - 1) / stride), SCCD_STATS_BLOCKS);
    *n_part_e = ge;
    *n_part_f = gf;
    hipLaunchKernelGGL(elem_stats2_k, dim3((unsigned)(ge + gf)), dim3(TPB), 0, c->stream, box_src(e), e->n, ge, box_src(f), f->n, stride, e->stats_head(),
                       e->stats_part(), f->stats_head(), f->stats_part());
    SCCD_HIP(hipGetLastError());
}
void launch_shard_window(sccd_ctx* c, const uint32_t* hist, const GridParams* g, int stride, int rank, int parts, ShardWindow* out)
{
    hipLaunchKernelGGL(shard_window_k, dim3(1), dim3(1024), 0, c->stream, hist, g, stride, rank, parts, out);
    SCCD_HIP(hipGetLastError());
}
void launch_cell_fill_append(sccd_ctx* c, const sccd_boxes* b, const GridParams* g, int cell_lo, int cell_hi,
                             uint32_t* cursor, uint32_t capacity, uint32_t* key, uint32_t* idx, bool tagged, uint32_t* place,
                             const ShardWindow* d_win)
{
    const int n = b->n;
    if (n == 0) return;
    const dim3 grid((unsigned)((n + FILL_BOXES - 1) / FILL_BOXES)), block(1024);
    const BoxSrc bs = box_src(b);
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, block, 0, c->stream, bs, n, g, cell_lo, cell_hi, cursor, capacity, key, idx, tagged ? 1 : 0, place,
                           d_win);
    };
    switch (src_kind(b)) {
    case 0: go(cell_fill_append_k<0>); break;
    case 1: go(cell_fill_append_k<1>); break;
    default: go(cell_fill_append_k<2>); break;
    }
    SCCD_HIP(hipGetLastError());
}
void launch_cell_fill_append_two(sccd_ctx* c, const sccd_boxes* A, const sccd_boxes* B, const GridParams* g,
                                 int cell_lo, int cell_hi, uint32_t* cursors, uint32_t capacity, uint32_t* key, uint32_t* idx,
                                 const ShardWindow* d_win)
{
    const int na = A->n, nb = B->n;
    const int blocks_a = (na + FILL_BOXES - 1) / FILL_BOXES, blocks_b = (nb + FILL_BOXES - 1) / FILL_BOXES;
    if (blocks_a + blocks_b == 0) return;
    const BoxSrc a = box_src(A), b = box_src(B);
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)(blocks_a + blocks_b)), dim3(1024), 0, c->stream, a, na, b, nb, blocks_a, g, cell_lo,
                           cell_hi, cursors, capacity, key, idx, d_win);
    };
    // (list A of a two-list build is the vertices, always there; list B may be lazy faces)
    SCCD_REQUIRE(src_kind(A) == 0, "broad phase: a lazy list A");
    if (src_kind(B) == 0) go(cell_fill_append2_k<0, 0>);
    else if (src_kind(B) == 2) go(cell_fill_append2_k<0, 2>);
    else go(cell_fill_append2_k<0, 1>);
    SCCD_HIP(hipGetLastError());
}
void launch_cell_count(sccd_ctx* c, const sccd_aabb* raw, int n, const GridParams* g, int cell_lo, int cell_hi,
                       uint32_t* counts)
{
    if (n == 0) return;
    hipLaunchKernelGGL(cell_count_k, dim3(grid_for(n)), dim3(TPB), 0, c->stream, raw, n, g, cell_lo, cell_hi, counts);
    SCCD_HIP(hipGetLastError());
}
void launch_cell_fill(sccd_ctx* c, const sccd_aabb* raw, int n, const GridParams* g, int cell_lo, int cell_hi,
                      const uint32_t* offsets, uint32_t* key, uint32_t* idx)
{
    if (n == 0) return;
    hipLaunchKernelGGL(cell_fill_k, dim3(grid_for(n)), dim3(TPB), 0, c->stream, raw, n, g, cell_lo, cell_hi, offsets,
                       key, idx);
    SCCD_HIP(hipGetLastError());
}
static RecordArgs record_args(const sccd_aabb* raw, const uint32_t* key, const uint32_t* idx, int m, const uint32_t* other, int n_other,
                              bool own_tagged, bool other_tagged, SortedList* out)
{
    const size_t n = ((size_t)m + SCCD_LIST_PAD + 63) & ~(size_t)63;
    SCCD_REQUIRE(5 * n < (1ull << 28), "broad phase: too many cell entries"); // (32-bit piece offsets in units of 16 bytes)
    out->recs.ensure(sizeof(uint4) * 5 * n);
    out->pstride = (uint32_t)n;
    return RecordArgs { raw, key, idx, m, own_tagged ? 1 : 0, other, n_other, other_tagged ? 1 : 0, out->recs.as<uint4>(), out->pstride };
}
void launch_entry_records(sccd_ctx* c, const sccd_aabb* raw, const uint32_t* key, const uint32_t* idx, int m,
                          const GridParams* g, int mode, const uint32_t* other, int n_other, bool own_tagged,
                          bool other_tagged, SortedList* out, const uint32_t* d_tot, int expect_bits)
{
    // (d_tot != null: m is the bound the launch is sized for, the real count is read on the device)
    if (m == 0) return;
    SCCD_REQUIRE(!(d_tot && mode != 0), "broad phase: device-side counts serve the one-list and the merged two-list records");
    const RecordArgs a = record_args(raw, key, idx, m, other, n_other, own_tagged, other_tagged, out);
    const dim3 grid((unsigned)((m + ER_THREADS - 1) / ER_THREADS)), block(ER_THREADS);
    if (mode == 0) hipLaunchKernelGGL(entry_record_k<0>, grid, block, 0, c->stream, a, g, d_tot, expect_bits);
    else if (mode == 1) hipLaunchKernelGGL(entry_record_k<1>, grid, block, 0, c->stream, a, g, d_tot, expect_bits);
    else hipLaunchKernelGGL(entry_record_k<2>, grid, block, 0, c->stream, a, g, d_tot, expect_bits);
    SCCD_HIP(hipGetLastError());
}
// the records of both lists of a two-list build in one launch (list A's rows look their first column up among keys_b, ...)
void launch_entry_records_two(sccd_ctx* c, const sccd_aabb* raw_a, const uint32_t* key_a, const uint32_t* idx_a, int ma,
                              const sccd_aabb* raw_b, const uint32_t* key_b, const uint32_t* idx_b, int mb, bool b_tagged,
                              const GridParams* g, SortedList* out_a, SortedList* out_b, const uint32_t* d_tot, int expect_bits,
                              const uint32_t* d_extq)
{
    if (ma == 0 || mb == 0) return;
    const RecordArgs a = record_args(raw_a, key_a, idx_a, ma, key_b, mb, false, b_tagged, out_a);
    const RecordArgs b = record_args(raw_b, key_b, idx_b, mb, key_a, ma, b_tagged, false, out_b);
    const int blocks_a = (ma + ER_THREADS - 1) / ER_THREADS, blocks_b = (mb + ER_THREADS - 1) / ER_THREADS;
    // (device-side counts: one block more than the bounds need -- the split between the lists moves with the real counts)
    hipLaunchKernelGGL(entry_record2_k, dim3((unsigned)(blocks_a + blocks_b + (d_tot ? 1 : 0))), dim3(ER_THREADS), 0, c->stream, a, b, blocks_a, g, d_tot, expect_bits, d_extq);
    SCCD_HIP(hipGetLastError());
}

int pick_sort_axis(sccd_ctx* c, const sccd_aabb* raw, int n, const sccd_aabb* raw_b, int n_b)
{
    if (n + n_b == 0) return 0;
    const int ga = n > 0 ? std::min(grid_for(n), SCCD_STATS_BLOCKS) : 0, gb = n_b > 0 ? std::min(grid_for(n_b), SCCD_STATS_BLOCKS) : 0;
    c->tmp2.ensure(sizeof(double) * 6 * (size_t)(ga + gb));
    double* part = c->tmp2.as<double>();
    if (ga) hipLaunchKernelGGL(centre_moments_k, dim3(ga), dim3(TPB), 0, c->stream, raw, n, part);
    if (gb) hipLaunchKernelGGL(centre_moments_k, dim3(gb), dim3(TPB), 0, c->stream, raw_b, n_b, part + 6 * ga);
    SCCD_HIP(hipGetLastError());
    std::vector<double> hp(6 * (size_t)(ga + gb));
    SCCD_HIP(hipMemcpyAsync(hp.data(), part, sizeof(double) * hp.size(), hipMemcpyDeviceToHost, c->stream));
    SCCD_HIP(hipStreamSynchronize(c->stream));
    double h[6] = { 0, 0, 0, 0, 0, 0 };
    for (int j = 0; j < ga + gb; j++)
        for (int k = 0; k < 6; k++) h[k] += hp[(size_t)j * 6 + k];
    double var[3];
    for (int k = 0; k < 3; k++) var[k] = h[3 + k] - h[k] * h[k] / (n + n_b);
    int ax = 0; // sort_and_sweep.cpp:188-195
    if (var[1] > var[0]) ax = 1;
    if (var[2] > var[ax]) ax = 2;
    return ax;
}
