// narrow.hip -- Tight-Inclusion narrow phase on gfx950.
//
// Replaces add_data<>, compute_tolerance<>, ccd_kernel<>, the host BFS loop ccd<>() and the
// CCDBuffer ring (src/scalable_ccd/cuda/narrow_phase/narrow_phase.cu:24-206,
// root_finder.cu:260-457, ccd_buffer.cuh:7-83).
//
// Two algorithms, same results for max_iter < 0 (the final TOI is a min over accepted domains
// and pruning by the running TOI never changes that min -- SURVEY Appendix A.20):
//   algo 0  np_walk_k (narrow_walk.inc): ONE persistent launch.  Every lane walks its query's
//           [t]x[u]x[v] tree depth-first (earlier half first, so the running TOI drops early) with
//           an on-chip stack; idle lanes pick fresh queries from a per-wave LDS staging area that
//           LDS-direct loads refill in the background.  Divergent bisection depth is absorbed per
//           lane: nobody waits for the deepest query of a batch.  No CCDData array in HBM, no
//           per-level host round trip.
//   algo 1  np_level_k: level-synchronous BFS with a host loop, the reference's scheme
//           (root_finder.cu:431-447).  Kept as the in-library cross-check.
#include "internal.hpp"
#include "ti_math.hpp"
#include "ti_math_f32.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------
// shared
__device__ __forceinline__ double toi_load(const unsigned long long* p)
{
    // relaxed agent-scope load: bypasses this CU's L1 so other CUs' atomicMin are seen
    return __longlong_as_double((long long)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// atomicMin on the IEEE bit pattern is valid for non-negative doubles (atomic_min_float.cuh:17-29)
__device__ __forceinline__ void toi_min(unsigned long long* p, double v)
{
    atomicMin(p, (unsigned long long)__double_as_longlong(v));
}

// ------------------------------------------------------------------------------------------
// algo 1: level-synchronous BFS (reference scheme)
// CCDData (ccd_data.cuh:8-26) minus ms, one ARRAY per field (f[k * n + q]: the 24 vertex coordinates, err, tol): the threads of a
// level are ordered by query, roughly, and a thread reads all 30 values of its query -- as 256-byte records a wave's sixteen loads
// each touched 64 cache lines, the same 64, and twenty waves of that per CU do not fit its L1: every load went to the L2 again (a level
// of two million domains: 670 us).  As arrays a load touches the few lines its wave's queries share.
struct LvlData {
    double* f;                    // [30][n]
    unsigned long long* toi_bits; // [n] per-query toi (TOI_PER_QUERY)
    int* nbr_checks;              // [n]
    long long n;
    static size_t bytes(long long n) { return (size_t)n * (30 * 8 + 8 + 4) + 64; }
    static LvlData carve(void* base, long long n)
    {
        LvlData d;
        d.f = static_cast<double*>(base);
        d.toi_bits = reinterpret_cast<unsigned long long*>(d.f + 30 * n);
        d.nbr_checks = reinterpret_cast<int*>(d.toi_bits + n);
        d.n = n;
        return d;
    }
};
struct LvlDomain { // CCDDomain (interval.cuh:30-44)
    double lo[3], hi[3];
    int query_id;
    int pad;
};

// F32 (SCCD_OPT_SCALAR = 1, the reference's float build): the vertices are cast to float first (ccd.cu:103-106) and
// every quantity is float arithmetic (ti_math_f32.hpp); the float values are stored widened in the same records.
template <bool VF, bool F32>
__global__ void np_level_init_k(const double* __restrict__ V, const int2* __restrict__ E, const int4* __restrict__ F,
                                const int2* __restrict__ pairs, long long first, long long n, double tol, bool use_ms,
                                LvlData data, LvlDomain* __restrict__ dom, const int* __restrict__ sel)
{
    // queries [first, first + n) of the call (of the selection `sel`, if given): data[] is indexed by query, dom[] by
    // position in the slice
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long long i = sel ? (long long)sel[first + j] : first + j;
    TIQuery q;
    ti_gather<VF>(V, E, F, pairs[i], q.v);
    if (F32) {
        TIQueryF qf;
#pragma unroll
        for (int a = 0; a < 8; a++)
#pragma unroll
            for (int k = 0; k < 3; k++) qf.v[a][k] = (float)q.v[a][k];
        tif_tolerance<VF>(qf.v, (float)tol, qf.tol);
        tif_error<VF>(qf.v, use_ms, qf.err);
#pragma unroll
        for (int a = 0; a < 8; a++)
#pragma unroll
            for (int k = 0; k < 3; k++) q.v[a][k] = qf.v[a][k];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            q.tol[k] = qf.tol[k];
            q.err[k] = qf.err[k];
        }
    } else {
        ti_tolerance<VF>(q.v, tol, q.tol);
        ti_error<VF>(q.v, use_ms, q.err);
    }
#pragma unroll
    for (int a = 0; a < 8; a++)
#pragma unroll
        for (int k = 0; k < 3; k++) data.f[(long long)(a * 3 + k) * data.n + i] = q.v[a][k];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        data.f[(long long)(24 + k) * data.n + i] = q.err[k];
        data.f[(long long)(27 + k) * data.n + i] = q.tol[k];
    }
    data.toi_bits[i] = 0x7FF0000000000000ull; // +inf (narrow_phase.cu:70)
    data.nbr_checks[i] = 0;
    LvlDomain r;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        r.lo[k] = 0.0;
        r.hi[k] = 1.0;
    }
    r.query_id = (int)i;
    r.pad = 0;
    dom[j] = r;
}

// What a thread of one level launch READS of the state the same launch WRITES (root_finder.cu:287-289, :295, :229:
// the query's check counter and TOI, the global TOI).  In the reference those reads race with the other threads'
// atomicAdd / atomicMin -- any interleaving is a legal outcome.  With a check limit (max_iter >= 0) the interleaving
// decides which domains are dropped, so the kernels then follow the ONE serialisation that does not depend on thread
// order: every thread of a launch reads before any thread of that launch writes.  np_level_snap_k copies that state
// in front of each level; the oracle restates the same order.  Without a limit the live values are read (earlier
// pruning, same result: Appendix A.20).
struct LvlSnap {
    unsigned long long toi_bits;
    int nbr_checks;
    int pad;
};
// d_n_cur (may be null): the level's length is still on the device (the second level of a pair: run_level_sync) -- min(*d_n_cur, n_cur)
__global__ void np_level_snap_k(const LvlDomain* __restrict__ cur, long long n_cur, LvlData data,
                                LvlSnap* __restrict__ snap, NarrowCounters* __restrict__ cnt, const unsigned long long* __restrict__ d_n_cur)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) cnt->toi_level = cnt->toi_bits;
    if (d_n_cur) n_cur = min(n_cur, (long long)*d_n_cur);
    if (i >= n_cur) return;
    const int q = cur[i].query_id; // (several domains of one query store the same values)
    LvlSnap s;
    s.toi_bits = data.toi_bits[q];
    s.nbr_checks = data.nbr_checks[q];
    s.pad = 0;
    snap[q] = s;
}

// One thread per live domain of the level.  The two counters every thread adds to -- the next level's length (a RETURNING atomic:
// the children's place) and the checks made -- are one word each: the compiler folds a wave's adds into one atomic, and a level of a
// million domains was still 16,000 returning atomics on one word, ~200 us of a 360 us launch (the word serves ~75 per us).  The
// block adds once: a scan of the children over its four waves, one atomic per counter per block.  (The ORDER of the next level's
// domains comes from atomics either way and decides nothing: LvlSnap.)  __launch_bounds__: without it the kernel is compiled for
// 1,024 threads and its 56 bytes of dynamically indexed private arrays, which the compiler keeps in LDS, take 57 KB per block -- two
// blocks of 256 per CU; with it 14 KB, and the registers decide (five waves per SIMD).
#ifndef NP_LEVEL_TPB_
#define NP_LEVEL_TPB_ 1024
#endif
constexpr int NP_LEVEL_TPB = NP_LEVEL_TPB_;
template <bool VF, int ARITH, bool F32>
__global__ __launch_bounds__(NP_LEVEL_TPB) void np_level_k(const LvlDomain* __restrict__ cur, long long n_cur, LvlDomain* __restrict__ nxt,
                           unsigned long long* __restrict__ n_nxt, LvlData data, double ms,
                           double tol, int max_iter, bool allow_zero_toi, bool per_query,
                           NarrowCounters* __restrict__ cnt, const LvlSnap* __restrict__ snap,
                           unsigned long long* __restrict__ n_after, const unsigned long long* __restrict__ d_n_cur)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *n_after = 0ull; // (a counter no level reads any more: it will count the level after the next -- run_level_sync)
    if (d_n_cur) n_cur = min(n_cur, (long long)*d_n_cur); // (np_level_snap_k)
    unsigned nk = 0;
    bool checked = false;
    LvlDomain dom {};
    int split = 0;
    double mid = 0.0;
    if (i < n_cur) {
        dom = cur[i];
        const long long qi = dom.query_id;
        const int before = snap ? snap[qi].nbr_checks : data.nbr_checks[qi]; // data_in copy, root_finder.cu:287-288
        atomicAdd(&data.nbr_checks[qi], 1);                                  // :289
        const double prune = snap
            ? __longlong_as_double((long long)(per_query ? snap[qi].toi_bits : cnt->toi_level))
            : (per_query ? toi_load(&data.toi_bits[qi]) : toi_load(&cnt->toi_bits));
        if (!(dom.lo[0] >= prune)                           // :295
            && !(max_iter >= 0 && before > max_iter)) {     // :303
            TIQuery q;
#pragma unroll
            for (int a = 0; a < 8; a++)
#pragma unroll
                for (int k = 0; k < 3; k++) q.v[a][k] = data.f[(long long)(a * 3 + k) * data.n + qi];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                q.err[k] = data.f[(long long)(24 + k) * data.n + qi];
                q.tol[k] = data.f[(long long)(27 + k) * data.n + qi];
            }
            TIStep s;
            if (F32) { // (every stored value is a float: the casts are exact)
                TIQueryF qf;
#pragma unroll
                for (int a = 0; a < 8; a++)
#pragma unroll
                    for (int k = 0; k < 3; k++) qf.v[a][k] = (float)q.v[a][k];
                float lo[3], hi[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    qf.err[k] = (float)q.err[k];
                    qf.tol[k] = (float)q.tol[k];
                    lo[k] = (float)dom.lo[k];
                    hi[k] = (float)dom.hi[k];
                }
                const TIStepF sf = tif_step<VF, ARITH>(qf, lo, hi, (float)ms, (float)tol, allow_zero_toi, (float)prune);
                s.accept = sf.accept;
                s.nk = sf.nk;
                s.split = sf.split;
                s.mid = sf.mid;
                s.checked = sf.checked;
            } else {
                s = ti_step<VF, ARITH>(q, dom.lo, dom.hi, ms, tol, allow_zero_toi, prune);
            }
            checked = s.checked;
            if (s.accept) {
                toi_min(&cnt->toi_bits, dom.lo[0]);
                toi_min(&data.toi_bits[qi], dom.lo[0]);
            }
            nk = (unsigned)s.nk;
            split = s.split;
            mid = s.mid;
        }
    }
    // the block's children and checks: one atomic each
    __shared__ unsigned s_kids[NP_LEVEL_TPB / 64], s_chk[NP_LEVEL_TPB / 64];
    __shared__ unsigned long long s_base;
    const int w = (int)(threadIdx.x >> 6);
    unsigned wave_kids;
    const unsigned incl = wave_incl_scan_dpp(nk, &wave_kids);
    const unsigned wave_chk = (unsigned)popc64(__ballot(checked));
    if (lane_id() == 0) {
        s_kids[w] = wave_kids;
        s_chk[w] = wave_chk;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned kids = 0, chk = 0;
        for (int k = 0; k < NP_LEVEL_TPB / 64; k++) {
            const unsigned v = s_kids[k];
            s_kids[k] = kids; // (-> the wave's offset in the block's range)
            kids += v;
            chk += s_chk[k];
        }
        s_base = kids ? atomicAdd(n_nxt, (unsigned long long)kids) : 0ull;
        if (chk) atomicAdd(&cnt->n_checks, (unsigned long long)chk);
    }
    __syncthreads();
    if (nk > 0) {
        const unsigned long long at = s_base + (unsigned long long)(s_kids[w] + incl - nk);
        LvlDomain c = dom;
        c.hi[split] = mid;
        nxt[at] = c;
        if (nk == 2) {
            c = dom;
            c.lo[split] = mid;
            nxt[at + 1] = c;
        }
    }
}

__global__ void np_fill_u64_k(unsigned long long* __restrict__ p, long long n, unsigned long long v)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void np_copy_per_query_k(LvlData data, long long n, double* __restrict__ out,
                                    const int* __restrict__ sel)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (!sel) out[j] = __longlong_as_double((long long)data.toi_bits[j]);
    else // a selection redone in level order: fold into what the work-queue kernel had found for the query
        atomicMin(reinterpret_cast<unsigned long long*>(out) + sel[j], data.toi_bits[sel[j]]);
}

// d_sel / n_sel: only those queries of the call (np_walk_k's overflow list), else all n_all
// walk_range(first, count): finish those queries of the call depth first (see below); may be empty
template <bool VF> void run_level_sync(sccd_ctx* c, const NarrowParams& p, NarrowCounters* d_cnt, long long n_all,
                                       double* d_per_query_toi, const int* d_sel = nullptr, long long n_sel = 0,
                                       const std::function<void(long long, long long)>& walk_range = nullptr)
{
    const bool use_ms = p.ms > 0; // narrow_phase.cu:128
    const long long n = d_sel ? n_sel : n_all; // queries to run; data[] / snap[] stay indexed by the query's own number
    c->np_scratch0.ensure(LvlData::bytes(n_all));
    const LvlData data = LvlData::carve(c->np_scratch0.p, n_all);
    // (three counters of live domains, used in turn: a level's kernel counts the next level's into one and clears the one after, and
    // the host reads a count through the mailbox (ReadBack: a publishing kernel and a polled word, ~8 us) -- a memset, a copy into
    // pageable memory and a blocking wait per level were most of a culled call's level-order time.  LEVELS GO IN PAIRS where memory
    // allows: the second is launched for twice the first one's length -- all it can be -- and takes the real length from the counter
    // the first one left: one round trip to the host per two levels, three buffers of domains in turn)
    c->tmp0.ensure(3 * sizeof(unsigned long long));
    unsigned long long* const d_n3 = c->tmp0.as<unsigned long long>();
    const int TPB = NP_LEVEL_TPB;
    // Level order keeps every live domain of a level in HBM, and a contact-rich scene doubles them
    // level after level.  The queries are therefore taken in slices (the reference batches too,
    // narrow_phase.cu:141-200): a slice whose level would not fit the budget is started again at
    // half the size -- what it found so far stays valid (accepted domains only ever lower a TOI).
    size_t free_b = 0, total_b = 0;
    SCCD_HIP(hipMemGetInfo(&free_b, &total_b));
    // (8 GB per level buffer = 1.4e8 live domains: beyond that a slice is cut; 4096 queries that still need
    // more are hopeless in level order, and failing early beats filling 288 GB first)
    // (SCCD_LEVEL_BUDGET_MB: a smaller budget per level buffer, for soak runs on shared test machines)
    const size_t budget_cap = lab_env().level_budget_mb > 0 ? std::max<size_t>(64, (size_t)lab_env().level_budget_mb) << 20 : (size_t)8 << 30;
    const size_t budget = std::min<size_t>(budget_cap,
                                           std::max<size_t>((free_b + c->np_scratch1.cap + c->np_scratch2.cap + c->np_scratch5.cap) / 4, (size_t)64 << 20));
    LvlSnap* snap = nullptr; // (a check limit: level-snapshot serialisation, see LvlSnap)
    if (p.max_iter >= 0) {
        c->np_scratch4.ensure(sizeof(LvlSnap) * (size_t)n_all);
        snap = c->np_scratch4.as<LvlSnap>();
    }
    long long slice = n;
    for (long long q0 = 0; q0 < n;) {
        const long long len = std::min(slice, n - q0);
        c->np_scratch1.ensure(sizeof(LvlDomain) * (size_t)len);
        if (c->scalar_f32)
            hipLaunchKernelGGL((np_level_init_k<VF, true>), dim3((unsigned)((len + TPB - 1) / TPB)), dim3(TPB), 0, c->stream,
                               p.V, p.E, p.F, p.pairs, q0, len, p.tol, use_ms, data, c->np_scratch1.as<LvlDomain>(), d_sel);
        else
            hipLaunchKernelGGL((np_level_init_k<VF, false>), dim3((unsigned)((len + TPB - 1) / TPB)), dim3(TPB), 0, c->stream,
                               p.V, p.E, p.F, p.pairs, q0, len, p.tol, use_ms, data, c->np_scratch1.as<LvlDomain>(), d_sel);
        long long n_cur = len;
        DevBuf* const bufs[3] = { &c->np_scratch1, &c->np_scratch2, &c->np_scratch5 };
        unsigned at = 0; // bufs[at] holds the level's domains
        bool fits = true;
        SCCD_HIP(hipMemsetAsync(d_n3, 0, 3 * sizeof(unsigned long long), c->stream));
        unsigned level = 0;
        // one level: n_bound live domains at most (d_len: the real number, if it is still on the device), from `from` into `to`
        auto launch_level = [&](const DevBuf* from, long long n_bound, const unsigned long long* d_len, DevBuf* to) {
            unsigned long long* const d_n = d_n3 + level % 3u;
            unsigned long long* const d_n_after = d_n3 + (level + 1u) % 3u;
            level += 1;
            const dim3 grid((unsigned)((n_bound + TPB - 1) / TPB));
            if (snap) hipLaunchKernelGGL(np_level_snap_k, grid, dim3(TPB), 0, c->stream, from->as<LvlDomain>(), n_bound, data, snap, d_cnt, d_len);
#define SCCD_LAUNCH_LEVEL(AR_, F32_)                                                                                   \
    hipLaunchKernelGGL((np_level_k<VF, AR_, F32_>), grid, dim3(TPB), 0, c->stream, from->as<LvlDomain>(), n_bound,       \
                       to->as<LvlDomain>(), d_n, data, p.ms, p.tol, p.max_iter, (bool)p.allow_zero_toi,                  \
                       d_per_query_toi != nullptr, d_cnt, snap, d_n_after, d_len)
            if (c->scalar_f32) {
                if (p.arith == 1) SCCD_LAUNCH_LEVEL(1, true);
                else SCCD_LAUNCH_LEVEL(0, true);
            } else {
                if (p.arith == 1) SCCD_LAUNCH_LEVEL(1, false);
                else SCCD_LAUNCH_LEVEL(0, false);
            }
#undef SCCD_LAUNCH_LEVEL
            SCCD_HIP(hipGetLastError());
            return d_n; // (where the next level's length will be)
        };
        while (n_cur > 0) { // root_finder.cu:431-447
            if (sizeof(LvlDomain) * (size_t)(2 * n_cur) > budget) {
                // (a single contact-rich query can have ~(1/tolerance)^2 live domains in level order: no
                // slice size helps then.  The depth-first work-queue kernel, or a check limit, is the way out.)
                if (len <= 4096) {
                    // Without a check limit and per-query output the result does not depend on the traversal (Appendix A.20):
                    // these queries are finished depth first by the work-queue kernel, which holds no domains in HBM at all,
                    // seeded with the TOI reached so far.  (Round 2 failed the call here: soak seed 20522, 67 faces in
                    // resting contact with a large minimum separation.)
                    if (p.max_iter < 0 && !d_per_query_toi && !d_sel && !c->scalar_f32 && walk_range) {
                        walk_range(q0, len);
                        n_cur = 0;
                        break; // (fits stays true: the slice is done)
                    }
                    throw SccdError { SCCD_E_NOMEM, "level-synchronous narrow phase: the live domains of one level exceed the memory budget" };
                }
                fits = false;
                break;
            }
            DevBuf* const cur = bufs[at];
            DevBuf* const nxt = bufs[(at + 1u) % 3u];
            DevBuf* const nx2 = bufs[(at + 2u) % 3u];
            const bool pair = sizeof(LvlDomain) * (size_t)(4 * n_cur) <= budget; // (the level behind this one at its largest)
            nxt->ensure(sizeof(LvlDomain) * (size_t)(2 * n_cur));
            if (pair) nx2->ensure(sizeof(LvlDomain) * (size_t)(4 * n_cur));
            const unsigned long long* d_len = launch_level(cur, n_cur, nullptr, nxt);
            if (pair) d_len = launch_level(nxt, 2 * n_cur, d_len, nx2);
            unsigned long long h_n = 0;
            {
                ReadBack rb(c);
                rb.add(&h_n, d_len, sizeof h_n);
                rb.sync();
            }
            n_cur = (long long)h_n;
            at = (at + (pair ? 2u : 1u)) % 3u;
        }
        if (!fits) {
            slice = std::max<long long>(4096, len / 8);
            continue; // the same queries again, fewer at a time
        }
        q0 += len;
    }
    if (d_per_query_toi) {
        hipLaunchKernelGGL(np_copy_per_query_k, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, c->stream, data,
                           n, d_per_query_toi, d_sel);
        SCCD_HIP(hipGetLastError());
    }
}

// ---- the certificate of a check limit (narrow_phase_end) -------------------------------------------------
// among the (query, time) records np_walk_k left, the smallest query number whose time is the final TOI
__global__ void np_argmin_k(const int* __restrict__ rec, unsigned n_rec, const unsigned long long* __restrict__ toi_word,
                            unsigned* __restrict__ best)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const unsigned long long t = __hip_atomic_load(toi_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)rec[3 * i + 1] == (unsigned)t && (unsigned)rec[3 * i + 2] == (unsigned)(t >> 32)) atomicMin(best, (unsigned)rec[3 * i]);
}
// that query's vertices (narrow_phase.cu:41-67), for the host
template <bool VF>
__global__ void np_fetch_query_k(const double* __restrict__ V, const int2* __restrict__ E, const int4* __restrict__ F,
                                 const int2* __restrict__ pairs, long long n, const unsigned* __restrict__ best, double* __restrict__ out)
{
    const unsigned qid = *best;
    if (qid >= (unsigned long long)n) return;
    double v[8][3];
    ti_gather<VF>(V, E, F, pairs[qid], v);
    for (int a = 0; a < 8; a++)
        for (int k = 0; k < 3; k++) out[3 * a + k] = v[a][k];
}

// both in ONE launch behind a walk kernel that recorded (run_walk: a check limit on ccd()'s step, where the host reads nothing back):
// out = {the query, the record count, its 24 coordinates}; the launch's verdict carries them to the host (VERDICT_CERT_AT)
template <bool VF>
__global__ __launch_bounds__(1024) void np_cert_k(const double* __restrict__ V, const int2* __restrict__ E, const int4* __restrict__ F,
                                                  const int2* __restrict__ pairs, const NarrowCounters* __restrict__ cnt,
                                                  const int* __restrict__ rec, unsigned cap, const unsigned long long* __restrict__ toi_word,
                                                  unsigned* __restrict__ out)
{
    __shared__ unsigned s_best;
    if (threadIdx.x == 0) s_best = 0xFFFFFFFFu;
    __syncthreads();
    const unsigned n_all = cnt->n_arg, n_rec = n_all < cap ? n_all : cap;
    const unsigned long long t = __hip_atomic_load(toi_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned best = 0xFFFFFFFFu;
    for (unsigned i = threadIdx.x; i < n_rec; i += 1024u)
        if ((unsigned)rec[3 * i + 1] == (unsigned)t && (unsigned)rec[3 * i + 2] == (unsigned)(t >> 32)) best = min(best, (unsigned)rec[3 * i]);
    if (best != 0xFFFFFFFFu) atomicMin(&s_best, best);
    __syncthreads();
    if (threadIdx.x != 0) return;
    out[0] = s_best;
    out[1] = n_all;
    if (s_best == 0xFFFFFFFFu) return;
    double v[8][3];
    ti_gather<VF>(V, E, F, pairs[s_best], v);
    double* const o = reinterpret_cast<double*>(out + 2);
    for (int a = 0; a < 8; a++)
        for (int k = 0; k < 3; k++) o[3 * a + k] = v[a][k];
}

// per-query output with a check limit: the queries whose unlimited bisection reported an impact (finite per-query TOI) ...
__global__ void np_select_finite_k(const double* __restrict__ per_query, long long n, int* __restrict__ list, unsigned* __restrict__ count)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool hit = i < n && per_query[i] < __builtin_huge_val();
    const unsigned long long m = __ballot(hit);
    __shared__ unsigned s_cnt[16], s_base; // (one atomic per block of 1024 queries, not per wave: the counter is one hot word)
    const int w = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
    if (lane_id() == 0) s_cnt[w] = (unsigned)popc64(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned sum = 0;
        for (int k = 0; k < nw; k++) {
            const unsigned v = s_cnt[k];
            s_cnt[k] = sum;
            sum += v;
        }
        s_base = sum ? atomicAdd(count, sum) : 0u;
    }
    __syncthreads();
    if (hit) list[s_base + s_cnt[w] + (unsigned)mbcnt64(m)] = (int)i;
}
// ... start again from "no impact" before they are redone in level order with the limit
__global__ void np_reset_selected_k(double* __restrict__ per_query, const int* __restrict__ list, unsigned n_sel)
{
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_sel) per_query[list[j]] = __builtin_huge_val();
}

} // namespace

#include "narrow_walk.inc"
#include "narrow_cull.inc"

// ------------------------------------------------------------------------------------------
// The FLOAT build (SCCD_OPT_SCALAR = 1, the reference's SCALABLE_CCD_USE_DOUBLE = OFF) depth first: one lane = one query at a
// time from the lane's own strided share of the list, walked to its end with the stackless walk of ti_math.hpp (NQDom / NQWalk:
// which dimension a node is split in depends on its depth alone in float, too) and the float arithmetic of ti_math_f32.hpp
// (tif_step: the operations of root_finder.cu:137-254 in float, in the reference's order).  No queues, no staging, no work
// sharing -- a float query is 30 registers, so a SIMD holds five or more waves and their divergence is covered by occupancy
// rather than by machinery -- and 20 x faster than the level-synchronous kernels that served the float build until round 4
// (14 ms per step on the 1M-triangle cloth: one launch and one read-back per level, every live domain in HBM).
// An interval [k 2^-d, (k + 1) 2^-d] and its mid-point are exact in float while k < 2^23: a query that bisects a dimension
// past level NF_MAX_LEVEL is LISTED (and dropped) and redone in level order by narrow_phase_end, like the double kernel's
// level-31 queries.  The result is the minimum over accepted domains either way (Appendix A.20): bit-equal to the oracle's
// float twin.  PQ: per-query output (a query is pruned by its own earliest impact only, root_finder.cu:297).
constexpr unsigned NF_MAX_LEVEL = 23;
// One lane walks one query alone: a query in resting contact under a large minimum separation can need 10^6 checks and more in
// float (Condition 1 is out of reach of float intervals, so every near-contact cell is bisected down to single ulps), and a lane
// does ~10^6 checks a second.  A query that has used NF_QUERY_BUDGET checks is therefore LISTED like a level-24 one, and the
// explicit-stack kernel gives a listed query NF_DFS_BUDGET checks before it raises the overflow flag and the call falls back to
// level order (which parallelises over the domains of a level -- or ends with SCCD_E_NOMEM, as it did for such scenes in round 3).
// No launch of the float build runs longer than about a second that way (soak seed 500388 ran for minutes without the budgets).
constexpr unsigned NF_QUERY_BUDGET = 1u << 16;
constexpr unsigned NF_DFS_BUDGET = 1u << 18;
constexpr int NF_REFILL_MIN = 12; // idle lanes that trigger a refill (or: nobody works) -- the refill's code runs for the whole wave
template <bool VF, int ARITH, bool PQ>
__global__ __launch_bounds__(256) void np_walk_f32_k(const double* __restrict__ V, const int2* __restrict__ E, const int4* __restrict__ F,
                                                     const int2* __restrict__ pairs, long long n, float ms, float tol, bool use_ms,
                                                     bool allow_zero_toi, NarrowCounters* __restrict__ cnt,
                                                     unsigned long long* __restrict__ per_query, int* __restrict__ ovf_list,
                                                     unsigned ovf_cap, unsigned long long* __restrict__ toi_word,
                                                     const unsigned long long* __restrict__ d_n)
{
    // d_n (may be null): the list's length is read HERE -- min(*d_n, n), n carrying the buffer's capacity: the launch was enqueued
    // before the host knew it (ccd(): right behind the pass's sweep and cull, like np_walk_k)
    if (d_n) {
        const long long n_dev = (long long)__hip_atomic_load(d_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n_dev < n) n = n_dev;
    }
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long next = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    TIQueryF q;
    NQDom dom = { 0u, 0u, 0u, 0u };
    NQWalk walk = { { 0ull, 0u }, { 0ull, 0u }, { 0ull, 0u } };
    bool has_q = false;
    float qtoi = __builtin_huge_valf();
    long long qid = 0;
    float toi = (float)toi_load(toi_word); // (every stored TOI is a float, widened: the casts are exact)
    unsigned step = 0, checks = 0, q_checks = 0;
    for (;;) {
        const bool want = !has_q && next < n;
        const unsigned long long wants = __ballot(want), busy = __ballot(has_q);
        if (wants != 0 && (busy == 0 || popc64(wants) >= NF_REFILL_MIN)) {
            if (want) {
                double vd[8][3];
                ti_gather<VF>(V, E, F, pairs[next], vd);
#pragma unroll
                for (int a = 0; a < 8; a++)
#pragma unroll
                    for (int k = 0; k < 3; k++) q.v[a][k] = (float)vd[a][k]; // vertices cast to float FIRST (ccd.cu:103-106)
                tif_tolerance<VF>(q.v, tol, q.tol);
                tif_error<VF>(q.v, use_ms, q.err);
                dom = NQDom { 0u, 0u, 0u, 0u };
                walk = NQWalk { { 0ull, 0u }, { 0ull, 0u }, { 0ull, 0u } };
                qid = next;
                qtoi = __builtin_huge_valf();
                q_checks = 0;
                has_q = true;
                next += stride;
            }
        } else if (busy == 0) {
            break; // nobody works, nobody has anything left to take
        }
        if ((++step & 15u) == 0u) toi = (float)toi_load(toi_word);
        if (has_q) {
            float lo[3], hi[3];
            {
                const unsigned d0 = dom.d & 255u, d1 = (dom.d >> 8) & 255u, d2 = (dom.d >> 16) & 255u;
                lo[0] = ldexpf((float)dom.k0, -(int)d0);
                hi[0] = lo[0] + ldexpf(1.0f, -(int)d0);
                lo[1] = ldexpf((float)dom.k1, -(int)d1);
                hi[1] = lo[1] + ldexpf(1.0f, -(int)d1);
                lo[2] = ldexpf((float)dom.k2, -(int)d2);
                hi[2] = lo[2] + ldexpf(1.0f, -(int)d2);
            }
            const TIStepF s = tif_step<VF, ARITH>(q, lo, hi, ms, tol, allow_zero_toi, PQ ? qtoi : toi);
            checks += s.checked ? 1u : 0u;
            q_checks += 1u;
            if (s.accept) {
                if (PQ && lo[0] < qtoi) {
                    qtoi = lo[0];
                    atomicMin(&per_query[qid], (unsigned long long)__double_as_longlong((double)lo[0]));
                }
                if (lo[0] < toi) {
                    toi = lo[0];
                    toi_min(toi_word, (double)lo[0]);
                }
            }
            if (s.nk >= 1) {
                const unsigned nd = dom.d + (1u << (8 * s.split));
                if (((nd >> (8 * s.split)) & 255u) > NF_MAX_LEVEL || q_checks > NF_QUERY_BUDGET) {
                    // the halves are no longer exact floats of the form k 2^-d (or: this query is too much for one lane): the
                    // whole query is handed on (narrow_phase_end)
                    const unsigned at = atomicAdd(&cnt->n_ovf, 1u);
                    if (at < ovf_cap) ovf_list[at] = (int)qid;
                    has_q = false;
                } else {
                    dom = nq_descend(walk, dom, s.split, s.nk == 2);
                }
            } else if (nqb_any(walk.pend)) {
                dom = nq_backtrack(walk, dom);
            } else {
                has_q = false;
            }
        }
    }
    unsigned long long c64 = checks;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c64 += __shfl_xor(c64, o, 64);
    if (lane_id() == 0 && c64) atomicAdd(&cnt->checks_part[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 7].n, c64);
}

// The queries np_walk_f32_k listed (their bisection passes level NF_MAX_LEVEL: scenes whose coordinates are large against
// the tolerance) depth first with an EXPLICIT stack of float (lo, hi) boxes in HBM: beyond that level an interval is no longer
// k 2^-d in float, so the boxes are carried as the reference carries them (SplitInterval, interval.cuh:18-28: mid = (lo + hi) / 2
// in float) -- tif_step on them is the reference's kernel body operation for operation.  One lane per listed query; a stack of
// NF_STACK boxes each (a depth-first stack holds at most one box per level of the tree and Condition 4 ends the tree where a
// float interval cannot be halved any more; a stack that fills up raises the overflow flag and the call falls back to level
// order).  Level order on these queries -- what round 3 did for the whole float build -- keeps every live domain of a level in
// HBM and outgrew any budget on 10 % of the soak's scaled scenes; the traversal cannot change the result without a check
// limit (Appendix A.20).
constexpr int NF_STACK = 160;
struct NFBox {
    float lo[3], hi[3];
};
template <bool VF, int ARITH, bool PQ>
__global__ __launch_bounds__(64) void np_dfs_f32_k(const double* __restrict__ V, const int2* __restrict__ E, const int4* __restrict__ F,
                                                   const int2* __restrict__ pairs, const int* __restrict__ sel, unsigned n_sel, float ms,
                                                   float tol, bool use_ms, bool allow_zero_toi, NarrowCounters* __restrict__ cnt,
                                                   unsigned long long* __restrict__ per_query, NFBox* __restrict__ stacks,
                                                   unsigned long long* __restrict__ toi_word)
{
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_sel) return;
    const int qid = sel[j];
    TIQueryF q;
    {
        double vd[8][3];
        ti_gather<VF>(V, E, F, pairs[qid], vd);
#pragma unroll
        for (int a = 0; a < 8; a++)
#pragma unroll
            for (int k = 0; k < 3; k++) q.v[a][k] = (float)vd[a][k];
    }
    tif_tolerance<VF>(q.v, tol, q.tol);
    tif_error<VF>(q.v, use_ms, q.err);
    NFBox* const st = stacks + (size_t)j * NF_STACK;
    int top = 0;
    st[top++] = NFBox { { 0.0f, 0.0f, 0.0f }, { 1.0f, 1.0f, 1.0f } };
    float toi = (float)toi_load(toi_word);
    float qtoi = PQ ? (float)__longlong_as_double((long long)per_query[qid]) : __builtin_huge_valf(); // (what the first pass found stays valid)
    unsigned step = 0;
    unsigned long long checks = 0;
    while (top > 0) {
        if ((++step & 15u) == 0u) toi = (float)toi_load(toi_word);
        if (step > NF_DFS_BUDGET) { // (one lane cannot finish this query in reasonable time: level order takes the call over)
            atomicOr(&cnt->overflow, 1u);
            break;
        }
        const NFBox d = st[--top];
        const TIStepF s = tif_step<VF, ARITH>(q, d.lo, d.hi, ms, tol, allow_zero_toi, PQ ? qtoi : toi);
        checks += s.checked ? 1u : 0u;
        if (s.accept) {
            if (PQ && d.lo[0] < qtoi) {
                qtoi = d.lo[0];
                atomicMin(&per_query[qid], (unsigned long long)__double_as_longlong((double)d.lo[0]));
            }
            if (d.lo[0] < toi) {
                toi = d.lo[0];
                toi_min(toi_word, (double)d.lo[0]);
            }
        }
        if (s.nk >= 1) {
            if (top + 2 > NF_STACK) { // (cannot happen on a tree that Condition 4 bounds; never write past the stack)
                atomicOr(&cnt->overflow, 1u);
                break;
            }
            // later half first, so that the earlier half is popped first (the walk kernels' order)
            if (s.nk == 2) {
                NFBox c2 = d;
                if (s.split == 0) c2.lo[0] = s.mid;
                else if (s.split == 1) c2.lo[1] = s.mid;
                else c2.lo[2] = s.mid;
                st[top++] = c2;
            }
            NFBox c1 = d;
            if (s.split == 0) c1.hi[0] = s.mid;
            else if (s.split == 1) c1.hi[1] = s.mid;
            else c1.hi[2] = s.mid;
            st[top++] = c1;
        }
    }
    if (checks) atomicAdd(&cnt->n_checks, checks);
}
static void run_dfs_f32(sccd_ctx* c, const NarrowParams& p, NarrowCounters* d_cnt, const int* d_sel, unsigned n_sel,
                        unsigned long long* per_query)
{
    if (n_sel == 0) return;
    c->np_scratch1.ensure(sizeof(NFBox) * (size_t)NF_STACK * n_sel);
    NFBox* const stacks = c->np_scratch1.as<NFBox>();
    const dim3 grid((n_sel + 63) / 64), block(64);
    unsigned long long* const tw = &d_cnt->toi_bits; // (seeded by the caller with the TOI reached so far, like the level-order rerun)
#define SCCD_LAUNCH_DF(VF_, AR_, PQ_)                                                                                              \
    hipLaunchKernelGGL((np_dfs_f32_k<VF_, AR_, PQ_>), grid, block, 0, c->stream, p.V, p.E, p.F, p.pairs, d_sel, n_sel, (float)p.ms, \
                       (float)p.tol, p.ms > 0, (bool)p.allow_zero_toi, d_cnt, per_query, stacks, tw)
#define SCCD_LAUNCH_DF2(VF_, AR_)                      \
    do {                                               \
        if (per_query) SCCD_LAUNCH_DF(VF_, AR_, true); \
        else SCCD_LAUNCH_DF(VF_, AR_, false);          \
    } while (0)
    if (p.is_vf) {
        if (p.arith == 1) SCCD_LAUNCH_DF2(true, 1);
        else SCCD_LAUNCH_DF2(true, 0);
    } else {
        if (p.arith == 1) SCCD_LAUNCH_DF2(false, 1);
        else SCCD_LAUNCH_DF2(false, 0);
    }
#undef SCCD_LAUNCH_DF2
#undef SCCD_LAUNCH_DF
    SCCD_HIP(hipGetLastError());
}

// THE DOUBLE BUILD'S TWIN (round 6): the queries np_walk_k LISTED -- a bisection that passes level 31 in some dimension, or a tolerance
// without an exact reciprocal: scenes measured in millimetres, tolerances many orders below the scene -- depth first on explicit
// (lo, hi) boxes, ti_step on them being the reference's kernel body operation for operation (mid = (lo + hi) / 2, Condition 4 where
// an interval cannot be halved).  Until round 6 these queries went to the level-synchronous kernels, whose live domains grow without
// bound on contact-rich ones (a resting contact under a tolerance of 1e-15 of the scene: every cell of a 2-D patch passes until the
// last level) -- SCCD_E_NOMEM where the oracle's depth-first walk returns at once.  Without a check limit the traversal cannot change
// the result (Appendix A.20).  A query that exhausts NF_DFS_BUDGET checks or its stack raises the overflow flag: level order takes
// the call over, as before.
constexpr int ND_STACK = 208; // (a depth-first stack holds at most one box per level: three dimensions down to one ulp and a margin)
struct NDBox {
    double lo[3], hi[3];
};
template <bool VF, int ARITH, bool PQ>
__global__ __launch_bounds__(64) void np_dfs_f64_k(const double* __restrict__ V, const int2* __restrict__ E, const int4* __restrict__ F,
                                                   const int2* __restrict__ pairs, const int* __restrict__ sel, unsigned n_sel, double ms,
                                                   double tol, bool use_ms, bool allow_zero_toi, NarrowCounters* __restrict__ cnt,
                                                   unsigned long long* __restrict__ per_query, NDBox* __restrict__ stacks,
                                                   unsigned long long* __restrict__ toi_word)
{
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_sel) return;
    const int qid = sel[j];
    TIQuery q;
    ti_gather<VF>(V, E, F, pairs[qid], q.v);
    ti_tolerance<VF>(q.v, tol, q.tol);
    ti_error<VF>(q.v, use_ms, q.err);
    NDBox* const st = stacks + (size_t)j * ND_STACK;
    int top = 0;
    st[top++] = NDBox { { 0.0, 0.0, 0.0 }, { 1.0, 1.0, 1.0 } };
    double toi = toi_load(toi_word);
    double qtoi = PQ ? __longlong_as_double((long long)per_query[qid]) : __builtin_huge_val(); // (what the first pass found stays valid)
    unsigned step = 0;
    unsigned long long checks = 0;
    while (top > 0) {
        if ((++step & 15u) == 0u) toi = toi_load(toi_word);
        if (step > NF_DFS_BUDGET) { // (one lane cannot finish this query in reasonable time: level order takes the call over)
            atomicOr(&cnt->overflow, 1u);
            break;
        }
        const NDBox d = st[--top];
        const TIStep s = ti_step<VF, ARITH>(q, d.lo, d.hi, ms, tol, allow_zero_toi, PQ ? qtoi : toi);
        checks += s.checked ? 1u : 0u;
        if (s.accept) {
            if (PQ && d.lo[0] < qtoi) {
                qtoi = d.lo[0];
                atomicMin(&per_query[qid], (unsigned long long)__double_as_longlong(d.lo[0]));
            }
            if (d.lo[0] < toi) {
                toi = d.lo[0];
                toi_min(toi_word, d.lo[0]);
            }
        }
        if (s.nk >= 1) {
            if (top + 2 > ND_STACK) { // (never write past the stack)
                atomicOr(&cnt->overflow, 1u);
                break;
            }
            // later half first, so that the earlier half is popped first (the walk kernels' order)
            if (s.nk == 2) {
                NDBox c2 = d;
                if (s.split == 0) c2.lo[0] = s.mid;
                else if (s.split == 1) c2.lo[1] = s.mid;
                else c2.lo[2] = s.mid;
                st[top++] = c2;
            }
            NDBox c1 = d;
            if (s.split == 0) c1.hi[0] = s.mid;
            else if (s.split == 1) c1.hi[1] = s.mid;
            else c1.hi[2] = s.mid;
            st[top++] = c1;
        }
    }
    if (checks) atomicAdd(&cnt->n_checks, checks);
}
static void run_dfs_f64(sccd_ctx* c, const NarrowParams& p, NarrowCounters* d_cnt, const int* d_sel, unsigned n_sel,
                        unsigned long long* per_query)
{
    if (n_sel == 0) return;
    c->np_scratch1.ensure(sizeof(NDBox) * (size_t)ND_STACK * n_sel);
    NDBox* const stacks = c->np_scratch1.as<NDBox>();
    const dim3 grid((n_sel + 63) / 64), block(64);
    unsigned long long* const tw = &d_cnt->toi_bits; // (seeded by the caller with the TOI reached so far, like the level-order rerun)
#define SCCD_LAUNCH_DD(VF_, AR_, PQ_)                                                                                      \
    hipLaunchKernelGGL((np_dfs_f64_k<VF_, AR_, PQ_>), grid, block, 0, c->stream, p.V, p.E, p.F, p.pairs, d_sel, n_sel, p.ms, \
                       p.tol, p.ms > 0, (bool)p.allow_zero_toi, d_cnt, per_query, stacks, tw)
#define SCCD_LAUNCH_DD2(VF_, AR_)                      \
    do {                                               \
        if (per_query) SCCD_LAUNCH_DD(VF_, AR_, true); \
        else SCCD_LAUNCH_DD(VF_, AR_, false);          \
    } while (0)
    if (p.is_vf) {
        if (p.arith == 1) SCCD_LAUNCH_DD2(true, 1);
        else SCCD_LAUNCH_DD2(true, 0);
    } else {
        if (p.arith == 1) SCCD_LAUNCH_DD2(false, 1);
        else SCCD_LAUNCH_DD2(false, 0);
    }
#undef SCCD_LAUNCH_DD2
#undef SCCD_LAUNCH_DD
    SCCD_HIP(hipGetLastError());
}

// ovf_list / ovf_cap: where queries beyond level NF_MAX_LEVEL are listed (always given: narrow_phase_end redoes them in level order)
// d_n / capacity: the list's length on the device (np_walk_f32_k); vx: the pass's verdict behind the launch (np_verdict_k), as run_walk
static void run_walk_f32(sccd_ctx* c, const NarrowParams& p, NarrowCounters* d_cnt, long long n, unsigned long long* per_query,
                         int* ovf_list, unsigned ovf_cap, const unsigned long long* d_n = nullptr, long long capacity = 0,
                         const VerdictExtras* vx = nullptr)
{
    if (d_n) n = std::min<long long>(capacity, (1ll << 31) - 4097); // (the kernel takes min(*d_n, n))
    SCCD_REQUIRE(n < (1ll << 31) - 4096, "narrow phase: at most 2^31 - 4097 queries per launch");
    // (lane-per-query streams: enough waves to fill the chip several times over -- a wave is as slow as its slowest lane)
    const long long want_blocks = d_n ? (long long)c->num_cus * 16 : (n + 255) / 256;
    const int blocks = (int)std::max<long long>(1, std::min<long long>(want_blocks, (long long)c->num_cus * 16));
    const dim3 grid((unsigned)blocks), block(256);
    unsigned long long* const tw = p.toi_word ? p.toi_word : &d_cnt->toi_bits;
#define SCCD_LAUNCH_NF(VF_, AR_, PQ_)                                                                                               \
    hipLaunchKernelGGL((np_walk_f32_k<VF_, AR_, PQ_>), grid, block, 0, c->stream, p.V, p.E, p.F, p.pairs, n, (float)p.ms, (float)p.tol, \
                       p.ms > 0, (bool)p.allow_zero_toi, d_cnt, per_query, ovf_list, ovf_cap, tw, d_n)
#define SCCD_LAUNCH_NF2(VF_, AR_)                  \
    do {                                           \
        if (per_query) SCCD_LAUNCH_NF(VF_, AR_, true); \
        else SCCD_LAUNCH_NF(VF_, AR_, false);      \
    } while (0)
    if (p.is_vf) {
        if (p.arith == 1) SCCD_LAUNCH_NF2(true, 1);
        else SCCD_LAUNCH_NF2(true, 0);
    } else {
        if (p.arith == 1) SCCD_LAUNCH_NF2(false, 1);
        else SCCD_LAUNCH_NF2(false, 0);
    }
#undef SCCD_LAUNCH_NF2
#undef SCCD_LAUNCH_NF
    SCCD_HIP(hipGetLastError());
    if (vx && c->verdict_dev && !per_query && lab_env().np_diag == 0) { // (the pass's verdict in pinned memory: run_walk)
        c->verdict_seq += 1;
        c->verdict_armed = true;
        hipLaunchKernelGGL(np_verdict_k, dim3(1), dim3(64), 0, c->stream, d_cnt, tw, 0ull, (unsigned long long*)nullptr,
                           reinterpret_cast<unsigned*>(c->verdict_dev), c->verdict_seq, 0, *vx);
        SCCD_HIP(hipGetLastError());
    }
}

// counters = {toi, zeros}: a kernel whose arguments carry the TOI.  (It was an upload from the pinned mirror: a copy kernel that
// reads host memory, 7 us -- and one that does not fit beside resident narrow-phase waves, so the helper's upload for the
// edge-edge launch sat 85 us in its queue.  Eight vector registers at most: tests/test_kernel_resources.py.)
__global__ __launch_bounds__(256) void np_counters_init_k(unsigned long long* __restrict__ cnt, int words, unsigned long long toi_bits)
{
    for (int k = threadIdx.x; k < words; k += 256) cnt[k] = k == 0 ? toi_bits : 0ull;
}
// *dst = min(*dst, *src) on c->stream: one pass's running TOI seeded with another's result, on the device (ccd() with a check limit:
// the second pass starts from the first one's result, ccd.cu:125-143, without the host having seen it)
__global__ void np_seed_word_k(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src)
{
    atomicMin(dst, __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
void narrow_seed_word(sccd_ctx* c, NarrowCounters* dst, const NarrowCounters* src)
{
    hipLaunchKernelGGL(np_seed_word_k, dim3(1), dim3(1), 0, c->stream, &dst->toi_bits, &src->toi_bits);
    SCCD_HIP(hipGetLastError());
}
void narrow_counters_upload(sccd_ctx* c, NarrowCounters* d_cnt, double toi)
{
    static_assert(offsetof(NarrowCounters, toi_bits) == 0 && sizeof(NarrowCounters) % 8 == 0, "np_counters_init_k: the TOI is word 0");
    unsigned long long bits;
    std::memcpy(&bits, &toi, 8);
    hipLaunchKernelGGL(np_counters_init_k, dim3(1), dim3(256), 0, c->stream, reinterpret_cast<unsigned long long*>(d_cnt),
                       (int)(sizeof(NarrowCounters) / 8), bits);
    SCCD_HIP(hipGetLastError());
    c->np_uploaded = true;
    c->np_uploaded_toi = toi;
}

// begin: upload the counters (unless narrow_counters_upload() already did) and launch; end: read the counters
// back, handle the work-queue kernel's overflow flags, hand the TOI over.  (Split so that a caller can enqueue
// unrelated work on another stream in between; narrow_phase_run() is the two back to back.)
bool narrow_uses_walk_kernel(const sccd_ctx* c, const NarrowParams& p, bool per_query)
{
    (void)per_query;
    // (check limits below SCCD_QUEUE_MIN_MAX_ITER cut queries off as a rule: straight to level order)
    // (per-query output with a limit: every query is pruned by its OWN earliest impact only (root_finder.cu:297), so the
    // queries are independent -- the fast kernel finds those that report an impact at all, narrow_phase_end redoes exactly those
    // in level order with the limit; any limit, small ones too)
    // (the float build: its own depth-first kernel, np_walk_f32_k -- without a check limit; limits stay in level order there)
    if (c->scalar_f32) return c->narrow_algo != 1 && p.max_iter < 0;
    return !(c->narrow_algo == 1
             || (p.max_iter >= 0 && (c->limit_level_order || (!per_query && p.max_iter < SCCD_QUEUE_MIN_MAX_ITER))));
}

CullSlabs narrow_cull_slabs(const sccd_ctx* c, const NarrowParams& p, double toi)
{
    CullSlabs sl;
    sl.t_end = (toi > 0.0 && toi < 1.0) ? toi : 1.0; // (a start from 0 launches nothing: narrow_phase_begin)
    sl.two = lab_env().cull_slabs && narrow_start_toi(c, p, toi, false) != toi; // (a list per half: exactly when the launches will be two)
    sl.t_mid = 0.5;
    if (!lab_env().cull_slabs) sl.t_end = 1.0; // (SCCD_CULL_SLABS=0: the whole step, one list -- round 5's first cull)
    return sl;
}

double narrow_start_toi(const sccd_ctx* c, const NarrowParams& p, double toi, bool per_query)
{
    // (exactly the launches narrow_phase_begin serves with the plain walk kernel; diagnostics builds count one launch)
    // (round 6: a check limit the fast kernel serves too -- the kernel runs WITHOUT the limit either way and the certificate is about
    // the TOI the CALL started with, whichever way the kernel got to its earliest accept: narrow_phase_begin)
    const bool limit_ok = p.max_iter < 0 || (!c->limit_level_order && p.max_iter >= SCCD_QUEUE_MIN_MAX_ITER);
    const bool two = c->two_halves && !c->two_halves_off && !per_query && limit_ok && !c->scalar_f32 && c->narrow_algo != 1 && lab_env().np_diag == 0 && toi > 0.5;
    return two ? 0.5 : toi;
}

// d_n / capacity: the list's length is still being made on the device when this is called (ccd(): the launch goes into the stream
// right behind the pass's sweep and cull) -- p.n_pairs is ignored, the walk kernel takes min(*d_n, capacity); plain launches of
// the double build only (no check limit, no per-query output), the caller's business
// vx (may be null; with d_n only): the pass's verdict is wanted in pinned memory behind the first launch, with these extras (run_walk)
void narrow_phase_begin(sccd_ctx* c, const NarrowParams& p, NarrowCounters* d_cnt, const double* h_toi_inout,
                        double* d_per_query_toi, const unsigned long long* d_n, long long capacity, const VerdictExtras* vx)
{
    // toi is in/out and must be >= 0 (narrow_phase.cu:126)
    SCCD_REQUIRE(*h_toi_inout >= 0, "narrow_phase: toi must be >= 0");
    c->np_limit_fast = false; // (context-sticky between begin and end: a begin whose end never came must not leave it set)
    c->np_pq_limit = false;
    c->verdict_armed = false; // (... nor an early verdict of a launch nobody ended: its word is an OLD call's -- ADVICE r05)
    // pinned mirror: [8 KB, 12 KB) the counters handed to the caller, [12 KB, 16 KB) the upload source
    // (two halves of time, narrow_walk.inc: the counters start from 0.5, the walk kernel's second launch goes on from the caller's TOI)
    const double start_toi = narrow_start_toi(c, p, *h_toi_inout, d_per_query_toi != nullptr);
    const double two_halves_from = start_toi != *h_toi_inout ? *h_toi_inout : 0.0;
    if (!(c->np_uploaded && std::memcmp(&c->np_uploaded_toi, &start_toi, 8) == 0)) narrow_counters_upload(c, d_cnt, start_toi);
    c->np_uploaded = false;
    // the reference's outer loop runs only while toi > 0 (narrow_phase.cu:136); in the
    // per-query build the guard is absent (:138)
    const bool run = (*h_toi_inout > 0) || d_per_query_toi != nullptr;
    const long long n = p.n_pairs;
    if (d_n) {
        SCCD_REQUIRE(!d_per_query_toi && narrow_uses_walk_kernel(c, p, false),
                     "narrow_phase: a list whose length is on the device is served by the plain walk kernels only");
        if (run) {
            ProfScope ps(c, p.is_vf ? SCCD_PROF_NARROW_VF : SCCD_PROF_NARROW_EE);
            if (c->scalar_f32) { // (no limit: narrow_uses_walk_kernel) the float build's depth-first kernel; it lists what it cannot hold
                const unsigned cap = (unsigned)std::min<long long>(std::max<long long>(capacity, 1), 1 << 20);
                c->np_scratch3_ovf.ensure(sizeof(int) * (size_t)cap);
                run_walk_f32(c, p, d_cnt, 0, nullptr, c->np_scratch3_ovf.as<int>(), cap, d_n, capacity, vx);
            } else if (p.max_iter >= 0) { // a check limit: the fast kernel WITHOUT it, recording who lowered the TOI (below; narrow_phase_end: the certificate)
                const unsigned cap = 1u << 20; // (the list's length is not known here: room for the most records a certificate looks at)
                c->np_scratch3_ovf.ensure(sizeof(int) * 4 * (size_t)cap + 256);
                NarrowParams pn = p;
                pn.max_iter = -2;
                c->np_toi_init = *h_toi_inout;
                c->np_limit_cap = cap;
                run_walk(c, pn, d_cnt, 0, nullptr, c->np_scratch3_ovf.as<int>(), cap, d_n, capacity, two_halves_from, vx);
                c->np_limit_fast = true;
            } else {
                run_walk(c, p, d_cnt, 0, nullptr, nullptr, 0, d_n, capacity, two_halves_from, vx);
            }
        }
        return;
    }
    // (counters that were started from 0.5 for the two halves of time get their launches even for an empty list: the kernel
    // between the two puts the caller's TOI back)
    SCCD_REQUIRE(!p.second.src || two_halves_from > 0.5 || start_toi <= 0.5, "narrow_phase: a list per half of time, but one launch over the whole step");
    if (run && (n > 0 || two_halves_from > 0.5)) {
        ProfScope ps(c, p.is_vf ? SCCD_PROF_NARROW_VF : SCCD_PROF_NARROW_EE);
        // A check limit (max_iter >= 0: the IPC Toolkit passes 10^7) is defined in the reference's LEVEL ORDER: it counts the
        // domains of a query as the breadth-first launches pop them (root_finder.cu:287-305), several times what a
        // depth-first walk with pruning checks.  Round 2 ran every such call on the level-synchronous kernels: 35.6 ms for
        // the 1M-triangle cloth against 1.35 ms without a limit.  Now the fast kernel runs WITHOUT the limit and
        // narrow_phase_end proves that the limit could not have changed the answer (ti_census.cpp: "the certificate");
        // only where that proof fails -- and for limits below 4096, per-query output and the float build -- the call is
        // (re)done in level order, bit-equal to the oracle's level-order restatement either way.
        const bool level_sync = !narrow_uses_walk_kernel(c, p, d_per_query_toi != nullptr);
        if (level_sync) {
            // queries whose levels outgrow the memory budget are finished depth first (run_level_sync: only where the result
            // cannot depend on the traversal): the TOI reached so far seeds the work-queue kernel, its counters start afresh
            auto walk_range = [&](long long first, long long count) {
                NarrowCounters hc;
                SCCD_HIP(hipMemcpyAsync(&hc, d_cnt, sizeof hc, hipMemcpyDeviceToHost, c->stream));
                SCCD_HIP(hipStreamSynchronize(c->stream));
                unsigned long long checks = hc.n_checks;
                for (int k = 0; k < 8; k++) checks += hc.checks_part[k].n;
                NarrowCounters* const h_up = reinterpret_cast<NarrowCounters*>(c->h_scalars.as<char>() + 12288);
                std::memset(h_up, 0, sizeof(NarrowCounters));
                h_up->toi_bits = hc.toi_bits;
                h_up->n_checks = checks;
                SCCD_HIP(hipMemcpyAsync(d_cnt, h_up, sizeof(NarrowCounters), hipMemcpyHostToDevice, c->stream));
                SCCD_HIP(hipStreamSynchronize(c->stream)); // (the pinned source is reused by the next upload)
                NarrowParams pr = p;
                pr.pairs = p.pairs + first;
                pr.n_pairs = count;
                run_walk(c, pr, d_cnt, count, nullptr);
            };
            if (p.is_vf) run_level_sync<true>(c, p, d_cnt, n, d_per_query_toi, nullptr, 0, walk_range);
            else run_level_sync<false>(c, p, d_cnt, n, d_per_query_toi, nullptr, 0, walk_range);
        } else {
            if (d_per_query_toi) { // every query starts at +inf (narrow_phase.cu:70)
                hipLaunchKernelGGL(np_fill_u64_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                                   reinterpret_cast<unsigned long long*>(d_per_query_toi), n, 0x7FF0000000000000ull);
                SCCD_HIP(hipGetLastError());
            }
            if (c->scalar_f32) { // (no limit: narrow_uses_walk_kernel) the float build's depth-first kernel; it lists what it cannot hold
                const unsigned cap = (unsigned)std::min<long long>(n, 1 << 20);
                c->np_scratch3_ovf.ensure(sizeof(int) * (size_t)cap);
                run_walk_f32(c, p, d_cnt, n, reinterpret_cast<unsigned long long*>(d_per_query_toi), c->np_scratch3_ovf.as<int>(), cap);
            } else if (p.max_iter >= 0 && d_per_query_toi) {
                // per-query output with a limit: the fast kernel WITHOUT the limit (bookkeeping instantiation, listing queries
                // beyond level 31 itself); narrow_phase_end redoes the queries that report an impact with the limit
                const unsigned cap = (unsigned)std::min<long long>(n, 1 << 20);
                c->np_scratch3_ovf.ensure(sizeof(int) * (size_t)cap);
                NarrowParams pn = p;
                pn.max_iter = -1;
                c->np_toi_init = *h_toi_inout;
                run_walk(c, pn, d_cnt, n, reinterpret_cast<unsigned long long*>(d_per_query_toi), c->np_scratch3_ovf.as<int>(), cap);
                c->np_pq_limit = true;
            } else if (p.max_iter >= 0) { // the fast kernel WITHOUT the limit, recording who lowered the TOI (behind the overflow list)
                const unsigned cap = (unsigned)std::min<long long>(std::max<long long>(n, 1024), 1 << 20);
                c->np_scratch3_ovf.ensure(sizeof(int) * 4 * (size_t)cap + 256);
                NarrowParams pn = p;
                pn.max_iter = -2;
                c->np_toi_init = *h_toi_inout;
                c->np_limit_cap = cap;
                run_walk(c, pn, d_cnt, n, nullptr, c->np_scratch3_ovf.as<int>(), cap, nullptr, 0, two_halves_from);
                c->np_limit_fast = true; // (only once the launch is enqueued)
            } else if (d_per_query_toi) { // (bookkeeping kernels: they can list queries beyond level 31 themselves)
                const unsigned cap = (unsigned)std::min<long long>(n, 1 << 20);
                c->np_scratch3_ovf.ensure(sizeof(int) * (size_t)cap);
                run_walk(c, p, d_cnt, n, reinterpret_cast<unsigned long long*>(d_per_query_toi), c->np_scratch3_ovf.as<int>(), cap);
            } else {
                run_walk(c, p, d_cnt, n, nullptr, nullptr, 0, nullptr, 0, two_halves_from);
            }
        }
    }
}

// The verdict a launch of this context armed (run_walk: sccd_ctx::verdict_seq is its number): waits for its word and returns the pinned
// buffer -- the pass's counters at 0, the extras where the caller put them -- or nullptr if the stream drained or failed without it.
// Polls (like ReadBack::sync); may be called again for the same verdict: the word stays.
const char* narrow_verdict_wait(sccd_ctx* c)
{
    const char* const from = c->verdict.as<char>();
    const unsigned long long* const word = reinterpret_cast<const unsigned long long*>(from + 2048);
    const unsigned long long want = c->verdict_seq;
    for (unsigned spins = 1; __atomic_load_n(word, __ATOMIC_ACQUIRE) != want; spins++) {
        __builtin_ia32_pause();
        if ((spins & 0xFFFFu) != 0) continue;
        const hipError_t e = hipStreamQuery(c->stream); // (a stream that drained or failed without the word: the caller's read-back finds out)
        if (e == hipErrorNotReady) continue;
        if (__atomic_load_n(word, __ATOMIC_ACQUIRE) != want) return nullptr;
        break;
    }
    return from;
}

void narrow_phase_end(sccd_ctx* c, const NarrowParams& p, NarrowCounters* d_cnt, double* h_toi_inout,
                      double* d_per_query_toi)
{
    NarrowCounters h;
    const long long n = p.n_pairs;
    bool have = false;
    const char* verdict_from = nullptr;
    const bool cert_in_verdict = c->np_cert_in_verdict;
    c->np_cert_in_verdict = false;
    if (c->verdict_armed) {
        // THE EARLY VERDICT (sccd_ctx::verdict, np_verdict_k): the first launch's counters are on the host the moment the kernel
        // behind it has run.  One launch, or a first half that found its impact: that is the pass's result -- what is still enqueued
        // behind that kernel finds nothing to do and is not waited for (the stream orders the next call behind it).  A second half
        // that has work: the whole stream, as before.
        c->verdict_armed = false;
        c->host_waits += 1;
        const char* const from = narrow_verdict_wait(c);
        verdict_from = from;
        if (from) {
            std::memcpy(&h, from, sizeof h);
            have = h.second_go == 0u;
            unsigned long long t_end; // (np_verdict_k's stamp: the step's end if this verdict is what the host returns on)
            std::memcpy(&t_end, from + 2056, sizeof t_end);
            if (have && t_end > c->step_t_last) c->step_t_last = t_end;
        }
    }
    if (!have) {
        // (a launch that kept its running TOI in another launch's word: that word is its result so far -- read in the same
        // round trip; a second, blocking copy here was 25 us at the end of every ccd() step)
        unsigned long long toi_elsewhere = 0;
        ReadBack rb(c);
        rb.add(&h, d_cnt, sizeof h);
        if (p.toi_word) rb.add(&toi_elsewhere, p.toi_word, 8);
        rb.sync();
        if (p.toi_word) h.toi_bits = toi_elsewhere;
    }
    for (int k = 0; k < 8; k++) h.n_checks += h.checks_part[k].n; // (the work-queue kernel counts in stripes)
    if (c->np_limit_fast) {
        // ---- the certificate (ti_census.cpp).  T* = h.toi_bits is the earliest accept of the full bisection trees; the
        // limited level-order run of the reference returns T* too if the query that holds it, bisected ALONE in level
        // order WITH the limit, still reaches it.  Otherwise -- or if anything about the fast pass was irregular -- the
        // call is redone in level order from the TOI it started with.
        c->np_limit_fast = false;
        const double toi_init = c->np_toi_init;
        double t_star;
        std::memcpy(&t_star, &h.toi_bits, 8);
        const unsigned cap = c->np_limit_cap; // (what the launch was given room for: narrow_phase_begin)
        bool certified = false;
        if (!h.overflow && !h.n_ovf) {
            if (!(t_star < toi_init)) certified = true; // nothing was accepted below the TOI the call started with
            else if (h.n_arg <= cap && h.n_arg > 0 && have && cert_in_verdict && verdict_from) {
                // (the holder and its coordinates came with the verdict: np_cert_k ran behind the launch -- no read-back)
                unsigned best = 0xFFFFFFFFu;
                double v[8][3];
                std::memcpy(&best, verdict_from + VERDICT_CERT_AT, sizeof best);
                std::memcpy(&v[0][0], verdict_from + VERDICT_CERT_AT + 8, sizeof v);
                if (best < (unsigned long long)n) {
                    bool gave_up = false;
                    const double alone = ti_census_level_order(v, p.is_vf, p.arith, p.ms, p.tol, p.max_iter, p.allow_zero_toi, toi_init,
                                                               /*max_live=*/1 << 22, &gave_up);
                    certified = !gave_up && alone == t_star;
                }
            } else if (h.n_arg <= cap && h.n_arg > 0) {
                int* const rec = c->np_scratch3_ovf.as<int>() + cap;
                unsigned* const d_best = reinterpret_cast<unsigned*>(c->np_scratch3_ovf.as<int>() + 4 * (size_t)cap);
                double* const d_v = reinterpret_cast<double*>(d_best + 2);
                SCCD_HIP(hipMemsetAsync(d_best, 0xFF, sizeof(unsigned), c->stream));
                hipLaunchKernelGGL(np_argmin_k, dim3((h.n_arg + 255) / 256), dim3(256), 0, c->stream, rec, h.n_arg,
                                   p.toi_word ? p.toi_word : &d_cnt->toi_bits, d_best);
                if (p.is_vf) hipLaunchKernelGGL(np_fetch_query_k<true>, dim3(1), dim3(1), 0, c->stream, p.V, p.E, p.F, p.pairs, n, d_best, d_v);
                else hipLaunchKernelGGL(np_fetch_query_k<false>, dim3(1), dim3(1), 0, c->stream, p.V, p.E, p.F, p.pairs, n, d_best, d_v);
                SCCD_HIP(hipGetLastError());
                unsigned best = 0xFFFFFFFFu;
                double v[8][3];
                {
                    ReadBack rb(c);
                    rb.add(&best, d_best, sizeof best);
                    rb.add(&v[0][0], d_v, sizeof v);
                    rb.sync();
                }
                if (best < (unsigned long long)n) {
                    bool gave_up = false;
                    const double alone = ti_census_level_order(v, p.is_vf, p.arith, p.ms, p.tol, p.max_iter, p.allow_zero_toi, toi_init,
                                                               /*max_live=*/1 << 22, &gave_up);
                    certified = !gave_up && alone == t_star;
                }
            }
        }
        if (!certified) {
            NarrowCounters h2;
            std::memset(&h2, 0, sizeof h2);
            std::memcpy(&h2.toi_bits, &toi_init, 8);
            SCCD_HIP(hipMemcpyAsync(d_cnt, &h2, sizeof h2, hipMemcpyHostToDevice, c->stream));
            if (p.is_vf) run_level_sync<true>(c, p, d_cnt, n, nullptr);
            else run_level_sync<false>(c, p, d_cnt, n, nullptr);
            SCCD_HIP(hipMemcpyAsync(&h, d_cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
            SCCD_HIP(hipStreamSynchronize(c->stream));
            if (h.overflow) throw SccdError { SCCD_E_OVERFLOW, "narrow phase: work queue capacity exhausted" };
        }
        h.overflow = 0;
        h.n_ovf = 0;
    }
    if (lab_env().np_diag && h.wave_steps)
        std::fprintf(stderr, "[sccd np] n=%lld checks=%llu wave_steps=%llu (waves %llu, mean %.1f, longest %llu) lane_util=%.3f refill_execs=%llu steals=%llu pops=%llu checked ahead=%llu\n",
                     n, h.n_checks, h.wave_steps, h.waves_run, (double)h.wave_steps / (double)std::max<unsigned long long>(1, h.waves_run),
                     h.max_wave_steps, (double)h.lane_steps / (64.0 * (double)h.wave_steps), h.refill_execs, h.steals,
                     h.pops_reg, h.checked_ahead);
    if (lab_env().np_diag && h.wave_steps)
        std::fprintf(stderr, "[sccd np] tail (after a wave's query stream ran dry): %llu of %llu wave-steps, longest %llu steps / %.0f of %.0f kcycles; mean tail %.0f kcycles\n",
                     h.tail_steps, h.wave_steps, h.max_tail_steps, h.max_tail_cycles / 1e3, h.max_total_cycles / 1e3,
                     (double)h.sum_tail_cycles / 1e3 / (double)std::max<unsigned long long>(1, h.waves_run));
    if (lab_env().np_diag && h.wave_steps) {
        std::fprintf(stderr, "[sccd np] waves by steps (x16):");
        for (int k = 0; k < 16; k++) std::fprintf(stderr, " %llu", h.wave_hist[k]);
        std::fprintf(stderr, " | mean steps per XCD:");
        for (int k = 0; k < 8; k++) std::fprintf(stderr, " %.0f(%llu)", (double)h.xcd_steps[k] / (double)std::max<unsigned long long>(1, h.xcd_waves[k]), h.xcd_waves[k]);
        std::fprintf(stderr, "\n");
    }
    if (lab_env().np_diag >= 3 && h.tail_steps) h.wave_steps = h.tail_steps; // (per tail step)
    if (lab_env().np_diag && h.stamp[4])
        std::fprintf(stderr, "[sccd np] cycles/wave-step: toi=%.0f pop=%.0f steal=%.0f refill=%.0f check=%.0f push=%.0f | cycles/ingest: total=%.0f gather wait=%.0f\n",
                     (double)h.stamp[0] / h.wave_steps, (double)h.stamp[1] / h.wave_steps, (double)h.stamp[2] / h.wave_steps,
                     (double)h.stamp[3] / h.wave_steps, (double)h.stamp[4] / h.wave_steps, (double)h.stamp[5] / h.wave_steps,
                     (double)h.stamp[6] / (double)std::max<unsigned long long>(1, h.refill_execs),
                     (double)h.stamp[7] / (double)std::max<unsigned long long>(1, h.refill_execs));
    if (h.overflow || h.n_ovf) {
        // Some bisection cannot be held as (numerator, level <= 31) -- tolerance / (3 x extent) below 2^-31, as in scenes
        // measured in millimetres -- or a check limit was reached in the opt-in fast mode.  What the kernel found so far
        // stays valid (accepted domains only ever lower a TOI).  First choice: redo ONLY the queries concerned in level
        // order.  A launch without the per-query bookkeeping could not name them: it is repeated with bookkeeping and an
        // overflow list (rare: one extra pass of the fast kernel instead of a level-order pass over everything, which
        // is several times slower and keeps every live domain of a level in HBM).
        if (c->np_peer_stream) { // a launch of another context shares this TOI word: let it finish before the counters are reset
            SCCD_HIP(hipStreamSynchronize(c->np_peer_stream));
            SCCD_HIP(hipMemcpyAsync(&h.toi_bits, &d_cnt->toi_bits, 8, hipMemcpyDeviceToHost, c->stream));
            SCCD_HIP(hipStreamSynchronize(c->stream));
        }
        // THE SECOND HALF'S OWN LIST (two halves of time with a cull per slab: run_walk).  Its launch was a plain one over
        // p.second.kept -- pairs the first half's list need not hold -- and a query it dropped raised the same flag: that list is
        // redone like the pass's own, from what is known by then (the fallback looked at p.pairs only, and a query
        // that lived in the second list alone was never bisected to the end).
        const bool second_ran = p.second.src != nullptr && h.second_go != 0u;
        const bool no_list_yet = (h.overflow & NQ_OVF_INTERVAL) != 0u;
        // one list of the pass: [bookkeeping pass that names the queries] -> those queries in level order -> (failing that) the whole
        // list in level order; h carries the TOI reached and the checks counted so far in, and out
        auto redo_list = [&](const NarrowParams& pl, long long nl, bool list_ready) {
            if (nl <= 0) return;
            // (room for the list: a query is listed by EVERY lane that holds a part of it when the part passes level 31 -- in the tail
            // of a launch a deep query's deferred halves are spread over the wave's idle lanes, and a scene whose every query goes deep
            // -- static edges under the reference's edge-edge tolerances: tests, "static" at 1e-12 -- listed each of its 4,000 queries
            // thirty times over: with room for one entry per query the list overflowed and the call went to level order, and out of memory)
            // (a list that IS there was made by the call's own launch, in the room narrow_phase_begin gave it: the buffer must not be
            // grown now -- DevBuf::ensure does not keep contents -- and holds min(nl, 2^20) entries)
            const unsigned cap = list_ready ? (unsigned)std::min<long long>(nl, 1 << 20) : (unsigned)std::min<long long>(64 * nl + 4096, 1 << 22);
            auto level_all = [&]() {
                NarrowCounters h2;
                std::memset(&h2, 0, sizeof h2);
                h2.toi_bits = h.toi_bits; // (what was found so far stays valid)
                SCCD_HIP(hipMemcpyAsync(d_cnt, &h2, sizeof h2, hipMemcpyHostToDevice, c->stream));
                SCCD_HIP(hipStreamSynchronize(c->stream)); // (h2 is on the stack)
                if (pl.is_vf) run_level_sync<true>(c, pl, d_cnt, nl, d_per_query_toi);
                else run_level_sync<false>(c, pl, d_cnt, nl, d_per_query_toi);
            };
            bool done = false;
            if (!list_ready) c->np_scratch3_ovf.ensure(sizeof(int) * (size_t)cap);
            int* d_list = c->np_scratch3_ovf.as<int>();
            unsigned long long checks_so_far = h.n_checks;
            unsigned ovf = h.overflow, n_ovf = h.n_ovf;
            if (!list_ready) { // no usable list yet: the bookkeeping pass, seeded with the TOI reached
                NarrowCounters h2;
                std::memset(&h2, 0, sizeof h2);
                h2.toi_bits = h.toi_bits;
                SCCD_HIP(hipMemcpyAsync(d_cnt, &h2, sizeof h2, hipMemcpyHostToDevice, c->stream));
                SCCD_HIP(hipStreamSynchronize(c->stream)); // (h2 is on the stack)
                run_walk(c, pl, d_cnt, nl, reinterpret_cast<unsigned long long*>(d_per_query_toi), d_list, cap);
                SCCD_HIP(hipMemcpyAsync(&h, d_cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
                if (pl.toi_word) SCCD_HIP(hipMemcpyAsync(&h.toi_bits, pl.toi_word, 8, hipMemcpyDeviceToHost, c->stream));
                SCCD_HIP(hipStreamSynchronize(c->stream));
                for (int k = 0; k < 8; k++) h.n_checks += h.checks_part[k].n;
                h.n_checks += checks_so_far;
                checks_so_far = h.n_checks;
                ovf = h.overflow;
                n_ovf = h.n_ovf;
            }
            if (!ovf && n_ovf <= cap) {
                if (n_ovf > 0) {
                    std::vector<int> ids(n_ovf);
                    SCCD_HIP(hipMemcpyAsync(ids.data(), d_list, sizeof(int) * ids.size(), hipMemcpyDeviceToHost, c->stream));
                    SCCD_HIP(hipStreamSynchronize(c->stream));
                    std::sort(ids.begin(), ids.end()); // (a query shared among lanes is listed by each of them)
                    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
                    SCCD_HIP(hipMemcpyAsync(d_list, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice, c->stream));
                    NarrowCounters h2;
                    std::memset(&h2, 0, sizeof h2);
                    h2.toi_bits = h.toi_bits;
                    SCCD_HIP(hipMemcpyAsync(d_cnt, &h2, sizeof h2, hipMemcpyHostToDevice, c->stream));
                    SCCD_HIP(hipStreamSynchronize(c->stream)); // (h2 and ids are on the stack)
                    if (pl.max_iter < 0) { // (the list holds at most 2^20 queries: 4 / 10 GB of stacks at the very most)
                        // the listed queries depth first on explicit boxes (np_dfs_f32_k / np_dfs_f64_k) -- no level of theirs
                        // has to fit anywhere; a query that is too much for one lane raises the overflow flag (level order below)
                        if (c->scalar_f32) run_dfs_f32(c, pl, d_cnt, d_list, (unsigned)ids.size(), reinterpret_cast<unsigned long long*>(d_per_query_toi));
                        else run_dfs_f64(c, pl, d_cnt, d_list, (unsigned)ids.size(), reinterpret_cast<unsigned long long*>(d_per_query_toi));
                    } else if (pl.is_vf) run_level_sync<true>(c, pl, d_cnt, nl, d_per_query_toi, d_list, (long long)ids.size());
                    else run_level_sync<false>(c, pl, d_cnt, nl, d_per_query_toi, d_list, (long long)ids.size());
                    SCCD_HIP(hipMemcpyAsync(&h, d_cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
                    SCCD_HIP(hipStreamSynchronize(c->stream));
                    h.n_checks += checks_so_far;
                }
                done = !h.overflow;
            }
            if (!done) {
                level_all();
                SCCD_HIP(hipMemcpyAsync(&h, d_cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
                SCCD_HIP(hipStreamSynchronize(c->stream));
                h.n_checks += checks_so_far;
                if (h.overflow) throw SccdError { SCCD_E_OVERFLOW, "narrow phase: work queue capacity exhausted" };
            }
            h.overflow = 0;
            h.n_ovf = 0;
        };
        long long n_second = 0;
        if (second_ran) { // (how long the list behind the first half's got: its length stayed on the device -- run_walk)
            unsigned long long k = 0;
            SCCD_HIP(hipMemcpyAsync(&k, p.second.d_n_kept, sizeof k, hipMemcpyDeviceToHost, c->stream));
            SCCD_HIP(hipStreamSynchronize(c->stream));
            n_second = (long long)std::min<unsigned long long>(k, (unsigned long long)std::max<long long>(p.second.capacity, 0));
        }
        NarrowParams p1 = p; // (the fallback's launches are single ones over a whole list: no second half of their own)
        p1.second = NarrowParams::SecondHalf();
        redo_list(p1, n, !no_list_yet);
        if (second_ran && no_list_yet && n_second > 0) {
            NarrowParams p2 = p1;
            p2.pairs = p.second.kept;
            p2.n_pairs = n_second;
            redo_list(p2, n_second, false);
        }
    }
    if (c->np_pq_limit && d_per_query_toi && n > 0) {
        // ---- per-query output with a check limit.  The reference counts the domains a query pops in LEVEL order and drops the
        // query past the limit (root_finder.cu:287-305); with per-query output a query is pruned by its own earliest impact only
        // (:297), so its bisection does not depend on any other query.  The fast kernel ran every query WITHOUT the limit: a
        // query that reported no impact has no accepted domain at all, with or without a limit; the queries that did report
        // one -- a few thousand of millions -- are redone, alone among themselves, by the level-synchronous kernels with the
        // limit, from "no impact".  The same values as the whole call in level order, bit for bit, at a few per cent of its cost.
        c->np_pq_limit = false;
        const double toi_init = c->np_toi_init;
        c->np_scratch3_ovf.ensure(sizeof(int) * (size_t)n + 256);
        int* const d_list = c->np_scratch3_ovf.as<int>();
        unsigned* const d_count = reinterpret_cast<unsigned*>(c->np_scratch3_ovf.as<char>() + sizeof(int) * (size_t)n + 64);
        SCCD_HIP(hipMemsetAsync(d_count, 0, sizeof(unsigned), c->stream));
        hipLaunchKernelGGL(np_select_finite_k, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, c->stream, d_per_query_toi, n, d_list, d_count);
        SCCD_HIP(hipGetLastError());
        unsigned n_sel = 0;
        {
            ReadBack rb(c);
            rb.add(&n_sel, d_count, sizeof n_sel);
            rb.sync();
        }
        const unsigned long long checks_so_far = h.n_checks;
        std::memcpy(&h.toi_bits, &toi_init, 8); // (nothing accepted anywhere: the TOI the call started with)
        if (n_sel > 0) {
            hipLaunchKernelGGL(np_reset_selected_k, dim3((n_sel + 255) / 256), dim3(256), 0, c->stream, d_per_query_toi, d_list, n_sel);
            NarrowCounters h2;
            std::memset(&h2, 0, sizeof h2);
            std::memcpy(&h2.toi_bits, &toi_init, 8);
            SCCD_HIP(hipMemcpyAsync(d_cnt, &h2, sizeof h2, hipMemcpyHostToDevice, c->stream));
            SCCD_HIP(hipStreamSynchronize(c->stream)); // (h2 is on the stack)
            if (p.is_vf) run_level_sync<true>(c, p, d_cnt, n, d_per_query_toi, d_list, (long long)n_sel);
            else run_level_sync<false>(c, p, d_cnt, n, d_per_query_toi, d_list, (long long)n_sel);
            SCCD_HIP(hipMemcpyAsync(&h, d_cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
            SCCD_HIP(hipStreamSynchronize(c->stream));
            if (h.overflow) throw SccdError { SCCD_E_OVERFLOW, "narrow phase: work queue capacity exhausted" };
            h.n_checks += checks_so_far;
        }
    }
    std::memcpy(h_toi_inout, &h.toi_bits, 8);
    // n_checks is read by the caller through d_cnt mirror
    std::memcpy(c->h_scalars.as<char>() + 8192, &h, sizeof h);
}

void narrow_phase_run(sccd_ctx* c, const NarrowParams& p, NarrowCounters* d_cnt, double* h_toi_inout,
                      double* d_per_query_toi)
{
    narrow_phase_begin(c, p, d_cnt, h_toi_inout, d_per_query_toi);
    narrow_phase_end(c, p, d_cnt, h_toi_inout, d_per_query_toi);
}
