// drivers.hip -- the C ABI of include/sccd.h, part 3: narrow_phase (narrow_phase.cuh:30-46), ccd() with and without the
// collision list (ccd.cu:14-146), the two halves of a multi-GPU step, ipc_ccd_strategy (ipc_ccd_strategy.cu:12-152).
#include "api_internal.hpp"

// ------------------------------------------------------------------------------------------
// narrow phase
static NarrowCounters* narrow_counters(sccd_ctx* c)
{
    return reinterpret_cast<NarrowCounters*>(c->scalars.as<char>() + 2048);
}

struct NarrowResult {
    unsigned long long n_checks;
};

// the list(s) a pass's narrow phase runs on: what the projection cull kept (narrow_cull.inc, bp_detect_partial) -- one list, or one
// per half of time -- or every overlap
static const int2* pass_pairs(const sccd_broad_phase* bp) { return bp->cull.on ? bp->kept.as<int2>() : bp->overlaps.as<int2>(); }
static int64_t pass_count(const sccd_broad_phase* bp) { return bp->cull.on ? bp->n_kept : bp->n_overlaps; }
static void pass_lists(const sccd_broad_phase* bp, NarrowParams* p)
{
    p->pairs = pass_pairs(bp);
    p->n_pairs = pass_count(bp);
    p->second = NarrowParams::SecondHalf();
    // (the second half of the pass's time: its own list, made when it is asked for -- run_walk; a pass whose lists gave no sweep at
    // all has no overlaps, and no counters of its own on the device either)
    if (bp->cull.on && bp->cull.slabs.two && bp->sweeps_in_call > 0) {
        SweepCounters* const sw = bp->ctx->scalars.as<SweepCounters>();
        p->second.src = bp->overlaps.as<int2>();
        p->second.d_n_src = &sw->n_pairs;
        p->second.capacity = (long long)bp->capacity;
        p->second.kept = bp->kept_b.as<int2>();
        p->second.d_n_kept = &sw->n_kept_b;
        p->second.t_lo = bp->cull.slabs.t_mid;
        p->second.t_hi = bp->cull.slabs.t_end;
    }
}
// ... and whether a pass of ccd() culls at all: both scalar builds (round 6: the float build too -- narrow_cull.inc), with or without a
// check limit (SCCD_OPT_NARROW_ALGO = 1 keeps the reference's own list).  A CHECK LIMIT changes nothing about the cull's claim: a culled query has no domain
// that passes the inclusion test behind an acceptance (narrow_cull.inc), whatever the order of the traversal and wherever a limit
// cuts it off; the reference counts a query's checks per query (root_finder.cu:287-305) and prunes by a TOI only accepted domains
// lower, so a query that accepts nothing changes neither another query's count nor the running TOI -- the limited level-order
// result on the kept list IS the result on the whole list (the certificate and the level-order fallback of narrow.hip run on the
// kept list; only the number of checks differs).  One slab, [0, toi]: launches with a limit are single ones (narrow_start_toi).
// toi: the pass's narrow launches start from this TOI at most -- the slabs of time the cull looks at (narrow_cull_slabs)
static NarrowParams narrow_params(sccd_ctx* c, const sccd_mesh* m, const int2* d_pairs, int64_t n, int is_vf, int max_iter,
                                  double tol, double ms, int allow_zero_toi);
static void pass_cull_setup(sccd_ctx* c, sccd_broad_phase* bp, const sccd_mesh* m, bool vf, double ms, int max_iter, double tol, double toi)
{
    // (SCCD_OPT_CULL = 1: where it pays -- the cull is a launch per sweep, ~5 us of a small step's latency chain; measured on folded
    // cloths and the cloth-on-ball scenes: a gain from ~20,000 triangles on, a loss of 15 us at 10,000.  2: always)
    const bool big_enough = c->cull_on >= 2 || (long long)m->nE + m->nF >= SCCD_CULL_MIN_ELEMENTS;
    bp->cull.on = c->cull_on && big_enough && c->narrow_algo != 1 && std::isfinite(tol) && tol > 0 && ms >= 0;
    bp->cull.mesh = m;
    bp->cull.is_vf = vf ? 1 : 0;
    bp->cull.ms = ms;
    bp->cull.tol = tol;
    bp->cull.slabs = CullSlabs();
    if (bp->cull.on) bp->cull.slabs = narrow_cull_slabs(c, narrow_params(c, m, nullptr, 0, vf ? 1 : 0, max_iter, tol, ms, 1), toi);
}
// The lists of a pass were made for launches that start at or below cull.slabs.t_end; a start ABOVE it (a speculative bound that did
// not hold: the step is redone from 1 on the pair lists that are still on the device) culls the pass's overlaps again, on the
// pass's own stream; the host waits for the counts.
static void pass_recull(sccd_ctx* c, sccd_broad_phase* bp, double toi, int max_iter)
{
    if (!bp->cull.on || toi <= bp->cull.slabs.t_end) return;
    sccd_ctx* const bc = bp->ctx;
    SweepCounters* const d_cnt = bc->scalars.as<SweepCounters>();
    bp->cull.slabs = narrow_cull_slabs(c, narrow_params(c, bp->cull.mesh, nullptr, 0, bp->cull.is_vf, max_iter, bp->cull.tol, bp->cull.ms, 1), toi);
    SweepCounters h {};
    h.n_pairs = (unsigned long long)bp->n_overlaps;
    copy_in(bc, &d_cnt->n_pairs, &h.n_pairs, sizeof h.n_pairs, 0);
    copy_in(bc, &d_cnt->n_kept, &h.n_kept, sizeof h.n_kept, 0);
    {
        ProfScope ps(bc, SCCD_PROF_CULL);
        NarrowParams p {};
        p.V = bp->cull.mesh->V.as<double>();
        p.E = bp->cull.mesh->E.as<int2>();
        p.F = bp->cull.mesh->F.as<int4>();
        p.pairs = bp->overlaps.as<int2>();
        p.is_vf = bp->cull.is_vf;
        p.ms = bp->cull.ms;
        p.tol = bp->cull.tol;
        bp->kept.ensure(sizeof(int2) * (size_t)std::max<int64_t>(bp->capacity, 1));
        if (bp->cull.slabs.two) bp->kept_b.ensure(sizeof(int2) * (size_t)std::max<int64_t>(bp->capacity, 1));
        narrow_cull_launch(bc, p, &d_cnt->n_pairs, (long long)bp->n_overlaps, bp->kept.as<int2>(), &d_cnt->n_kept, 0.0,
                           bp->cull.slabs.two ? bp->cull.slabs.t_mid : bp->cull.slabs.t_end);
    }
    ReadBack rb(bc);
    rb.add(&h, d_cnt, sizeof h);
    rb.sync();
    bp->n_kept = (int64_t)h.n_kept;
}

static NarrowParams narrow_params(sccd_ctx* c, const sccd_mesh* m, const int2* d_pairs, int64_t n, int is_vf, int max_iter,
                                  double tol, double ms, int allow_zero_toi)
{
    if (c->scalar_f32) { // the float build takes Scalar (= float) arguments (narrow_phase.cuh:30-46)
        tol = (double)(float)tol;
        ms = (double)(float)ms;
    }
    // Condition 1 (root_finder.cu:322) can only end a bisection for a positive finite tolerance; the reference
    // asserts nothing and would bisect down to empty intervals (Condition 4) -- refused here instead
    SCCD_REQUIRE(tol > 0 && std::isfinite(tol), "narrow_phase: tolerance must be positive and finite");
    SCCD_REQUIRE(ms >= 0 && std::isfinite(ms), "narrow_phase: minimum separation must be >= 0 and finite");
    NarrowParams p;
    p.V = m->V.as<double>();
    p.E = m->E.as<int2>();
    p.F = m->F.as<int4>();
    p.pairs = d_pairs;
    p.n_pairs = n;
    p.is_vf = is_vf;
    p.max_iter = max_iter;
    p.tol = tol;
    p.ms = ms;
    p.allow_zero_toi = allow_zero_toi;
    p.arith = c->arith;
    return p;
}
static NarrowResult narrow_result(sccd_ctx* c)
{
    NarrowCounters h;
    std::memcpy(&h, c->h_scalars.as<char>() + 8192, sizeof h);
    return NarrowResult { h.n_checks };
}
static NarrowResult run_narrow(sccd_ctx* c, const sccd_mesh* m, const int2* d_pairs, int64_t n, int is_vf, int max_iter,
                               double tol, double ms, int allow_zero_toi, double* toi, double* d_per_query)
{
    const NarrowParams p = narrow_params(c, m, d_pairs, n, is_vf, max_iter, tol, ms, allow_zero_toi);
    if (c->scalar_f32) *toi = (double)(float)*toi;
    narrow_phase_run(c, p, narrow_counters(c), toi, d_per_query);
    return narrow_result(c);
}

// ... of a pass of ccd(): on the cull's list(s)
static NarrowResult run_narrow_pass(sccd_ctx* c, const sccd_mesh* m, sccd_broad_phase* bp, int is_vf, int max_iter, double tol, double ms,
                                    int allow_zero_toi, double* toi)
{
    pass_recull(c, bp, *toi, max_iter);
    NarrowParams p = narrow_params(c, m, nullptr, 0, is_vf, max_iter, tol, ms, allow_zero_toi);
    pass_lists(bp, &p);
    if (c->scalar_f32) *toi = (double)(float)*toi;
    narrow_phase_run(c, p, narrow_counters(c), toi, nullptr);
    return narrow_result(c);
}

// copy_out_collisions (narrow_phase.cu:84-103): the queries with toi < 1, appended as (aid, bid, toi).  The filter runs on
// the device (ballot + one atomic per wave); only the records that survive cross the bus.
__global__ void collisions_compact_k(const int2* __restrict__ pairs, const double* __restrict__ per_query, long long n,
                                     sccd_collision* __restrict__ out, long long* __restrict__ out_idx,
                                     unsigned long long* __restrict__ n_out)
{
    // ONE atomic per block of 1024 queries that holds a hit (the counter is one hot word: an atomic per wave -- 30,000 of them on
    // the 1M-triangle cloth's 64,521 collisions -- made this scan of 40 MB a 218 us kernel)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const double t = i < n ? per_query[i] : 2.0;
    const bool hit = t < 1;
    const unsigned long long mask = __ballot(hit);
    __shared__ unsigned long long s_cnt[16], s_base;
    const int w = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
    if (lane_id() == 0) s_cnt[w] = (unsigned long long)popc64(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long sum = 0;
        for (int k = 0; k < nw; k++) {
            const unsigned long long v = s_cnt[k];
            s_cnt[k] = sum;
            sum += v;
        }
        s_base = sum ? atomicAdd(n_out, sum) : 0ull;
    }
    __syncthreads();
    if (hit) {
        const int2 p = pairs[i];
        const unsigned long long at = s_base + s_cnt[w] + (unsigned long long)mbcnt64(mask);
        out[at] = sccd_collision { p.x, p.y, t };
        out_idx[at] = i;
    }
}
// ordered: the records in the order of the pair list (sccd_narrow_phase: the list is the caller's); else as the blocks reserved their slots
static void copy_out_collisions(sccd_ctx* c, const int2* d_pairs, const double* d_pq, int64_t n, std::vector<sccd_collision>& acc, bool ordered)
{
    if (n <= 0) return;
    DevBuf& out = c->col_out;
    DevBuf& idx = c->col_idx;
    out.ensure(sizeof(sccd_collision) * (size_t)n);
    idx.ensure(sizeof(long long) * ((size_t)n + 8)); // (+ the counter behind the query numbers)
    unsigned long long* const d_k = reinterpret_cast<unsigned long long*>(idx.as<long long>() + n);
    SCCD_HIP(hipMemsetAsync(d_k, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(collisions_compact_k, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, c->stream, d_pairs, d_pq,
                       (long long)n, out.as<sccd_collision>(), idx.as<long long>(), d_k);
    SCCD_HIP(hipGetLastError());
    unsigned long long k = 0;
    {
        ReadBack rb(c);
        rb.add(&k, d_k, sizeof k);
        rb.sync();
    }
    if (k == 0) return;
    if (!ordered) {
        // (ccd() with the collision list: the queries are what the projection cull kept, in the order ITS blocks appended them --
        // "query order" means nothing there, and the reference's own order comes from atomics and is unspecified: narrow_phase.cu:84-103.
        // No second copy, no sort of 64,000 indices on the host: 0.6 ms of the call on the 1M-triangle cloth)
        const size_t at0 = acc.size();
        acc.resize(at0 + (size_t)k);
        SCCD_HIP(hipMemcpy(acc.data() + at0, out.p, sizeof(sccd_collision) * (size_t)k, hipMemcpyDeviceToHost));
        return;
    }
    std::vector<sccd_collision> rec((size_t)k);
    std::vector<long long> at((size_t)k);
    SCCD_HIP(hipMemcpy(rec.data(), out.p, sizeof(sccd_collision) * (size_t)k, hipMemcpyDeviceToHost));
    SCCD_HIP(hipMemcpy(at.data(), idx.p, sizeof(long long) * (size_t)k, hipMemcpyDeviceToHost));
    // waves reserve their slots in order of arrival: put the records back into query order (what a serial
    // copy_out_collisions gives; the reference's own order comes from atomics and is unspecified)
    std::vector<size_t> order((size_t)k);
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return at[a] < at[b]; });
    acc.reserve(acc.size() + (size_t)k);
    for (size_t i = 0; i < order.size(); i++) acc.push_back(rec[order[i]]);
}
static sccd_collision* collisions_to_c(const std::vector<sccd_collision>& acc)
{
    sccd_collision* o = (sccd_collision*)std::malloc(std::max<size_t>(16, sizeof(sccd_collision) * acc.size()));
    if (!o) throw SccdError { SCCD_E_NOMEM, "host allocation failed" };
    if (!acc.empty()) std::memcpy(o, acc.data(), sizeof(sccd_collision) * acc.size());
    return o;
}

extern "C" int sccd_narrow_phase(sccd_ctx* c, const sccd_mesh* m, const int32_t* pairs, int64_t n, int pairs_on_device,
                                 int is_vf, int max_iter, double tol, double ms, int allow_zero_toi, double* toi,
                                 sccd_collision** collisions, int64_t* n_collisions)
{
    if (!c || !m || !toi) return SCCD_E_INVALID;
    if (collisions) *collisions = nullptr;
    if (n_collisions) *n_collisions = 0;
    return guarded(c, [&] {
        SCCD_REQUIRE(n >= 0 && (n == 0 || pairs), "narrow_phase: bad pair list");
        SCCD_REQUIRE(*toi >= 0, "narrow_phase: toi must be >= 0");
        const int2* d_pairs = reinterpret_cast<const int2*>(pairs);
        std::vector<int32_t> h_pairs;
        if (n > 0) {
            // validate indices on the host copy (the reference asserts nothing and would fault)
            if (!pairs_on_device) {
                const int na = is_vf ? m->nV : m->nE, nb = is_vf ? m->nF : m->nE;
                for (int64_t i = 0; i < n; i++)
                    SCCD_REQUIRE(pairs[2 * i] >= 0 && pairs[2 * i] < na && pairs[2 * i + 1] >= 0 && pairs[2 * i + 1] < nb,
                                 "narrow_phase: pair index out of range");
                c->np_scratch3.ensure(sizeof(int2) * (size_t)n);
                copy_in(c, c->np_scratch3.p, pairs, sizeof(int2) * (size_t)n, 0);
                d_pairs = c->np_scratch3.as<int2>();
            }
        }
        double* d_pq = nullptr;
        DevBuf& pq = c->col_pq;
        if (collisions && n > 0) {
            pq.ensure(sizeof(double) * (size_t)n);
            d_pq = pq.as<double>();
        }
        // THE PROJECTION CULL in front of a caller's list as well (round 6; narrow_cull.inc): a pair it drops has no domain the
        // bisection could accept in [0, toi] -- with or without a check limit (pass_cull_setup) -- so the kept list returns what the
        // whole list returns.  Not with the per-query records (their order is the caller's list's), not for SCCD_OPT_NARROW_ALGO = 1
        // (the reference's own traversal of the reference's own list), from a list length on (SCCD_OPT_CULL = 2: always).
        const bool cull = !collisions && c->cull_on && c->narrow_algo != 1 && n > 0 && *toi > 0 && std::isfinite(tol) && tol > 0 && ms >= 0
            && std::isfinite(ms) && (c->cull_on >= 2 || n >= SCCD_CULL_MIN_PAIRS);
        if (cull) {
            c->np_cull_list.ensure(sizeof(int2) * (size_t)n + 64);
            int2* const kept = c->np_cull_list.as<int2>();
            unsigned long long* const d_cnt = reinterpret_cast<unsigned long long*>(c->np_cull_list.as<char>() + ((sizeof(int2) * (size_t)n + 15) & ~(size_t)15));
            const unsigned long long h_cnt[2] = { (unsigned long long)n, 0ull }; // {pairs in, pairs kept}
            copy_in(c, d_cnt, h_cnt, sizeof h_cnt, 0);
            NarrowParams p {};
            p.V = m->V.as<double>();
            p.E = m->E.as<int2>();
            p.F = m->F.as<int4>();
            p.pairs = d_pairs;
            p.is_vf = is_vf;
            p.ms = ms;
            p.tol = tol;
            {
                ProfScope ps(c, SCCD_PROF_CULL);
                narrow_cull_launch(c, p, d_cnt, (long long)n, kept, d_cnt + 1, 0.0, std::min(*toi, 1.0));
            }
            unsigned long long k = 0;
            ReadBack rb(c);
            rb.add(&k, d_cnt + 1, sizeof k);
            rb.sync();
            SCCD_REQUIRE((int64_t)k <= n, "narrow_phase: the cull kept more than it was given");
            d_pairs = kept;
            n = (int64_t)k;
        }
        run_narrow(c, m, d_pairs, n, is_vf, max_iter, tol, ms, allow_zero_toi, toi, d_pq);
        if (collisions && n > 0) {
            std::vector<sccd_collision> acc;
            copy_out_collisions(c, d_pairs, d_pq, n, acc, /*ordered=*/true);
            *collisions = collisions_to_c(acc);
            if (n_collisions) *n_collisions = (int64_t)acc.size();
        }
    });
}

static void query_cull(sccd_ctx* c, const sccd_mesh* m, const int32_t* pairs, int64_t n, int is_vf, double ms, double tol, double t_lo, double t_hi,
                       int32_t* kept, int64_t* n_kept)
{
    SCCD_REQUIRE(n >= 0 && (n == 0 || (pairs && kept)), "query_cull: bad pair list");
    SCCD_REQUIRE(tol > 0 && std::isfinite(tol) && ms >= 0 && std::isfinite(ms), "query_cull: tolerance must be positive and finite, minimum separation >= 0");
    SCCD_REQUIRE(t_lo >= 0 && t_lo < t_hi && t_hi <= 1, "query_cull: the slab must satisfy 0 <= t_lo < t_hi <= 1");
    if (n == 0) return;
    const int na = is_vf ? m->nV : m->nE, nb = is_vf ? m->nF : m->nE;
    for (int64_t i = 0; i < n; i++)
        SCCD_REQUIRE(pairs[2 * i] >= 0 && pairs[2 * i] < na && pairs[2 * i + 1] >= 0 && pairs[2 * i + 1] < nb, "query_cull: pair index out of range");
    DevBuf d_in, d_out, d_cnt;
    d_in.ensure(sizeof(int2) * (size_t)n);
    d_out.ensure(sizeof(int2) * (size_t)n);
    d_cnt.ensure(16);
    const unsigned long long h_cnt[2] = { (unsigned long long)n, 0ull }; // {pairs in, pairs kept}
    copy_in(c, d_in.p, pairs, sizeof(int2) * (size_t)n, 0);
    copy_in(c, d_cnt.p, h_cnt, sizeof h_cnt, 0);
    NarrowParams p {};
    p.V = m->V.as<double>();
    p.E = m->E.as<int2>();
    p.F = m->F.as<int4>();
    p.pairs = d_in.as<int2>();
    p.is_vf = is_vf;
    p.ms = ms;
    p.tol = tol;
    narrow_cull_launch(c, p, d_cnt.as<unsigned long long>(), (long long)n, d_out.as<int2>(), d_cnt.as<unsigned long long>() + 1, t_lo, t_hi);
    unsigned long long k = 0;
    SCCD_HIP(hipMemcpyAsync(&k, d_cnt.as<unsigned long long>() + 1, sizeof k, hipMemcpyDeviceToHost, c->stream));
    SCCD_HIP(hipStreamSynchronize(c->stream));
    SCCD_REQUIRE((int64_t)k <= n, "query_cull: kept more than it was given");
    if (k) SCCD_HIP(hipMemcpy(kept, d_out.p, sizeof(int2) * (size_t)k, hipMemcpyDeviceToHost));
    *n_kept = (int64_t)k;
}

extern "C" int sccd_query_cull(sccd_ctx* c, const sccd_mesh* m, const int32_t* pairs, int64_t n, int is_vf, double ms, double tol,
                               int32_t* kept, int64_t* n_kept)
{
    if (!c || !m || !n_kept) return SCCD_E_INVALID;
    *n_kept = 0;
    return guarded(c, [&] { query_cull(c, m, pairs, n, is_vf, ms, tol, 0.0, 1.0, kept, n_kept); });
}

extern "C" int sccd_query_cull_slab(sccd_ctx* c, const sccd_mesh* m, const int32_t* pairs, int64_t n, int is_vf, double ms, double tol,
                                    double t_lo, double t_hi, int32_t* kept, int64_t* n_kept)
{
    if (!c || !m || !n_kept) return SCCD_E_INVALID;
    *n_kept = 0;
    return guarded(c, [&] { query_cull(c, m, pairs, n, is_vf, ms, tol, t_lo, t_hi, kept, n_kept); });
}

// ------------------------------------------------------------------------------------------
// drivers

// partial_ccd<run_vf> (ccd.cu:14-78): build, then alternate detect_overlaps_partial / narrow_phase
// (bp may belong to the helper context: its sweeps then run on that context's stream; every sweep ends with a host
// round trip, so the narrow phase on c->stream starts after the pairs are complete either way)
static void ccd_pass(sccd_ctx* c, const sccd_mesh* m, Pipeline* pl, sccd_broad_phase* bp, bool vf, double ms, int max_iter,
                     double tol, int allow_zero_toi, double* toi, sccd_stats* st, bool built = false, bool swept = false,
                     std::function<void()>* before_narrow = nullptr)
{
    if (!swept) pass_cull_setup(c, bp, m, vf, ms, max_iter, tol, *toi); // (a sweep that is enqueued already was set up by its caller)
    if (built) {} // (ccd() had the lists built already, by the helper)
    else if (vf) bp_build(bp, &pl->vb, &pl->fb);
    else bp_build(bp, &pl->eb, nullptr);
    bool started = swept; // (... and the first sweep enqueued as well: bp_detect_partial(bp, 1))
    while (bp->cursor < bp->total_rows) {
        {
            // ahead of the sweep: one copy less between sweep and narrow phase (from the TOI the launch will start from:
            // narrow_start_toi -- 0.5 where the walk kernel runs its two halves of time)
            const NarrowParams p0 = narrow_params(c, m, nullptr, 0, vf ? 1 : 0, max_iter, tol, ms, allow_zero_toi);
            narrow_counters_upload(c, narrow_counters(c), narrow_start_toi(c, p0, *toi, false));
        }
        if (!started && bp->cull.on) // (every chunk's cull looks at what is left of the step)
            bp->cull.slabs = narrow_cull_slabs(c, narrow_params(c, m, nullptr, 0, vf ? 1 : 0, max_iter, tol, ms, allow_zero_toi), *toi);
        bp_detect_partial(bp, started ? 2 : 0);
        started = false;
        if (before_narrow && *before_narrow) {
            (*before_narrow)();
            *before_narrow = nullptr; // once
        }
        const NarrowResult r = run_narrow_pass(c, m, bp, vf ? 1 : 0, max_iter, tol, ms, allow_zero_toi, toi);
        if (st) {
            (vf ? st->n_vf_pairs : st->n_ee_pairs) += bp->n_overlaps;
            (vf ? st->n_vf_culled : st->n_ee_culled) += bp->n_overlaps - pass_count(bp);
            (vf ? st->n_vf_checks : st->n_ee_checks) += (int64_t)r.n_checks;
        }
    }
    if (st) (vf ? st->n_vf_candidates : st->n_ee_candidates) = bp->candidates;
}

static void ccd_on_mesh_from(sccd_ctx* c, const sccd_mesh* m, double ms, int max_iter, double tol, int allow_zero_toi, double toi0,
                             double* toi_out, sccd_stats* st, bool* lists_resident);

// The device's own clock around one ccd() call on a mesh (sccd_ctx::device_span_ns): armed here, stamped by the call's first kernel and
// by the kernels behind its read-backs, read when the call is over -- whatever way it went (redone from 1, rebuilt, chunked).
struct StepSpan {
    sccd_ctx* c;
    explicit StepSpan(sccd_ctx* ctx) : c(ctx)
    {
        c->step_stamp_armed = true;
        c->step_t_last = 0;
        if (c->side) c->side->step_t_last = 0;
        c->device_span_ns = -1;
    }
    ~StepSpan()
    {
        const bool stamped = !c->step_stamp_armed; // (a call that launched no vertex boxes -- an empty mesh -- has no span)
        c->step_stamp_armed = false;
        if (!stamped || !c->mailbox.p) return;
        unsigned long long t0;
        std::memcpy(&t0, c->mailbox.as<char>() + SCCD_MAILBOX_BYTES + 64, sizeof t0);
        const unsigned long long t1 = std::max(c->step_t_last, c->side ? c->side->step_t_last : 0ull);
        if (t1 > t0) c->device_span_ns = (long long)((double)(t1 - t0) * 1e6 / (double)std::max(1, c->wall_clock_khz));
    }
};

// ccd() of ccd.cu:80-146 on a resident mesh.  THE SPECULATIVE BOUND (round 4): the reference starts every call from toi = 1
// (ccd.cu:125) and the vertex-face pass then runs without a bound until its first contact query has been bisected to the end --
// 55 dependent steps, three quarters of the launch, in which every query also explores the later halves of its time splits:
// 5.9 M of the 13.8 M vertex-face checks of the 1M-triangle cloth.  A caller that steps a simulation calls ccd() on a mesh that
// hardly moved: if the previous call on this mesh found an impact at T, this one starts from min(1, 1.125 T).  narrow_phase's toi is
// in/out (narrow_phase.cu:126): the call returns min(bound, earliest accepted domain below it), and pruning by a bound never
// removes a domain earlier than the bound -- so a result BELOW the bound is exactly what a start from 1 returns.  A result AT the
// bound proves nothing (the earliest impact may lie beyond it): the step is redone from 1.  Counted (SCCD_OPT_TOI_GUESS_HITS /
// _MISSES), off with SCCD_OPT_TOI_GUESS = 0 or SCCD_SPECULATE=0; calls with a check limit start from 1 as before (their
// certificate is about the TOI the call started with).
static void ccd_on_mesh(sccd_ctx* c, const sccd_mesh* m, double ms, int max_iter, double tol, int allow_zero_toi,
                        double* toi_out, sccd_stats* st)
{
    const bool same_mesh = c->toi_guess_mesh == (const void*)m && c->toi_guess_n[0] == m->nV && c->toi_guess_n[1] == m->nE
        && c->toi_guess_n[2] == m->nF;
    bool spec = c->toi_guess_on && lab_env().speculate && same_mesh && max_iter < 0 && c->toi_guess < 1.0 && c->toi_guess > 0.0;
    if (spec && c->toi_guess_rest > 0) { // (a bound broke lately: a few steps from 1 before the next try)
        c->toi_guess_rest -= 1;
        spec = false;
    }
    // THE TWO HALVES OF TIME ARE A BET (narrow_walk.inc): they pay when the earliest impact lies before 0.5 -- 0.85 instead of 0.95 ms
    // on the 1M-triangle cloth -- and cost a second cull and a second launch per pass when it lies later or there is none (+ 20 %
    // on a 500k-triangle cloth whose impact comes at 0.59).  What the LAST call on this mesh returned settles the bet for this one
    // (same history, same switch as the bound: SCCD_OPT_TOI_GUESS); exact either way -- only the number of launches changes.
    struct HalvesOff {
        sccd_ctx* c;
        ~HalvesOff() { c->two_halves_off = 0; }
    } halves_off { c };
    c->two_halves_off = (c->two_halves == 1 && c->toi_guess_on && lab_env().speculate && same_mesh && c->toi_last >= 0.5) ? 1 : 0;
    // (... and the size of the mesh: the second launch, its cull and the kernel between them are ~25 us of launches in a row -- more
    // than a small step's narrow phase has to give: 0.296 -> 0.323 ms on the 10k-triangle cloth-on-ball scene, 0.97 -> 0.86 on the
    // 1M-triangle cloth.  SCCD_OPT_TWO_HALVES = 2: always)
    if (c->two_halves == 1 && (long long)m->nE + m->nF < SCCD_TWO_HALVES_MIN_ELEMENTS) c->two_halves_off = 1;
    double toi = 1.0;
    bool resident = false;
    // (the float build's kernels and run_narrow() round the TOI they start from to float: the bound must BE the value they start
    // from, or a result "below the bound" could be the rounded bound itself -- ADVICE r04)
    const double bound = c->scalar_f32 ? (double)(float)c->toi_guess : c->toi_guess;
    spec = spec && bound < 1.0 && bound > 0.0;
    if (spec) {
        ccd_on_mesh_from(c, m, ms, max_iter, tol, allow_zero_toi, bound, &toi, st, &resident);
        if (toi < bound) {
            c->toi_guess_hits += 1;
            c->toi_guess_backoff = std::max(4, c->toi_guess_backoff / 2);
        } else { // nothing was accepted below the bound: the earliest impact lies at or beyond it -- from 1, as the reference does
            c->toi_guess_misses += 1;
            c->toi_guess_rest = c->toi_guess_backoff;
            c->toi_guess_backoff = std::min(64, c->toi_guess_backoff * 2);
            if (resident) {
                // both pair lists are still on the device, each swept in one chunk: only the narrow phases again (ccd.cu:125-143)
                Pipeline* const pl = pipeline_of(c);
                toi = 1.0;
                // (the culls looked at the step up to the bound: run_narrow_pass makes the lists again for what the pass now starts from)
                const NarrowResult rv = run_narrow_pass(c, m, &pl->bp, 1, max_iter, tol, ms, allow_zero_toi, &toi);
                const NarrowResult re = run_narrow_pass(c, m, &pl->bp_ee, 0, max_iter, tol, ms, allow_zero_toi, &toi);
                if (st) {
                    st->n_vf_checks += (int64_t)rv.n_checks;
                    st->n_ee_checks += (int64_t)re.n_checks;
                }
            } else {
                ccd_on_mesh_from(c, m, ms, max_iter, tol, allow_zero_toi, 1.0, &toi, st, nullptr);
            }
        }
    } else {
        ccd_on_mesh_from(c, m, ms, max_iter, tol, allow_zero_toi, 1.0, &toi, st, nullptr);
    }
    c->toi_guess = (toi < 1.0 && toi > 0.0) ? std::min(1.0, toi * 1.125) : 1.0; // (an eighth above: the pruning a bound buys comes in dyadic steps -- 1.25 x 0.408 = 0.51 keeps the later half of every first time split alive, 0.459 does not)
    c->toi_last = toi;
    c->toi_guess_mesh = (const void*)m;
    c->toi_guess_n[0] = m->nV;
    c->toi_guess_n[1] = m->nE;
    c->toi_guess_n[2] = m->nF;
    *toi_out = toi;
}

// One pass's share of a step's stats
static void pass_stats(sccd_stats* st, bool vf, const sccd_broad_phase* bp, const NarrowResult& r)
{
    if (!st) return;
    (vf ? st->n_vf_pairs : st->n_ee_pairs) += bp->n_overlaps;
    (vf ? st->n_vf_culled : st->n_ee_culled) += bp->n_overlaps - pass_count(bp);
    (vf ? st->n_vf_checks : st->n_ee_checks) += (int64_t)r.n_checks;
    (vf ? st->n_vf_candidates : st->n_ee_candidates) = bp->candidates;
}

// THE STEP (round 6).  ccd() of ccd.cu:80-146 with NO HOST IN IT: one thread enqueues both passes' whole chains -- boxes, the two
// speculative builds (build.hip), sweeps, culls, walk kernels -- on two streams ordered by events, and only then waits, for the two
// passes' VERDICTS: np_verdict_k, the kernel behind each pass's (first) walk launch, leaves the launch's counters in host-coherent
// pinned memory and, behind them, everything else the host used to read back one round trip at a time -- the sweep's counters (pair
// count, overflow), the grid and entry counts the speculative build really had, a rank's cell window.  Every launch takes its sizes
// from device memory (the sweep and records kernels since round 3, the edge-edge walk kernel since round 5, the vertex-face one since
// round 6: np_walk_k walk_deal).  The host looks at the verdicts when both are in: a guess that held, a pair buffer that was large
// enough, no query beyond the walk kernel's reach -- the common case -- and the step is over with zero read-backs; anything else and
// the pass concerned is done again the slow way, from what is known (whatever a walk kernel put into a running TOI on the way is an
// accepted domain of a true pair: a speculative build that fails emits only pairs of boxes that do overlap).
// Rounds 3-5 grew this from the other end -- a worker thread for the edge-edge build, a flag between the threads for the records gate,
// four read-backs per step, the vertex-face walk launched by the host once it knew the count -- and the host's share showed in the
// driver's clock (20-step means 10-25 % above the median on a shared host).  One thread is enough: enqueueing is
// ~150 us of host time at the front of a step the device needs 800 us for.
// Paths with a host inside (check limits, chunked sweeps, the float build, level order): the same two chains, the passes in sequence.
static void ccd_on_mesh_from(sccd_ctx* c, const sccd_mesh* m, double ms, int max_iter, double tol, int allow_zero_toi, double toi0,
                             double* toi_out, sccd_stats* st, bool* lists_resident)
{
    if (lists_resident) *lists_resident = false;
    Pipeline* pl = pipeline_of(c);
    if (st) std::memset(st, 0, sizeof *st);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double before[SCCD_PROF_COUNT];
    if (st && c->profile == 1) {
        SCCD_HIP(hipEventCreate(&e0));
        SCCD_HIP(hipEventCreate(&e1));
        SCCD_HIP(hipEventRecord(e0, c->stream));
        merge_side_profile(c);
        std::memcpy(before, c->prof_ms, sizeof before);
    }
    auto finish = [&](double toi) {
        *toi_out = toi;
        if (!(st && c->profile == 1)) return;
        SCCD_HIP(hipEventRecord(e1, c->stream));
        SCCD_HIP(hipEventSynchronize(e1));
        float msf = 0;
        SCCD_HIP(hipEventElapsedTime(&msf, e0, e1));
        st->ms_total = msf;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        merge_side_profile(c); // (the edge-edge half of the step ran on the helper context)
        st->ms_boxes = c->prof_ms[SCCD_PROF_BOXES] - before[SCCD_PROF_BOXES];
        st->ms_sort = c->prof_ms[SCCD_PROF_SORT] - before[SCCD_PROF_SORT];
        st->ms_sweep = (c->prof_ms[SCCD_PROF_SWEEP] - before[SCCD_PROF_SWEEP]) + (c->prof_ms[SCCD_PROF_SWEEP_EE] - before[SCCD_PROF_SWEEP_EE]);
        st->ms_narrow = (c->prof_ms[SCCD_PROF_NARROW_VF] - before[SCCD_PROF_NARROW_VF])
            + (c->prof_ms[SCCD_PROF_NARROW_EE] - before[SCCD_PROF_NARROW_EE]) + (c->prof_ms[SCCD_PROF_CULL] - before[SCCD_PROF_CULL]);
    };
    // (a rank of a multi-GPU job builds the edge and face boxes of its window of cells only: see boxes_from_mesh)
    const bool lazy_ef = c->shard_count > 1 && c->sort_axis >= 0 && c->max_overlap_cutoff == 0 && !c->build_scan;
    // TWO STREAMS: the edge-edge chain on a helper context (own stream, scratch, counters and pinned mailboxes) beside the
    // vertex-face chain.  Both build chains are short, latency-bound kernels, so two of them interleave almost for free; the edge-edge
    // sweep runs beside the vertex-face narrow phase (dependent gathers against vector issue), the edge-edge walk kernel beside the tail
    // of the vertex-face one.  SCCD_OPT_PASSES_APART: one stream, the passes one after the other (measurements).
    const bool two_streams = !c->passes_apart && m->nE > 0;
    if (two_streams && !c->side) {
        if (sccd_create(c->device, &c->side) != SCCD_OK) throw SccdError { SCCD_E_NOMEM, "ccd: cannot create the helper context" };
        SCCD_HIP(hipEventCreateWithFlags(&c->side_event, hipEventDisableTiming));
        SCCD_HIP(hipEventCreateWithFlags(&c->side_event2, hipEventDisableTiming));
        SCCD_HIP(hipEventCreateWithFlags(&c->side_event3, hipEventDisableTiming));
        SCCD_HIP(hipEventCreateWithFlags(&c->side_event4, hipEventDisableTiming));
        pl->bp_ee.ctx = c->side;
    }
    // (two streams: the edge boxes are the helper stream's first kernel, beside the face boxes on this one -- boxes_from_mesh)
    const bool split_boxes = two_streams && !lazy_ef;
    boxes_from_mesh(c, m, ms, pl, true, true, true, lazy_ef, split_boxes ? c->side_event : nullptr); // inflation radius = min_distance (ccd.cu:112)
    double toi = toi0; // ccd.cu:125 starts from 1; ccd_on_mesh may hand a bound over
    if (!two_streams) {
        ccd_pass(c, m, pl, &pl->bp, true, ms, max_iter, tol, allow_zero_toi, &toi, st);
        ccd_pass(c, m, pl, &pl->bp, false, ms, max_iter, tol, allow_zero_toi, &toi, st);
        finish(toi);
        return;
    }
    sccd_ctx* const sc = c->side;
    sc->sort_axis = c->sort_axis;
    sc->sweep_algo = c->sweep_algo;
    sc->cell_factor_milli = c->cell_factor_milli;
    sc->build_scan = c->build_scan;
    sc->shard_rank = c->shard_rank;
    sc->shard_count = c->shard_count;
    sc->overlap_capacity = c->overlap_capacity;
    sc->max_overlap_cutoff = c->max_overlap_cutoff;
    sc->memory_limit_mb = c->memory_limit_mb;
    sc->profile = c->profile;
    sc->arith = c->arith;
    sc->scalar_f32 = c->scalar_f32;
    sc->narrow_algo = c->narrow_algo;
    sc->limit_level_order = c->limit_level_order;
    sc->two_halves = c->two_halves;
    sc->two_halves_off = c->two_halves_off;
    sc->cull_on = c->cull_on;
    sc->sweep_blocks_per_cu = 0;
    const NarrowParams pv0 = narrow_params(c, m, nullptr, 0, 1, max_iter, tol, ms, allow_zero_toi);
    const NarrowParams pe0 = narrow_params(sc, m, nullptr, 0, 0, max_iter, tol, ms, allow_zero_toi);
    // (chunked sweeps, level order -- SCCD_OPT_NARROW_ALGO = 1, check limits below 4,096, under SCCD_OPT_LIMIT_LEVEL_ORDER or in the
    // float build -- and diagnostics keep the passes in sequence, with the host between them)
    // A CHECK LIMIT the fast kernel serves (narrow.hip, the certificate: the kernel runs WITHOUT the limit, the host proves afterwards
    // that the limit could not have changed the answer) goes the same way since round 6 -- each pass on a running TOI of ITS OWN (the
    // proof is about the pass's own earliest accept: no shared word, no peer), both from the TOI the call started with; the
    // edge-edge pass's proof then starts from the vertex-face pass's result, as the reference's second pass does (end_pass below).
    const bool enqueue_all = c->max_overlap_cutoff == 0 && narrow_uses_walk_kernel(c, pv0, false)
        && narrow_uses_walk_kernel(sc, pe0, false) && lab_env().np_diag == 0 && c->verdict_dev && sc->verdict_dev;
    // WHICH PASS GOES FIRST.  The second sweep waits for the first (two sweeps at once gain nothing: both are bound by instruction issue),
    // so the step's critical path is one pass's whole chain plus the other pass's tail -- and the edge-edge tail is the long one (5.1 M
    // pairs against 1.5 M on the 1M-triangle cloth).  Leading with the EDGE-EDGE chain was measured in round 6 (this code runs either
    // way: A leads, B follows): 0.797-0.807 against 0.797-0.812 ms per step, three interleaved rounds, and 2 % WORSE where the earliest
    // impact comes late (profiles/r06/ab_which_pass_leads.log) -- the step is work-conserving, the chip is busy either way.  The
    // vertex-face chain leads, as in the reference (ccd.cu:125-143; the order never decides the minimum: Appendix A.20).
    const bool ee_first = false;
    struct PassRun {
        sccd_ctx* ctx;
        sccd_broad_phase* bp;
        bool vf;
        NarrowParams p;
        double toi;
        bool launched = false, stands = false;
    };
    PassRun run_vf { c, &pl->bp, true, pv0, toi }, run_ee { sc, &pl->bp_ee, false, pe0, toi };
    PassRun& A = ee_first ? run_ee : run_vf; // the chain that leads
    PassRun& B = ee_first ? run_vf : run_ee;
    // THE RECORDS GATE (build.hip records_gate_*): the second chain's records kernel is ordered, on the device, behind the END of the
    // leading chain's -- two bandwidth-bound kernels side by side take twice their time each, and only the leading one is on the way to
    // the first sweep; the other then runs beside that sweep (instruction issue).  From a mesh size on (a rank of a multi-GPU job: its
    // share of the mesh counts): below it the cross-queue wait costs what the gate gains.
    c->records_gate_signal = c->records_gate_wait = sc->records_gate_signal = sc->records_gate_wait = nullptr;
    if (((long long)m->nE + m->nF) / std::max(1, c->shard_count) >= SCCD_RECORDS_GATE_MIN_ELEMENTS) {
        if (!c->records_gate.ev) SCCD_HIP(hipEventCreateWithFlags(&c->records_gate.ev, hipEventDisableTiming));
        c->records_gate.recorded = false;
        A.ctx->records_gate_signal = B.ctx->records_gate_wait = &c->records_gate;
    }
    struct GateOff {
        sccd_ctx *c, *sc;
        ~GateOff() { c->records_gate_signal = c->records_gate_wait = sc->records_gate_signal = sc->records_gate_wait = nullptr; }
    } gate_off { c, sc };
    if (!split_boxes) SCCD_HIP(hipEventRecord(c->side_event, c->stream)); // the boxes are complete behind this point
    // ---- both build chains, this thread, the leading chain first: a chain's grid kernel also starts the counters of its pass's sweep
    // and narrow launch -- and, where the two walk kernels will keep a word each (a pass in two halves of time), leaves the other
    // pass's word there as the PEER to publish accepted times to (narrow_walk.inc; launches that share one word need none)
    const bool peers = enqueue_all && max_iter < 0 && !(narrow_start_toi(c, pv0, toi, false) == toi && narrow_start_toi(sc, pe0, toi, false) == toi);
    auto build_vf = [&] {
        c->np_init_pending = true;
        c->np_init_toi = narrow_start_toi(c, pv0, toi, false);
        c->np_init_peer = peers ? &narrow_counters(sc)->toi_bits : nullptr;
        pass_cull_setup(c, &pl->bp, m, true, ms, max_iter, tol, toi);
        bp_build(&pl->bp, &pl->vb, &pl->fb);
        c->np_init_pending = false;
        c->np_init_peer = nullptr;
    };
    // (the edge boxes enqueued AHEAD of the vertex-face build -- the helper's chain 40 us earlier on the device -- were measured in
    // round 6: 0.794-0.798 against 0.789-0.803 ms, nothing: profiles/r06/ab_edge_boxes_before_the_vertex_face_build.log)
    auto build_ee = [&] {
        SCCD_HIP(hipStreamWaitEvent(sc->stream, c->side_event, 0));
        if (split_boxes) edge_boxes_on(sc, m, pl);
        sc->np_init_pending = true;
        sc->np_init_toi = narrow_start_toi(sc, pe0, toi, false);
        sc->np_init_peer = peers ? &narrow_counters(c)->toi_bits : nullptr;
        pass_cull_setup(sc, &pl->bp_ee, m, false, ms, max_iter, tol, toi);
        bp_build(&pl->bp_ee, &pl->eb, nullptr);
        sc->np_init_pending = false;
        sc->np_init_peer = nullptr;
    };
    if (ee_first) {
        build_ee();
        build_vf();
    } else {
        build_vf();
        build_ee();
    }
    // The second pass's SWEEP goes into its stream behind the END OF THE LEADING SWEEP (an event between that sweep and its cull:
    // bp->after_sweep), and the LEADING WALK KERNEL behind the END OF THE SECOND SWEEP (the same way): it then runs beside the second
    // pass's cull, which lives on gather latency, instead of beside its sweep, which is bound by instruction issue like the walk kernel
    // itself.  (Rounds 4-5 started the walk kernel as soon as the second sweep's blocks were resident -- side by side, the two took
    // little longer than the longer one alone, it was thought; measured in round 6, three interleaved rounds of 100 steps: 0.786 /
    // 0.790 / 0.795 ms per step against 0.805 / 0.801 / 0.811, and 0.822 against 0.841 where the earliest impact comes late;
    // behind the second pass's CULL as well -- both walk kernels at once, with full or capped grids: no better than before --
    // profiles/r06/ab_leading_walk_behind_the_other_sweep.log, ab_both_walk_kernels_at_once.log.)
    bool b_swept = false, b_sweep_event = false;
    auto start_b_sweep = [&] {
        SCCD_HIP(hipEventRecord(c->side_event3, A.ctx->stream));
        SCCD_HIP(hipStreamWaitEvent(B.ctx->stream, c->side_event3, 0));
        SCCD_HIP(hipEventRecord(c->side_event4, B.ctx->stream)); // (the second stream has passed its wait: its sweep is next)
        if (enqueue_all)
            B.bp->after_sweep = [&] {
                SCCD_HIP(hipEventRecord(c->side_event2, B.ctx->stream)); // (the second sweep is complete behind this point)
                b_sweep_event = true;
            };
        try {
            bp_detect_partial(B.bp, 1);
        } catch (...) {
            B.bp->after_sweep = nullptr;
            throw;
        }
        B.bp->after_sweep = nullptr;
        b_swept = true;
    };
    if (!enqueue_all) { // (A is the vertex-face pass on this context, B the edge-edge pass on the helper's)
        std::function<void()> hook = start_b_sweep;
        ccd_pass(c, m, pl, &pl->bp, true, ms, max_iter, tol, allow_zero_toi, &toi, st, /*built=*/true, /*swept=*/false, &hook);
        ccd_pass(c, m, pl, &pl->bp_ee, false, ms, max_iter, tol, allow_zero_toi, &toi, st, /*built=*/true, /*swept=*/b_swept);
        finish(toi);
        return;
    }
    // ---- the sweeps and culls
    A.bp->after_sweep = start_b_sweep;
    try {
        bp_detect_partial(A.bp, 1);
    } catch (...) {
        A.bp->after_sweep = nullptr;
        throw;
    }
    A.bp->after_sweep = nullptr;
    if (!b_swept) start_b_sweep(); // (nothing to sweep in the leading lists)
    // ---- the walk kernels, each right behind its pass's cull, reading the list's length on the device; each followed by its verdict
    auto extras_of = [](const sccd_broad_phase* bp) {
        VerdictExtras x {};
        sccd_ctx* const bc = bp->ctx;
        x.src[0] = bc->scalars.as<unsigned>();
        x.off_words[0] = VERDICT_SWEEP_AT / 4;
        x.n_words[0] = (unsigned)(sizeof(SweepCounters) / 4);
        x.n = 1;
        if (bp->speculative) {
            x.src[1] = reinterpret_cast<const unsigned*>(bp->grid.as<char>() + 512);
            x.off_words[1] = VERDICT_BUILT_AT / 4;
            x.n_words[1] = (unsigned)(sizeof(GridReadBack) / 4);
            x.n = 2;
            if (bp->spec_window) {
                x.src[2] = reinterpret_cast<const unsigned*>(bp->grid.as<char>() + 1024);
                x.off_words[2] = VERDICT_WINDOW_AT / 4;
                x.n_words[2] = (unsigned)(sizeof(ShardWindow) / 4);
                x.n = 3;
            }
        }
        return x;
    };
    static_assert(VERDICT_SWEEP_AT + sizeof(SweepCounters) <= VERDICT_BUILT_AT && VERDICT_BUILT_AT + sizeof(GridReadBack) <= VERDICT_WINDOW_AT
                      && VERDICT_WINDOW_AT + sizeof(ShardWindow) <= 4096 && sizeof(NarrowCounters) <= 2048,
                  "the verdict buffer's layout");
    A.launched = A.bp->sweeps_in_call == 1; // (else: no rows, no pairs, nothing to walk)
    B.launched = B.bp->sweeps_in_call == 1;
    // (the two kernels share ONE running TOI -- the leading pass's word: each prunes with what the other finds, the final minimum does
    // not depend on the order, Appendix A.20 -- unless a pass runs its two halves of time: its word then holds the bound 0.5 for a
    // while, which the other pass must not prune by; each keeps its own word then and publishes what it ACCEPTS to its peer's)
    const bool share_word = A.launched && B.launched && !peers && max_iter < 0;
    struct PeerGuard { // (fallback paths of the leading context wait for the other stream's launch before they reset a word it shares)
        sccd_ctx* c;
        ~PeerGuard() { c->np_peer_stream = nullptr; }
    } peer_guard { A.ctx };
    auto begin_walk = [&](PassRun& R) {
        pass_lists(R.bp, &R.p); // (the buffers; the counts are on the device)
        if (&R == &B && share_word) R.p.toi_word = &narrow_counters(A.ctx)->toi_bits;
        const SweepCounters* const sw = R.ctx->scalars.as<SweepCounters>();
        const VerdictExtras x = extras_of(R.bp);
        narrow_phase_begin(R.ctx, R.p, narrow_counters(R.ctx), &R.toi, nullptr, R.bp->cull.on ? &sw->n_kept : &sw->n_pairs, (long long)R.bp->capacity, &x);
    };
    if (A.launched) {
        // (beside the second pass's cull, not its sweep: above.  With a check limit the second walk kernel waits for the END of this
        // one -- its word is seeded with this one's result -- so this one goes as early as it can: beside the second sweep, once that
        // sweep's blocks are resident, as in rounds 4-5: 0.93 against 1.00 ms per step at max_iter = 1e7)
        if (b_sweep_event && max_iter < 0) SCCD_HIP(hipStreamWaitEvent(A.ctx->stream, c->side_event2, 0));
        else if (b_swept) SCCD_HIP(hipStreamWaitEvent(A.ctx->stream, c->side_event4, 0));
        begin_walk(A);
    }
    if (B.launched) {
        A.ctx->np_peer_stream = B.ctx->stream;
        if (max_iter >= 0 && A.launched) {
            // a check limit: the second pass starts from the first one's RESULT, as the reference's does -- its word is seeded on the
            // device, behind the first pass's walk kernel (which ends before the second pass's cull does: nothing waits long)
            SCCD_HIP(hipEventRecord(c->side_event, A.ctx->stream));
            SCCD_HIP(hipStreamWaitEvent(B.ctx->stream, c->side_event, 0));
            narrow_seed_word(B.ctx, narrow_counters(B.ctx), narrow_counters(A.ctx));
        }
        begin_walk(B);
    }
    // ---- everything is enqueued.  The verdicts: the first attempt's counters come with them (bp_detect_partial settles the
    // speculative build and the pair buffer on those; only a pass whose first attempt did NOT stand waits for anything else)
    auto settle = [&](PassRun& R) { // -> R.stands: the walk launch that is in the stream stands
        SweepFirstRead first;
        bool have_first = false;
        if (R.launched && R.ctx->verdict_armed) {
            if (const char* const from = narrow_verdict_wait(R.ctx)) {
                std::memcpy(&first.h, from + VERDICT_SWEEP_AT, sizeof first.h);
                std::memcpy(&first.built, from + VERDICT_BUILT_AT, sizeof first.built);
                std::memcpy(&first.hwin, from + VERDICT_WINDOW_AT, sizeof first.hwin);
                have_first = true;
            }
        }
        R.bp->pre_read = have_first ? &first : nullptr;
        try {
            bp_detect_partial(R.bp, 2);
        } catch (...) {
            R.bp->pre_read = nullptr;
            throw;
        }
        R.bp->pre_read = nullptr;
        R.stands = R.launched && R.bp->sweeps_in_call == 1 && R.bp->cursor >= R.bp->total_rows;
    };
    auto redo_pass = [&](PassRun& R) {
        // the walk kernel that went into the stream ahead of the verdict ran on a list that has been made again since: let it drain,
        // forget its verdict, and do the pass's narrow phase(s) the host's way, chunk by chunk if need be
        SCCD_HIP(hipStreamSynchronize(R.ctx->stream));
        R.ctx->verdict_armed = false;
        NarrowResult r = run_narrow_pass(R.ctx, m, R.bp, R.vf ? 1 : 0, max_iter, tol, ms, allow_zero_toi, &R.toi);
        pass_stats(st, R.vf, R.bp, r);
        while (R.bp->cursor < R.bp->total_rows) {
            if (R.bp->cull.on) R.bp->cull.slabs = narrow_cull_slabs(R.ctx, narrow_params(R.ctx, m, nullptr, 0, R.vf ? 1 : 0, max_iter, tol, ms, allow_zero_toi), R.toi);
            bp_detect_partial(R.bp, 0);
            r = run_narrow_pass(R.ctx, m, R.bp, R.vf ? 1 : 0, max_iter, tol, ms, allow_zero_toi, &R.toi);
            pass_stats(st, R.vf, R.bp, r);
        }
    };
    auto end_pass = [&](PassRun& R, PassRun& other, bool other_still_running) {
        if (R.stands) {
            pass_lists(R.bp, &R.p); // (now with the counts)
            if (&R == &B && share_word) R.p.toi_word = &narrow_counters(A.ctx)->toi_bits;
            // (a check limit: the second pass of the reference starts from the first one's RESULT, ccd.cu:125-143 -- its certificate,
            // and its level-order fallback, start from there; the launch itself started from the call's TOI, a looser bound, under
            // which a query's level-order count can only be larger: what it certifies holds from the tighter start as well)
            if (&R == &B && max_iter >= 0 && R.ctx->np_limit_fast) R.ctx->np_toi_init = std::min(R.ctx->np_toi_init, A.toi);
            narrow_phase_end(R.ctx, R.p, narrow_counters(R.ctx), &R.toi, nullptr);
            if (&R == &B && max_iter >= 0) R.toi = std::min(R.toi, A.toi);
            pass_stats(st, R.vf, R.bp, narrow_result(R.ctx));
        } else if (R.launched || R.bp->n_overlaps > 0 || R.bp->cursor < R.bp->total_rows) {
            if (other_still_running && other.launched) SCCD_HIP(hipStreamSynchronize(other.ctx->stream)); // (it may share this pass's word, or publish to it)
            else R.toi = std::min(R.toi, other.toi); // (the other pass's result seeds the pass that is done again: ccd.cu:125-143)
            redo_pass(R);
        } else if (st) {
            (R.vf ? st->n_vf_candidates : st->n_ee_candidates) = R.bp->candidates;
        }
    };
    // (the leading pass is settled and ended while the other's walk kernel still runs: whatever its end costs the host -- the
    // certificate of a check limit is ~100 us of bisection on the host -- lies in that kernel's shade)
    settle(A);
    end_pass(A, B, /*other_still_running=*/true);
    A.ctx->np_peer_stream = nullptr;
    settle(B);
    end_pass(B, A, /*other_still_running=*/false);
    toi = std::min(A.toi, B.toi);
    // (both passes are behind us; if each pair list was swept once, in one chunk, it is still on the device)
    if (lists_resident) *lists_resident = A.stands && B.stands;
    finish(toi);
}

extern "C" int sccd_ccd_mesh(sccd_ctx* c, const sccd_mesh* m, double ms, int max_iter, double tol, int allow_zero_toi,
                             double* toi, sccd_stats* stats)
{
    if (!c || !m || !toi) return SCCD_E_INVALID;
    return guarded(c, [&] {
        SCCD_REQUIRE(m->ctx == c, "ccd: mesh belongs to another context");
        StepSpan span(c);
        ccd_on_mesh(c, m, ms, max_iter, tol, allow_zero_toi, toi, stats);
    });
}

// ccd() that starts from a CALLER'S bound: min(bound, earliest accepted domain below it) -- narrow_phase's toi is in / out
// (narrow_phase.cu:126), and a result below the bound is what a start from 1 returns.  Nothing of the context's own history is used
// or kept (no speculative bound, no redo; the two halves of time by the option and the mesh's size alone): the bound is the caller's
// business -- the ranks of a multi-GPU job start from 1.125 x the REDUCED result of their last step and redo the step only if the
// reduced result is the bound itself (sccd/dist.py GlobalPrior).
extern "C" int sccd_ccd_mesh_from(sccd_ctx* c, const sccd_mesh* m, double ms, int max_iter, double tol, int allow_zero_toi, double bound,
                                  double* toi, sccd_stats* stats)
{
    if (!c || !m || !toi) return SCCD_E_INVALID;
    return guarded(c, [&] {
        SCCD_REQUIRE(m->ctx == c, "ccd: mesh belongs to another context");
        SCCD_REQUIRE(bound > 0 && bound <= 1, "ccd: the bound a call starts from lies in (0, 1]");
        struct HalvesOff {
            sccd_ctx* c;
            ~HalvesOff() { c->two_halves_off = 0; }
        } halves_off { c };
        c->two_halves_off = (c->two_halves == 1 && (long long)m->nE + m->nF < SCCD_TWO_HALVES_MIN_ELEMENTS) ? 1 : 0;
        const double b = c->scalar_f32 ? (double)(float)bound : bound; // (the value the float build's kernels start from: ccd_on_mesh)
        double t = 1.0;
        StepSpan span(c);
        ccd_on_mesh_from(c, m, ms, max_iter, tol, allow_zero_toi, max_iter < 0 ? b : 1.0, &t, stats, nullptr);
        c->toi_guess_mesh = nullptr; // (the context's own history knows nothing of this call)
        *toi = t;
    });
}

constexpr size_t TOI_OUT_MIRROR = 11280; // the source of sccd_ccd_mesh_dev's 8-byte upload in the pinned mirror (common.hpp: h_scalars)
extern "C" int sccd_ccd_mesh_dev(sccd_ctx* c, const sccd_mesh* m, double ms, int max_iter, double tol, int allow_zero_toi,
                                 double* d_toi, double* toi, sccd_stats* stats)
{
    if (!c || !m || !d_toi) return SCCD_E_INVALID;
    return guarded(c, [&] {
        SCCD_REQUIRE(m->ctx == c, "ccd: mesh belongs to another context");
        double t = toi ? *toi : 1.0;
        {
            StepSpan span(c);
            ccd_on_mesh(c, m, ms, max_iter, tol, allow_zero_toi, &t, stats);
        }
        // (the slot is rewritten by the next call's end at the earliest: that call has synchronised with this stream by then)
        double* const slot = reinterpret_cast<double*>(c->h_scalars.as<char>() + TOI_OUT_MIRROR);
        *slot = t;
        SCCD_HIP(hipMemcpyAsync(d_toi, slot, sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (toi) *toi = t;
    });
}

extern "C" int sccd_ccd_mesh_prepare(sccd_ctx* c, const sccd_mesh* m, double ms)
{
    if (!c || !m) return SCCD_E_INVALID;
    return guarded(c, [&] {
        SCCD_REQUIRE(m->ctx == c, "ccd: mesh belongs to another context");
        boxes_from_mesh(c, m, ms, pipeline_of(c), true, true, true);
    });
}

extern "C" int sccd_ccd_mesh_pass(sccd_ctx* c, const sccd_mesh* m, int is_vf, double ms, int max_iter, double tol,
                                  int allow_zero_toi, double* toi, sccd_stats* st)
{
    if (!c || !m || !toi) return SCCD_E_INVALID;
    return guarded(c, [&] {
        SCCD_REQUIRE(m->ctx == c, "ccd: mesh belongs to another context");
        Pipeline* pl = pipeline_of(c);
        SCCD_REQUIRE(pl->vb.n == m->nV && pl->eb.n == m->nE && pl->fb.n == m->nF,
                     "ccd_mesh_pass: call sccd_ccd_mesh_prepare first");
        if (st) std::memset(st, 0, sizeof *st);
        ccd_pass(c, m, pl, &pl->bp, is_vf != 0, ms, max_iter, tol, allow_zero_toi, toi, st);
    });
}

extern "C" int sccd_ccd(sccd_ctx* c, const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                        const int32_t* F, int nF, double ms, int max_iter, double tol, int allow_zero_toi,
                        int memory_limit_GB, double* toi)
{
    if (!c || !toi) return SCCD_E_INVALID;
    const int64_t saved_limit = c->memory_limit_mb; // memory_limit_GB applies to this call (ccd.cu:40-43)
    if (memory_limit_GB > 0) c->memory_limit_mb = (int64_t)memory_limit_GB << 10;
    struct Restore {
        sccd_ctx* c;
        int64_t v;
        ~Restore() { c->memory_limit_mb = v; }
    } restore { c, saved_limit };
    sccd_mesh* m = nullptr;
    int rc = guarded(c, [&] { m = scratch_mesh_from_host(c, V0, V1, nV, E, nE, F, nF, /*defer_verdict=*/true); });
    if (rc != SCCD_OK) return rc;
    double t = *toi; // (the step runs on clamped indices until the verdict is in: its result is discarded with a bad mesh)
    rc = sccd_ccd_mesh(c, m, ms, max_iter, tol, allow_zero_toi, &t, nullptr);
    const int rc_mesh = guarded(c, [&] { mesh_deferred_verdict(c); }); // (an index out of range outranks whatever the step made of it)
    if (rc_mesh == SCCD_OK && rc == SCCD_OK) *toi = t;
    return rc_mesh != SCCD_OK ? rc_mesh : rc;
}

// ccd() with the per-query collision list (ccd.cu:14-78 in a SCALABLE_CCD_TOI_PER_QUERY build): build, then alternate
// detect_overlaps_partial / narrow_phase with per-query output; the pairs stay on the device throughout
static void ccd_pass_collisions(sccd_ctx* c, const sccd_mesh* m, Pipeline* pl, bool vf, double ms, int max_iter, double tol,
                                int allow_zero_toi, double* toi, std::vector<sccd_collision>& acc)
{
    // THE PROJECTION CULL serves the collision list too (round 6): a culled pair has no accepted domain, hence no per-query impact and
    // no record (narrow_phase.cu:84-103 lists toi < 1 only) -- under any check limit as well.  Per-query output prunes a query by its
    // OWN earliest impact alone (root_finder.cu:297): the slab is the whole step, whatever the running TOI is.
    pass_cull_setup(c, &pl->bp, m, vf, ms, max_iter, tol, 1.0);
    pl->bp.cull.slabs = CullSlabs(); // ([0, 1], one list)
    if (vf) bp_build(&pl->bp, &pl->vb, &pl->fb);
    else bp_build(&pl->bp, &pl->eb, nullptr);
    DevBuf& pq = c->col_pq;
    while (pl->bp.cursor < pl->bp.total_rows) {
        bp_detect_partial(&pl->bp);
        const int64_t n = pass_count(&pl->bp);
        if (n > 0) pq.ensure(sizeof(double) * (size_t)n);
        run_narrow(c, m, pass_pairs(&pl->bp), n, vf ? 1 : 0, max_iter, tol, ms, allow_zero_toi, toi,
                   n > 0 ? pq.as<double>() : nullptr);
        copy_out_collisions(c, pass_pairs(&pl->bp), pq.as<double>(), n, acc, /*ordered=*/false);
    }
}

extern "C" int sccd_ccd_collisions(sccd_ctx* c, const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                                   const int32_t* F, int nF, double ms, int max_iter, double tol, int allow_zero_toi,
                                   int memory_limit_GB, double* toi, sccd_collision** collisions, int64_t* n_collisions)
{
    if (!c || !toi || !collisions || !n_collisions) return SCCD_E_INVALID;
    *collisions = nullptr;
    *n_collisions = 0;
    const int64_t saved_limit = c->memory_limit_mb; // memory_limit_GB applies to this call (ccd.cu:40-43)
    if (memory_limit_GB > 0) c->memory_limit_mb = (int64_t)memory_limit_GB << 10;
    struct Restore {
        sccd_ctx* c;
        int64_t v;
        ~Restore() { c->memory_limit_mb = v; }
    } restore { c, saved_limit };
    return guarded(c, [&] {
        sccd_mesh* const m = scratch_mesh_from_host(c, V0, V1, nV, E, nE, F, nF);
        Pipeline* pl = pipeline_of(c);
        boxes_from_mesh(c, m, ms, pl, true, true, true); // inflation radius = min_distance (ccd.cu:112)
        double t = 1;                                    // ccd.cu:125
        std::vector<sccd_collision> acc;
        ccd_pass_collisions(c, m, pl, true, ms, max_iter, tol, allow_zero_toi, &t, acc);
        ccd_pass_collisions(c, m, pl, false, ms, max_iter, tol, allow_zero_toi, &t, acc);
        *collisions = collisions_to_c(acc);
        *n_collisions = (int64_t)acc.size();
        *toi = t;
    });
}

// partial_ipc_ccd_strategy<run_vf> (ipc_ccd_strategy.cu:12-92)
static void ipc_pass(sccd_ctx* c, const sccd_mesh* m, Pipeline* pl, bool vf, double ms, int max_iter, double tol,
                     double* earliest)
{
    // THE PROJECTION CULL serves both runs of a chunk: the list kept for (ms, the slab [0, earliest]) is a superset of what the
    // conservative re-run -- ms = 0, the same start -- could need: the cull's threshold grows with the minimum separation (ms itself
    // and the larger error filter of root_finder.cu:97-113), so a pair beyond reach under ms is beyond reach without it
    pass_cull_setup(c, &pl->bp, m, vf, ms, max_iter, tol, *earliest);
    if (vf) bp_build(&pl->bp, &pl->vb, &pl->fb);
    else bp_build(&pl->bp, &pl->eb, nullptr);
    while (pl->bp.cursor < pl->bp.total_rows) {
        if (pl->bp.cull.on) // (every chunk's cull looks at what is left of the step)
            pl->bp.cull.slabs = narrow_cull_slabs(c, narrow_params(c, m, nullptr, 0, vf ? 1 : 0, max_iter, tol, ms, 1), *earliest);
        bp_detect_partial(&pl->bp);
        const double before = *earliest;
        run_narrow(c, m, pass_pairs(&pl->bp), pass_count(&pl->bp), vf ? 1 : 0, max_iter, tol, ms,
                   /*allow_zero_toi=*/1, earliest, nullptr);
        if (*earliest < 1e-6) { // :72-91: conservative re-run without minimum separation
            *earliest = before;
            run_narrow(c, m, pass_pairs(&pl->bp), pass_count(&pl->bp), vf ? 1 : 0, /*max_iter=*/-1, tol,
                       /*ms=*/0.0, /*allow_zero_toi=*/0, earliest, nullptr);
            *earliest *= 0.8;
        }
    }
}

extern "C" int sccd_ipc_ccd_strategy(sccd_ctx* c, const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                                     const int32_t* F, int nF, double ms, int max_iter, double tol, double* toi)
{
    if (!c || !toi) return SCCD_E_INVALID;
    return guarded(c, [&] {
        sccd_mesh* const m = scratch_mesh_from_host(c, V0, V1, nV, E, nE, F, nF);
        Pipeline* pl = pipeline_of(c);
        boxes_from_mesh(c, m, ms, pl, true, true, true); // ipc_ccd_strategy.cu:123-125
        double earliest = 1.0;                           // :136
        ipc_pass(c, m, pl, true, ms, max_iter, tol, &earliest);
        ipc_pass(c, m, pl, false, ms, max_iter, tol, &earliest);
        *toi = earliest;
    });
}

