// build.hip -- the C ABI of include/sccd.h, part 2: BroadPhase (broad_phase.cuh:15-92, broad_phase.cu:29-252) -- the cell
// grid, the entry lists (one-pass append fill or count -> scan -> fill), the merged sort, the sorted records, the speculative
// build, a rank's window of cells, and detect_overlaps_partial with its overflow rerun, cursor and memory limit.
#include "api_internal.hpp"

extern "C" int sccd_broad_phase_create(sccd_ctx* c, sccd_broad_phase** out)
{
    if (!c || !out) return SCCD_E_INVALID;
    *out = new sccd_broad_phase();
    (*out)->ctx = c;
    return SCCD_OK;
}

extern "C" void sccd_broad_phase_destroy(sccd_broad_phase* bp)
{
    if (!bp) return;
    (void)hipSetDevice(bp->ctx->device);
    (void)hipStreamSynchronize(bp->ctx->stream);
    delete bp;
}

// cell size = factor x mean box extent per minor axis (grid_setup_k); SCCD_OPT_CELL_FACTOR_MILLI: thousandths, 0 = the default 4
static double cell_factor(const sccd_ctx* c)
{
    if (c->cell_factor_milli == 0) return 4.0;
    return c->cell_factor_milli > 0 ? c->cell_factor_milli / 1000.0 : 1e300; // < 0 switches the grid off (one cell)
}

// An item goes to the window its midpoint (in running weight) falls into: boundaries are
// monotone, cover [0, n) and no window is more than one item's weight away from total / parts.
static void shard_bounds(const uint32_t* w, int n, int parts, int* bounds)
{
    unsigned long long total = 0;
    for (int k = 0; k < n; k++) total += w[k];
    bounds[0] = 0;
    unsigned long long run = 0;
    int k = 0;
    for (int r = 1; r < parts; r++) {
        const unsigned long long target = total * (unsigned long long)r / (unsigned long long)parts;
        while (k < n && run + w[k] / 2 < target) run += w[k++];
        bounds[r] = k;
    }
    bounds[parts] = n;
}

extern "C" int sccd_shard_bounds(const uint32_t* weights, int n, int parts, int* bounds)
{
    if (n < 0 || parts < 1 || !bounds || (n > 0 && !weights)) return SCCD_E_INVALID;
    shard_bounds(weights, n, parts, bounds);
    return SCCD_OK;
}

constexpr int SHARD_HIST_STRIDE = 8; // the shard histogram looks at every 8th box

// One list: count -> scan | (host learns the totals of BOTH lists in one round trip) | fill ->
// sort -> gather.
static void list_count(sccd_ctx* c, const sccd_boxes* b, const GridParams* gp, int cell_lo, int cell_hi, SortedList* L,
                       uint32_t* d_total)
{
    const int n = b->n;
    L->m = 0;
    if (n == 0) return;
    L->offsets.ensure(sizeof(uint32_t) * ((size_t)n + 64));
    uint32_t* counts = L->offsets.as<uint32_t>();
    {
        ProfScope ps(c, SCCD_PROF_BOXES);
        launch_cell_count(c, b->raw.as<sccd_aabb>(), n, gp, cell_lo, cell_hi, counts);
    }
    {
        ProfScope ps(c, SCCD_PROF_SORT);
        exclusive_scan_u32(c, counts, counts, n, d_total);
    }
}
// `filled`: key / idx already hold the entries (the one-pass append of the sharded build).  Sorts the list's (key, box
// index) pairs; the records follow once BOTH lists of a build are sorted (a row's first column is looked up among the
// other list's keys).
// d_n_real (speculative build): `total` is the padded number of pairs that is sorted, the real count sits in device memory
static void list_sort(sccd_ctx* c, const sccd_boxes* b, const GridParams* gp, int cell_lo, int cell_hi, uint32_t total,
                      int key_bits, SortedList* L, bool filled = false, const uint32_t* d_n_real = nullptr)
{
    const int n = b->n;
    L->m = 0;
    if (n == 0 || total == 0) return; // (no box of this list touches the rank's cells)
    SCCD_REQUIRE(total < (1u << 31), "broad phase: too many cell entries");
    const size_t m = total, pad = SCCD_LIST_PAD;
    L->m = (int)m;
    if (!filled) {
        L->key.ensure(sizeof(uint32_t) * (m + pad));
        L->idx.ensure(sizeof(uint32_t) * (m + pad));
        ProfScope ps(c, SCCD_PROF_BOXES);
        launch_cell_fill(c, b->raw.as<sccd_aabb>(), n, gp, cell_lo, cell_hi, L->offsets.as<uint32_t>(),
                         L->key.as<uint32_t>(), L->idx.as<uint32_t>());
    }
    {
        ProfScope ps(c, SCCD_PROF_SORT);
        c->sort_tmp_keys.ensure(sizeof(uint32_t) * (m + pad));
        c->sort_tmp_vals.ensure(sizeof(uint32_t) * (m + pad));
        if (radix_sort_pairs_u32(c, L->key.as<uint32_t>(), L->idx.as<uint32_t>(), (int64_t)m, key_bits, d_n_real)) {
            // odd number of passes: the sorted pairs sit in the ping-pong buffers -- swap, no copy
            std::swap(L->key.p, c->sort_tmp_keys.p);
            std::swap(L->key.cap, c->sort_tmp_keys.cap);
            std::swap(L->idx.p, c->sort_tmp_vals.p);
            std::swap(L->idx.cap, c->sort_tmp_vals.cap);
        }
    }
}
// THE RECORDS GATE (drivers.hip, ccd()): the two record kernels of a step -- vertices + faces on the caller's stream, edges on the
// helper's -- are bandwidth-bound and side by side each takes twice its time, but only the first is on the way to the first sweep.
// The helper's kernel is ordered behind the END of the caller's: it then runs beside the vertex-face sweep (instruction issue).
static void records_gate_signal(sccd_ctx* c)
{
    StageGate* g = c->records_gate_signal;
    if (!g || g->recorded) return;
    SCCD_HIP(hipEventRecord(g->ev, c->stream));
    g->recorded = true;
}
static void records_gate_wait(sccd_ctx* c)
{
    // (one thread enqueues both chains, this stream's behind the other's: the event is recorded by now if that chain has a records
    // kernel at all -- a build that took another path leaves the gate open)
    StageGate* g = c->records_gate_wait;
    if (g && g->recorded) SCCD_HIP(hipStreamWaitEvent(c->stream, g->ev, 0));
}
// the sorted records of the lists of a build whose (key, index) pairs are sorted, each list in its own arrays
static void lists_records(sccd_ctx* c, const sccd_boxes* A, const sccd_boxes* B, const GridParams* gp, SortedList* LA,
                          SortedList* LB, const uint32_t* d_tot = nullptr, int expect_bits = 0)
{
    ProfScope ps(c, SCCD_PROF_BOXES);
    if (!B) {
        records_gate_wait(c);
        launch_entry_records(c, A->raw.as<sccd_aabb>(), LA->key.as<uint32_t>(), LA->idx.as<uint32_t>(), LA->m, gp, 0, nullptr,
                             0, false, false, LA, d_tot, expect_bits);
        records_gate_signal(c);
        return;
    }
    SCCD_REQUIRE(!d_tot, "broad phase: device-side counts serve the one-list and the merged two-list build");
    if (LA->m == 0 || LB->m == 0) return;
    records_gate_wait(c);
    launch_entry_records_two(c, A->raw.as<sccd_aabb>(), LA->key.as<uint32_t>(), LA->idx.as<uint32_t>(), LA->m,
                             B->raw.as<sccd_aabb>(), LB->key.as<uint32_t>(), LB->idx.as<uint32_t>(), LB->m, /*b_tagged=*/false, gp, LA, LB);
    records_gate_signal(c);
}

// BroadPhase::build (broad_phase.cu:29-101) together with the key split + sort the reference
// does in the DeviceAABBs constructor (aabb.cu:75-111): the lists are sorted HERE because the
// cell grid is derived from both lists of the build.
// Both lists of a two-list build in one sort: list B's entries (their keys carry the tag bit, the top bit of the
// sorted key) sit in list A's buffers among list A's (the fill placed both by one cursor), the pairs are sorted once, and
// the result is list A followed by list B.
// list A keeps the merged key array (its first total_a entries); list B gets its keys back without the tag from the
// gather.  Lists that were filled by the one-pass append only (entries already in key / idx).
// d_tot (speculative build): total_a / total_b are BOUNDS -- their sum is sorted, padded behind the real pairs -- and the
// real counts {A, B, A + B} sit in device memory
static void lists_finish_merged(sccd_ctx* c, const sccd_boxes* A, const sccd_boxes* B, const GridParams* gp, uint32_t total_a,
                                uint32_t total_b, int key_bits, SortedList* LA, SortedList* LB, const uint32_t* d_tot = nullptr,
                                const uint32_t* d_extq = nullptr)
{
    const size_t ma = total_a, mb = total_b, m = ma + mb, pad = SCCD_LIST_PAD;
    SCCD_REQUIRE(m < (1u << 31), "broad phase: too many cell entries");
    LA->m = (int)ma;
    LB->m = (int)mb;
    SCCD_REQUIRE(LA->key.cap >= sizeof(uint32_t) * (m + pad) && LA->idx.cap >= sizeof(uint32_t) * (m + pad),
                 "broad phase: merged list buffers too small");
    {
        ProfScope ps(c, SCCD_PROF_SORT);
        c->sort_tmp_keys.ensure(sizeof(uint32_t) * (m + pad));
        c->sort_tmp_vals.ensure(sizeof(uint32_t) * (m + pad));
        if (radix_sort_pairs_u32(c, LA->key.as<uint32_t>(), LA->idx.as<uint32_t>(), (int64_t)m, key_bits, d_tot ? d_tot + 2 : nullptr)) {
            std::swap(LA->key.p, c->sort_tmp_keys.p);
            std::swap(LA->key.cap, c->sort_tmp_keys.cap);
            std::swap(LA->idx.p, c->sort_tmp_vals.p);
            std::swap(LA->idx.cap, c->sort_tmp_vals.cap);
        }
    }
    {
        // the sorted pairs: list A's (keys as they are), then list B's (keys with the tag).  Each list's rows look their
        // first column up among the other list's keys, tag and all.
        ProfScope ps(c, SCCD_PROF_BOXES);
        const uint32_t* keys = LA->key.as<uint32_t>();
        const uint32_t* idx = LA->idx.as<uint32_t>();
        records_gate_wait(c);
        launch_entry_records_two(c, A->raw.as<sccd_aabb>(), keys, idx, (int)ma, B->raw.as<sccd_aabb>(), keys + ma, idx + ma, (int)mb,
                                 /*b_tagged=*/true, gp, LA, LB, d_tot, key_bits, d_extq);
        records_gate_signal(c);
    }
}

// a lazy list (internal.hpp) built in full after all: every path but the device-window fill reads the whole raw array
static void materialise(sccd_ctx* c, const sccd_boxes* b)
{
    if (!b || !b->lazy) return;
    sccd_boxes* m = const_cast<sccd_boxes*>(b); // (pipeline-owned: ccd_on_mesh made it lazy)
    if (b->kind == BOX_EDGE)
        launch_edge_boxes(c, b->lazy_vb, reinterpret_cast<const int2*>(b->lazy_elems), b->n, m->raw.as<sccd_aabb>());
    else
        launch_face_boxes(c, b->lazy_vb, reinterpret_cast<const int4*>(b->lazy_elems), b->n, m->raw.as<sccd_aabb>());
    m->lazy = false;
}

static bool speculate_env() { return lab_env().speculate; }
static bool over_budget(uint32_t t, int n) { return (int64_t)t > std::max<int64_t>(3 * (int64_t)n, (int64_t)n + 4096); }

void bp_build(sccd_broad_phase* bp, const sccd_boxes* A, const sccd_boxes* B)
{
    sccd_ctx* c = bp->ctx;
    SCCD_REQUIRE(A != nullptr, "BroadPhase::build: boxes are null");
    bp->A = A;
    bp->B = B;
    bp->built = true;
    bp->cursor = 0;
    bp->n_overlaps = 0;
    bp->n_kept = 0;
    bp->candidates = 0;
    bp->candidates_done = 0;
    bp->la.m = bp->lb.m = 0;
    bp->speculative = false;
    bp->one_class = false;
    bp->la.kind = A->kind;
    bp->lb.kind = B ? B->kind : BOX_UNKNOWN;
    bp->total_rows = 0;
    // two lists with an empty side produce nothing (sort_and_sweep.cpp:221-223)
    if (A->n == 0 || (B && B->n == 0)) return;

    // the grid block: (unused) | params + the lists' totals @512 (GridReadBack) | a rank's cell window @1024 | cell histogram @4096
    bp->grid.ensure(4096 + sizeof(uint32_t) * SCCD_MAX_CELLS);
    GridParams* gp = reinterpret_cast<GridParams*>(bp->grid.as<char>() + 512);
    uint32_t* d_total = reinterpret_cast<uint32_t*>(bp->grid.as<char>() + 512 + offsetof(GridReadBack, total));
    static_assert(512 + sizeof(GridReadBack) <= 1024 && 1024 + sizeof(ShardWindow) <= 4096, "grid buffer layout");
    {
        // lazy lists live on the device-window path only.  (One GPU: computing the edge and face boxes inside the fill was measured
        // in round 4 -- boxes class 0.267 -> 0.275 ms with the passes apart, the step the same: the eager builder stays.)
        if (!(c->shard_count > 1 && !c->build_scan && c->sort_axis >= 0)) {
            materialise(c, A);
            materialise(c, B);
        }
    }
    int axis = c->sort_axis;
    if (axis < 0) axis = pick_sort_axis(c, A->raw.as<sccd_aabb>(), A->n);
    {
        ProfScope ps(c, SCCD_PROF_BOXES);
        ensure_stats(c, A);
        if (B) ensure_stats(c, B);
    }
    const int n_total = A->n + (B ? B->n : 0);
    const double cf = cell_factor(c);
    bp->cell_lo = 0;
    bp->cell_hi = 1 << 30;
    bp->row_shard = false;
    unsigned long long window_est = 0; // entries of this rank's cell window, estimated from the sampled histogram
    // Two lists are sorted in ONE go: the entries of list B carry a tag bit on top of the key, so the sorted array
    // is list A followed by list B (one histogram and one set of radix passes instead of two).  Only the one-pass append
    // build can do it.
    const bool scan_build_env = c->build_scan != 0;
    const bool want_merged = B != nullptr && !scan_build_env;
    // ONE sweep class for vertices x faces: every pair is found from the FACE's row, whose window reaches back over the
    // vertices that start before it (entry_record_body) -- a vertex box is a point's path: tiny along the sort axis -- instead
    // of a second class with the vertices as rows: each list is read once, not twice (other lists: two classes)
    const bool one_class = want_merged && A->kind == BOX_VERTEX && B->kind == BOX_FACE && c->sweep_algo != 1;
    // Multi-GPU, first attempt: the rank's window of cells is dealt out ON THE DEVICE (shard_window_k) and the fill reads it
    // from there -- no host round trip between the histogram and the fill.  What the host would have decided from the
    // histogram (coarsen the grid: too much replication; too few cells to deal out: shard by rows) is checked when the
    // totals come back, on the same GLOBAL numbers, hence alike on every rank; then the build starts over the slow way.
    ShardWindow* const d_win = reinterpret_cast<ShardWindow*>(bp->grid.as<char>() + 1024);
    bool device_window_tried = false, device_window_redo = false;
    for (int shrink = 0;; shrink++) {
        // (a sharded build: the sampled cell histogram behind the grid block is zeroed by the same launch)
        launch_grid_setup(c, A->stats_head(), A->stats_part(), A->n_part, B ? B->stats_head() : nullptr,
                          B ? B->stats_part() : nullptr, B ? B->n_part : 0, n_total, axis, cf, shrink, gp,
                          d_total, want_merged, c->shard_count > 1 ? bp->grid.as<uint32_t>() + 1024 : nullptr);
        const bool can_shrink = shrink < 10;
        // Multi-GPU: every rank takes a contiguous window of cells with an equal share of the
        // entries, and builds / sorts / sweeps only that window.  A pair is reported from exactly
        // one cell (owns_pair), hence by exactly one rank: no exchange of boxes or pairs.
        const bool device_window = c->shard_count > 1 && shrink == 0 && !device_window_tried && !scan_build_env;
        if (device_window) {
            device_window_tried = true;
            uint32_t* d_hist = bp->grid.as<uint32_t>() + 1024; // bytes [4096, ...) of the grid block (zeroed by grid_setup_k)
            launch_cell_hist(c, A, B, gp, SHARD_HIST_STRIDE, d_hist);
            launch_shard_window(c, d_hist, gp, SHARD_HIST_STRIDE, c->shard_rank, c->shard_count, d_win);
            bp->row_shard = false;
        } else if (c->shard_count > 1) {
            uint32_t* d_hist = bp->grid.as<uint32_t>() + 1024; // bytes [4096, ...) of the grid block (zeroed by grid_setup_k)
            launch_cell_hist(c, A, B, gp, SHARD_HIST_STRIDE, d_hist);
            static thread_local std::vector<uint32_t> hist_v(SCCD_MAX_CELLS);
            uint32_t* hist = hist_v.data();
            GridParams hgp;
            {
                // (64 KB: too big for the pinned mirror's small-read area; a plain copy, once per sharded build)
                SCCD_HIP(hipMemcpyAsync(hist, d_hist, sizeof(uint32_t) * SCCD_MAX_CELLS, hipMemcpyDeviceToHost, c->stream));
                ReadBack rb(c);
                rb.add(&hgp, gp, sizeof hgp);
                rb.sync();
            }
            unsigned long long total = 0; // (estimate: the histogram counts every SHARD_HIST_STRIDE-th box)
            for (int k = 0; k < hgp.n_cells; k++) total += (unsigned long long)hist[k] * SHARD_HIST_STRIDE;
            // same replication budget as the single-GPU build, decided on the whole grid so that
            // every rank coarsens alike
            if (can_shrink && total > (unsigned long long)std::max<int64_t>(3 * (int64_t)n_total, (int64_t)n_total + 4096))
                continue;
            if (hgp.n_cells >= 4 * c->shard_count) {
                std::vector<int> bounds(c->shard_count + 1);
                shard_bounds(hist, hgp.n_cells, c->shard_count, bounds.data());
                bp->cell_lo = bounds[c->shard_rank];
                bp->cell_hi = bounds[c->shard_rank + 1];
                bp->row_shard = false;
                window_est = 0;
                for (int k = bp->cell_lo; k < bp->cell_hi; k++) window_est += (unsigned long long)hist[k] * SHARD_HIST_STRIDE;
            } else {
                bp->row_shard = true; // (almost) one cell: every rank sorts everything and takes a slice of the rows
            }
        }
        ShardWindow hwin {};
        auto read_totals = [&](uint32_t (&total)[2], GridParams& hgp) {
            GridReadBack g;
            ReadBack rb(c);
            rb.add(&g, gp, sizeof g);
            if (device_window) rb.add(&hwin, d_win, sizeof hwin);
            rb.sync();
            hgp = g.gp;
            total[0] = g.total[0];
            total[1] = g.total[1];
        };
        const bool windowed_build = c->shard_count > 1 && !bp->row_shard;
        // SCCD_OPT_BUILD_SCAN selects count -> device-wide prefix scan -> fill (entries in box order: a
        // reproducible entry order, 0.15 ms slower per step on the 1M-triangle cloth)
        const bool scan_build = scan_build_env;
        // (a run sharded by ROWS needs the same sorted order on every rank: equal keys must keep box order)
        if (windowed_build || (!scan_build && !bp->row_shard)) {
            // One pass (count + fill by atomic append, a block scan per 1024 boxes) over every box
            // of the list instead of count, scan and fill.  Room for the entries: the replication
            // budget (single GPU) or the histogram estimate (cell window); an overflow is counted,
            // never written, and the pass repeated with exactly enough room.
            // (a window dealt out on the device: its size is not known here -- an even share of the replication budget and
            // a margin; the overflow check below makes up for a wrong guess)
            if (device_window) {
                const unsigned long long share = 3ull * (unsigned long long)std::max(A->n, B ? B->n : 0) / (unsigned long long)c->shard_count;
                window_est = share + share / 4;
            }
            unsigned long long cap = windowed_build ? window_est + window_est / 4 + 65536
                                                    : (unsigned long long)std::max<int64_t>(3 * (int64_t)std::max(A->n, B ? B->n : 0), (int64_t)std::max(A->n, B ? B->n : 0) + 4096);
            uint32_t total[2] = { 0, 0 };
            GridParams hgp;
            for (int fill_round = 0;; fill_round++) {
                SCCD_REQUIRE(cap < (1ull << 31), "broad phase: too many cell entries");
                const size_t pad = 64;
                if (fill_round > 0) SCCD_HIP(hipMemsetAsync(d_total, 0, 4 * sizeof(uint32_t), c->stream)); // (grid_setup_k zeroed them for round 0)
                {
                    ProfScope ps(c, SCCD_PROF_BOXES);
                    // (merged sort: list A's buffers also take list B's entries behind its own)
                    bp->la.key.ensure(sizeof(uint32_t) * ((want_merged ? 2 : 1) * (size_t)cap + pad));
                    bp->la.idx.ensure(sizeof(uint32_t) * ((want_merged ? 2 : 1) * (size_t)cap + pad));
                    // (merged sort: both lists fill list A's buffers, placed by ONE shared cursor -- their entries mix, the sort
                    // separates them by the tag bit; copying list B's entries behind list A's afterwards cost two launches
                    // of the build's latency chain)
                    uint32_t* const d_place = want_merged ? d_total + 2 : nullptr;
                    if (B) {
                        bp->lb.key.ensure(sizeof(uint32_t) * ((size_t)cap + pad));
                        bp->lb.idx.ensure(sizeof(uint32_t) * ((size_t)cap + pad));
                    }
                    const ShardWindow* const win = device_window ? d_win : nullptr;
                    if (want_merged) {
                        launch_cell_fill_append_two(c, A, B, gp, bp->cell_lo, bp->cell_hi, d_total, (uint32_t)(2 * cap),
                                                    bp->la.key.as<uint32_t>(), bp->la.idx.as<uint32_t>(), win);
                    } else {
                        launch_cell_fill_append(c, A, gp, bp->cell_lo, bp->cell_hi, d_total, (uint32_t)cap, bp->la.key.as<uint32_t>(),
                                                bp->la.idx.as<uint32_t>(), false, d_place, win);
                        if (B)
                            launch_cell_fill_append(c, B, gp, bp->cell_lo, bp->cell_hi, d_total + 1, (uint32_t)cap,
                                                    bp->lb.key.as<uint32_t>(), bp->lb.idx.as<uint32_t>(), false, nullptr, win);
                    }
                }
                // THE SPECULATIVE BUILD (internal.hpp sccd_broad_phase::guess): the same lists were built before -- sort,
                // records and (bp_detect_partial) the sweep are enqueued right away for that build's counts plus a margin;
                // the kernels read the real counts on the device and the host checks them when the sweep's counters come back.
                {
                    const sccd_broad_phase::Guess& gs = bp->guess;
                    const bool one_or_merged = !B || want_merged;
                    if (speculate_env() && gs.valid && fill_round == 0 && shrink == 0 && one_or_merged && (device_window || (!windowed_build && c->shard_count == 1))
                        && gs.n_a == A->n && gs.n_b == (B ? B->n : 0) && gs.axis == axis && gs.cell_factor == cf && c->max_overlap_cutoff == 0
                        && c->sweep_algo != 1) {
                        // (the margin: 1/32 of the last build's count, at least 4,096 entries -- for lists of a few thousand entries
                        // a quarter of the count: with 4,096 on top a small list's bound did not fit its buffers and it never
                        // built speculatively)
                        auto margin = [](uint32_t t) { return std::max<uint32_t>(t / 32u, std::min<uint32_t>(4096u, t / 4u + 64u)); };
                        const uint32_t ba = gs.total[0] + margin(gs.total[0]);
                        const uint32_t bb = B ? gs.total[1] + margin(gs.total[1]) : 0u;
                        if (gs.total[0] > 0 && (!B || gs.total[1] > 0) && (unsigned long long)ba + bb <= (want_merged ? 2 : 1) * cap) {
                            bp->spec_bound[0] = ba;
                            bp->spec_bound[1] = bb;
                            bp->spec_sorted = ba + bb;
                            bp->spec_cap = (uint32_t)cap;
                            bp->spec_window = device_window;
                            if (want_merged) {
                                bp->one_class = one_class;
                                lists_finish_merged(c, A, B, gp, ba, bb, gs.key_bits, &bp->la, &bp->lb, d_total, one_class ? d_total + 3 : nullptr);
                            } else {
                                list_sort(c, A, gp, bp->cell_lo, bp->cell_hi, ba, gs.key_bits, &bp->la, true, d_total);
                                lists_records(c, A, nullptr, gp, &bp->la, &bp->lb, d_total, gs.key_bits);
                            }
                            bp->speculative = true;
                            break;
                        }
                    }
                }
                {
                    ProfScope ps(c, SCCD_PROF_SORT);
                    read_totals(total, hgp);
                }
                if (device_window) { // what the host used to decide before the fill, now that the numbers are here
                    bp->cell_lo = hwin.cell_lo;
                    bp->cell_hi = hwin.cell_hi;
                    device_window_redo = (can_shrink && hwin.total_est > (unsigned long long)std::max<int64_t>(3 * (int64_t)n_total, (int64_t)n_total + 4096))
                        || hwin.n_cells < 4 * c->shard_count;
                    if (device_window_redo) break;
                }
                const unsigned long long need = std::max<unsigned long long>(total[0], B ? total[1] : 0);
                if (need <= cap) break;
                if (!windowed_build && can_shrink) break; // over the replication budget: the grid gets coarser below
                cap = need + 1024; // estimate too low (the sample missed a crowded cell): once more, with room
            }
            if (bp->speculative) break; // (everything is enqueued; bp_detect_partial checks the guess)
            if (device_window_redo) { // the same grid again (shrink stays 0), the slow way: histogram on the host, then as before
                device_window_redo = false;
                materialise(c, A); // (the slow way reads whole lists)
                materialise(c, B);
                bp->cell_lo = 0; // (the slow way decides the window -- or the row shard -- afresh)
                bp->cell_hi = 1 << 30;
                shrink--;
                continue;
            }
            if (!windowed_build && can_shrink) {
                if (over_budget(total[0], A->n) || (B && over_budget(total[1], B->n))) continue;
            }
            if (shrink == 0 && (!windowed_build || device_window)) { // what the next build of these lists may expect
                bp->guess.valid = true;
                bp->guess.n_a = A->n;
                bp->guess.n_b = B ? B->n : 0;
                bp->guess.axis = axis;
                bp->guess.cell_factor = cf;
                bp->guess.key_bits = hgp.key_bits;
                bp->guess.total[0] = total[0];
                bp->guess.total[1] = B ? total[1] : 0;
            } else {
                bp->guess.valid = false;
            }
            if (want_merged) {
                // (a side without entries in this rank's cells: no pair can come of it -- sort_and_sweep.cpp:221-223)
                bp->one_class = one_class;
                if (total[0] > 0 && total[1] > 0)
                    lists_finish_merged(c, A, B, gp, total[0], total[1], hgp.key_bits, &bp->la, &bp->lb, nullptr, one_class ? d_total + 3 : nullptr);
                else bp->la.m = bp->lb.m = 0;
            } else {
                list_sort(c, A, gp, bp->cell_lo, bp->cell_hi, total[0], hgp.key_bits, &bp->la, true);
                if (B) list_sort(c, B, gp, bp->cell_lo, bp->cell_hi, total[1], hgp.key_bits, &bp->lb, true);
                lists_records(c, A, B, gp, &bp->la, &bp->lb);
            }
            break;
        }
        list_count(c, A, gp, bp->cell_lo, bp->cell_hi, &bp->la, d_total);
        if (B) list_count(c, B, gp, bp->cell_lo, bp->cell_hi, &bp->lb, d_total + 1);
        uint32_t total[2] = { 0, 0 };
        GridParams hgp;
        {
            ProfScope ps(c, SCCD_PROF_SORT);
            read_totals(total, hgp);
        }
        // replication into cells beyond the budget: coarsen the grid (decided per list, whole grid only)
        const bool windowed = bp->cell_lo > 0 || bp->cell_hi < (1 << 30);
        auto over = [&](uint32_t t, int n) { return (int64_t)t > std::max<int64_t>(3 * (int64_t)n, (int64_t)n + 4096); };
        if (can_shrink && !windowed && (over(total[0], A->n) || (B && over(total[1], B->n)))) continue;
        list_sort(c, A, gp, bp->cell_lo, bp->cell_hi, total[0], hgp.key_bits, &bp->la);
        if (B) list_sort(c, B, gp, bp->cell_lo, bp->cell_hi, total[1], hgp.key_bits, &bp->lb);
        lists_records(c, A, B, gp, &bp->la, &bp->lb);
        break;
    }
    if (B && (bp->la.m == 0 || bp->lb.m == 0)) bp->la.m = bp->lb.m = 0; // nothing to pair in this window
    bp->total_rows = (int64_t)bp->la.m + (B ? bp->lb.m : 0);
}

extern "C" int sccd_broad_phase_build(sccd_broad_phase* bp, const sccd_boxes* A, const sccd_boxes* B)
{
    if (!bp) return SCCD_E_INVALID;
    return guarded(bp->ctx, [&] { bp_build(bp, A, B); });
}

extern "C" int sccd_broad_phase_is_complete(const sccd_broad_phase* bp)
{
    return (!bp || bp->cursor >= bp->total_rows) ? 1 : 0;
}
extern "C" int64_t sccd_broad_phase_num_boxes(const sccd_broad_phase* bp)
{
    if (!bp || !bp->A) return 0;
    return (int64_t)bp->A->n + (bp->B ? bp->B->n : 0);
}
extern "C" int64_t sccd_broad_phase_candidates(const sccd_broad_phase* bp) { return bp ? bp->candidates : 0; }

extern "C" int sccd_boxes_variance_axis(sccd_ctx* c, const sccd_boxes* A, const sccd_boxes* B, int* axis)
{
    if (!c || !A || !axis) return SCCD_E_INVALID;
    return guarded(c, [&] {
        *axis = pick_sort_axis(c, A->raw.as<sccd_aabb>(), A->n, B ? B->raw.as<sccd_aabb>() : nullptr, B ? B->n : 0);
    });
}

// Fallback shard when the grid has too few cells to deal out: an equal slice of the rows.
static void shard_rows(sccd_ctx* c, bool row_shard, int lo, int hi, int* out_lo, int* out_hi)
{
    *out_lo = lo;
    *out_hi = hi;
    if (!row_shard || c->shard_count <= 1 || hi <= lo) return;
    const long long n = hi - lo;
    *out_lo = lo + (int)(n * c->shard_rank / c->shard_count);
    *out_hi = lo + (int)(n * (c->shard_rank + 1) / c->shard_count);
}

// phase 0: the whole step.  phase 1: enqueue the first attempt only (ranges + sweep), no read-back -- ccd() starts the
// edge-edge sweep this way beside the vertex-face narrow phase; phase 2: finish what phase 1 started (read the counters
// back, rerun on overflow as usual).
// A speculative build (bp_build) against what it really had -- the grid and the lists' entry counts (and a rank's cell window)
// as read back from the device: did the guess hold?  Everything the slow build looks at between the fill and the sort.  On
// success the lists' sizes become the real ones and the next build's guess follows the scene; on failure the guess is dropped
// (the caller builds again, the slow way).
static bool speculation_settle(sccd_broad_phase* bp, const GridReadBack& built, const ShardWindow& hwin)
{
    const sccd_broad_phase::Guess& gs = bp->guess;
    const bool two = bp->B != nullptr;
    const uint32_t ta = built.total[0], tb = two ? built.total[1] : 0u;
    const int64_t n_total = (int64_t)bp->A->n + (two ? bp->B->n : 0);
    bool ok_forced_miss = false;
    const bool ok = built.gp.key_bits == gs.key_bits                  // the sort ran the right passes
        && ta > 0 && (!two || tb > 0)                                 // (an empty side ends a build early)
        && ta <= bp->spec_bound[0] && tb <= bp->spec_bound[1]         // records and sweep saw every entry
        && (unsigned long long)ta + tb <= bp->spec_sorted             // ... and so did the sort
        && std::max(ta, tb) <= bp->spec_cap                           // the fill dropped nothing
        && (bp->spec_window                                           // no coarser grid was due, nor a split by rows
                ? !(hwin.total_est > (unsigned long long)std::max<int64_t>(3 * n_total, n_total + 4096)) && hwin.n_cells >= 4 * bp->ctx->shard_count
                : !over_budget(ta, bp->A->n) && !(two && over_budget(tb, bp->B->n)));
    bp->speculative = false;
    if (ok && lab_env().spec_break_every > 0 && (bp->ctx->spec_hits + bp->ctx->spec_misses + 1) % lab_env().spec_break_every == 0)
        ok_forced_miss = true; // (measurement: the guess held, the build is redone as if it had not)
    (ok && !ok_forced_miss ? bp->ctx->spec_hits : bp->ctx->spec_misses) += 1;
    if (ok_forced_miss) {
        bp->guess.valid = false; // (like a real miss: the rebuild waits for its counts and becomes the next guess)
        return false;
    }
    if (!ok) {
        bp->guess.valid = false;
        return false;
    }
    if (bp->spec_window) {
        bp->cell_lo = hwin.cell_lo;
        bp->cell_hi = hwin.cell_hi;
    }
    bp->la.m = (int)ta;
    bp->lb.m = (int)tb;
    bp->total_rows = (int64_t)ta + tb;
    bp->guess.total[0] = ta;
    bp->guess.total[1] = tb;
    return true;
}

void bp_detect_partial(sccd_broad_phase* bp, int phase)
{
    sccd_ctx* c = bp->ctx;
    if (!bp->built) throw SccdError { SCCD_E_NOT_BUILT, "Must initialize build broad phase before detecting overlaps!" };
    bp->n_overlaps = 0;
    bp->n_kept = 0;
    if (phase != 2) bp->sweeps_in_call = 0;
    if (bp->speculative && phase != 2 && (c->max_overlap_cutoff > 0 || c->sweep_algo == 1)) {
        // the options were changed between build and sweep to ones a speculative sweep does not serve (chunks of rows, the
        // plain sweep): read what was built now, and go on with real sizes -- or build again
        GridReadBack built;
        ShardWindow hwin {};
        {
            ReadBack rb(c);
            rb.add(&built, bp->grid.as<char>() + 512, sizeof built);
            if (bp->spec_window) rb.add(&hwin, bp->grid.as<char>() + 1024, sizeof hwin);
            rb.sync();
        }
        if (!speculation_settle(bp, built, hwin)) bp_build(bp, bp->A, bp->B);
    }
    if (bp->cursor >= bp->total_rows) return;
    const SortedList* A = &bp->la;
    const SortedList* B = bp->B ? &bp->lb : nullptr;
    const GridParams* gp = reinterpret_cast<const GridParams*>(bp->grid.as<char>() + 512);
    const int64_t cutoff = c->max_overlap_cutoff > 0 ? c->max_overlap_cutoff : bp->total_rows;
    const int64_t chunk_lo = bp->cursor;
    int64_t chunk_hi = std::min(bp->cursor + cutoff, bp->total_rows);

    SweepCounters* d_cnt = c->scalars.as<SweepCounters>();
    // SCCD_OPT_SWEEP_ALGO: 0 / 2 / 3 the band sweep (window staging -> skewed filter -> queue -> confirm), 1 plain SAP cross-check.
    // Capacity sizing (MemoryHandler, memory_handler.cpp:11-79): the overlap list may use half of
    // the memory limit (SCCD_OPT_MEMORY_LIMIT_MB / ccd()'s memory_limit_GB; default: whatever
    // hipMalloc grants).  A chunk whose pairs do not fit is re-swept over HALF its rows
    // (MAX_OVERLAP_CUTOFF >>= 1, memory_handler.cpp:64-72) and the cursor advances by what was done.
    const int64_t limit_pairs = c->memory_limit_mb > 0
        ? std::max<int64_t>(1024, (c->memory_limit_mb << 20) / 2 / (int64_t)sizeof(int2))
        : (int64_t)1 << 40;
    if (bp->capacity == 0) {
        int64_t cap = c->overlap_capacity > 0 ? c->overlap_capacity : std::max<int64_t>(1 << 20, 32 * bp->total_rows);
        cap = std::min(cap, limit_pairs);
        for (;;) {
            try {
                bp->overlaps.ensure(sizeof(int2) * (size_t)cap);
                break;
            } catch (const SccdError& e) {
                if (e.code != SCCD_E_NOMEM || cap <= (1 << 16)) throw;
                (void)hipGetLastError();
                cap /= 2;
            }
        }
        bp->capacity = cap;
    }
    int64_t chunk_rows = chunk_hi - chunk_lo;
    for (int attempt = 0;; attempt++) { // overflow -> exact-size rerun (broad_phase.cu:142-203)
        chunk_hi = chunk_lo + chunk_rows;
        // rows of this chunk per sweep class
        int a_lo = (int)std::min<int64_t>(chunk_lo, A->m), a_hi = (int)std::min<int64_t>(chunk_hi, A->m);
        int b_lo = 0, b_hi = 0;
        if (B) {
            b_lo = (int)std::max<int64_t>(0, chunk_lo - A->m);
            b_hi = (int)std::max<int64_t>(0, chunk_hi - A->m);
        }
        shard_rows(c, bp->row_shard, a_lo, a_hi, &a_lo, &a_hi);
        if (B) shard_rows(c, bp->row_shard, b_lo, b_hi, &b_lo, &b_hi);

        if (phase == 2 && attempt == 0) goto launched; // (phase 1 enqueued this attempt)
        // pairs and candidate tests of THIS attempt (the first sweep behind a build: its grid kernel zeroed them)
        if (c->sweep_cnt_cleared && attempt == 0) c->sweep_cnt_cleared = false;
        else SCCD_HIP(hipMemsetAsync(d_cnt, 0, sizeof(SweepCounters), c->stream));
        {
            ProfScope ps(c, (!B && bp->la.kind == BOX_EDGE) ? SCCD_PROF_SWEEP_EE : SCCD_PROF_SWEEP);
            // (a speculative build: the lists' sizes are bounds, the kernels take the real counts from device memory)
            const uint32_t* const d_tot = bp->speculative
                ? reinterpret_cast<const uint32_t*>(bp->grid.as<char>() + 512 + offsetof(GridReadBack, total)) : nullptr;
            if (!B) {
                launch_sweep(c, A, A, gp, a_lo, a_hi, EMIT_ONE_LIST, bp->overlaps.as<int2>(), bp->capacity, d_cnt, d_tot, d_tot, bp->guess.key_bits);
            } else if (bp->one_class) { // (list B's rows only: their windows reach back -- bp_build)
                launch_sweep(c, B, A, gp, b_lo, b_hi, EMIT_ROWS_B, bp->overlaps.as<int2>(), bp->capacity, d_cnt, d_tot ? d_tot + 1 : nullptr,
                             d_tot, bp->guess.key_bits);
            } else {
                launch_sweep_two(c, A, B, gp, a_lo, a_hi, b_lo, b_hi, bp->overlaps.as<int2>(), bp->capacity, d_cnt, d_tot, bp->guess.key_bits);
            }
        }
        bp->sweeps_in_call += 1;
        if (bp->after_sweep) {
            const std::function<void()> h = std::move(bp->after_sweep);
            bp->after_sweep = nullptr;
            h();
        }
        if (bp->cull.on) { // (behind EVERY sweep of this pass, first attempts and reruns alike)
            ProfScope ps(c, SCCD_PROF_CULL);
            bp->kept.ensure(sizeof(int2) * (size_t)bp->capacity);
            NarrowParams p {};
            p.V = bp->cull.mesh->V.as<double>();
            p.E = bp->cull.mesh->E.as<int2>();
            p.F = bp->cull.mesh->F.as<int4>();
            p.pairs = bp->overlaps.as<int2>();
            p.is_vf = bp->cull.is_vf;
            p.ms = bp->cull.ms;
            p.tol = bp->cull.tol;
            // (the first slab of the pass's time; the second half's list is made between the walk kernel's two launches, if at all)
            if (bp->cull.slabs.two) bp->kept_b.ensure(sizeof(int2) * (size_t)bp->capacity);
            narrow_cull_launch(c, p, &d_cnt->n_pairs, (long long)bp->capacity, bp->kept.as<int2>(), &d_cnt->n_kept, 0.0,
                               bp->cull.slabs.two ? bp->cull.slabs.t_mid : bp->cull.slabs.t_end);
        }
        if (phase == 1) return;
    launched:
        SweepCounters h;
        GridReadBack built; // (speculative build: the grid and the entry counts it really had)
        ShardWindow hwin {}; // (... of a rank of a multi-GPU job: the cell window it was dealt on the device)
        if (bp->pre_read && attempt == 0) { // (they came with the pass's verdict: sccd_broad_phase::pre_read)
            h = bp->pre_read->h;
            built = bp->pre_read->built;
            hwin = bp->pre_read->hwin;
            bp->pre_read = nullptr;
            bp->rb_ctx = nullptr;
        } else {
            bp->pre_read = nullptr;
            sccd_ctx* const rc = (bp->rb_ctx && attempt == 0) ? bp->rb_ctx : c; // (see sccd_broad_phase::rb_ctx)
            if (rc != c) SCCD_HIP(hipStreamWaitEvent(rc->stream, bp->rb_after, 0));
            bp->rb_ctx = nullptr;
            ReadBack rb(rc);
            rb.add(&h, d_cnt, sizeof h);
            if (bp->speculative) rb.add(&built, bp->grid.as<char>() + 512, sizeof built);
            if (bp->speculative && bp->spec_window) rb.add(&hwin, bp->grid.as<char>() + 1024, sizeof hwin);
            rb.sync();
        }
        if (bp->speculative) {
            if (!speculation_settle(bp, built, hwin)) {
                // build again, the slow way (the guess is gone: bp_build waits for the counts), and sweep that
                const int64_t done = bp->candidates_done;
                bp_build(bp, bp->A, bp->B);
                bp->candidates_done = done;
                bp_detect_partial(bp, 0);
                bp->sweeps_in_call += 1; // (the sweep of the failed guess counts: the first attempt did NOT stand)
                return;
            }
            chunk_hi = bp->total_rows; // (a speculative build is swept in one chunk: bp_build)
            // ... and a re-sweep after an overflow covers the SETTLED rows, not the padded bounds the first sweep was launched
            // for: it runs without the device-side counts, and rows behind a list's real end hold whatever an earlier build
            // left there -- stale records at best, wild ids at worst (a fault in the narrow kernel: round 4, the memory-limit test
            // run on a fresh context)
            chunk_rows = chunk_hi - chunk_lo;
        }
        {
            unsigned long long cs = 0;
            for (int k = 0; k < 32; k++) cs += h.cand_parts[k];
            bp->candidates = bp->candidates_done + (int64_t)cs; // (a chunk swept again after an overflow counts once)
            if (lab_env().sweep_diag)
                std::fprintf(stderr, "[sweep] rows %lld pairs %llu tests %llu | filter blocks %llu groups %llu confirm rounds %llu segments staged %llu\n",
                             (long long)(chunk_hi - chunk_lo), (unsigned long long)h.n_pairs, cs, h.diag[0], h.diag[1], h.diag[2], h.diag[3]);
        }
        if ((int64_t)h.n_pairs <= bp->capacity) {
            bp->n_overlaps = (int64_t)h.n_pairs;
            bp->n_kept = bp->cull.on ? (int64_t)h.n_kept : bp->n_overlaps;
            break;
        }
        SCCD_REQUIRE(attempt < 64, "broad phase: overlap buffer keeps overflowing");
        const int64_t want = (int64_t)h.n_pairs + (int64_t)h.n_pairs / 16 + 1024;
        bool grown = false;
        if (want <= limit_pairs) {
            try {
                bp->overlaps.ensure(sizeof(int2) * (size_t)want);
                bp->capacity = want;
                grown = true;
            } catch (const SccdError& e) {
                if (e.code != SCCD_E_NOMEM) throw;
                (void)hipGetLastError();
                // the old buffer was released by ensure(): get the previous size back
                bp->overlaps.ensure(sizeof(int2) * (size_t)bp->capacity);
            }
        }
        if (!grown) {
            if (chunk_rows <= 1)
                throw SccdError { SCCD_E_NOMEM, "Insufficient memory to increase overlap size; cannot allocate even a single box's overlaps." };
            chunk_rows = (chunk_rows + 1) / 2;
        }
    }
    bp->candidates_done = bp->candidates;
    bp->cursor = chunk_hi; // thread_start_box_id += MAX_OVERLAP_CUTOFF (broad_phase.cu:207)
}

extern "C" int sccd_broad_phase_detect_overlaps_partial(sccd_broad_phase* bp, const int32_t** d_pairs, int64_t* n)
{
    if (!bp) return SCCD_E_INVALID;
    return guarded(bp->ctx, [&] {
        bp_detect_partial(bp);
        if (d_pairs) *d_pairs = bp->overlaps.as<int32_t>();
        if (n) *n = bp->n_overlaps;
    });
}

extern "C" int sccd_broad_phase_detect_overlaps(sccd_broad_phase* bp, int32_t** pairs, int64_t* n)
{
    if (!bp || !pairs || !n) return SCCD_E_INVALID;
    *pairs = nullptr;
    *n = 0;
    return guarded(bp->ctx, [&] {
        sccd_ctx* c = bp->ctx;
        if (!bp->built) throw SccdError { SCCD_E_NOT_BUILT, "Must initialize build broad phase before detecting overlaps!" };
        std::vector<int32_t> acc;
        int64_t cand = 0;
        while (bp->cursor < bp->total_rows) { // broad_phase.cu:236-247
            bp_detect_partial(bp);
            cand = bp->candidates;
            const size_t at = acc.size();
            acc.resize(at + 2 * (size_t)bp->n_overlaps);
            if (bp->n_overlaps) {
                SCCD_HIP(hipMemcpyAsync(acc.data() + at, bp->overlaps.p, sizeof(int2) * (size_t)bp->n_overlaps,
                                        hipMemcpyDeviceToHost, c->stream));
                SCCD_HIP(hipStreamSynchronize(c->stream));
            }
        }
        bp->candidates = cand;
        int32_t* o = (int32_t*)std::malloc(std::max<size_t>(8, acc.size() * sizeof(int32_t)));
        if (!o) throw SccdError { SCCD_E_NOMEM, "host allocation failed" };
        if (!acc.empty()) std::memcpy(o, acc.data(), acc.size() * sizeof(int32_t));
        *pairs = o;
        *n = (int64_t)(acc.size() / 2);
    });
}

