// api.hip -- the C ABI of include/sccd.h: context, HBM-resident objects and the host drivers
// (ccd(), BroadPhase, narrow_phase, ipc_ccd_strategy of the reference, see sccd.h for the
// file:line each entry point replaces).  Host code only; kernels live in the other .hip files.
#include "internal.hpp"
#include "grid.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

// ------------------------------------------------------------------------------------------
// error trampolines
static thread_local std::string g_create_error;

template <class Fn> static int guarded(sccd_ctx* c, Fn&& fn)
{
    try {
        if (c) SCCD_HIP(hipSetDevice(c->device));
        fn();
        return SCCD_OK;
    } catch (const SccdError& e) {
        (void)hipGetLastError(); // (sticky: a later launch check must not trip over this call's failure)
        if (c) c->err = e.msg;
        else g_create_error = e.msg;
        return e.code;
    } catch (const std::bad_alloc&) {
        if (c) c->err = "host allocation failed";
        return SCCD_E_NOMEM;
    } catch (const std::exception& e) {
        if (c) c->err = e.what();
        return SCCD_E_INVALID;
    }
}

// One persistent helper thread per context: ccd() hands it the construction of the edge-edge lists.
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, busy = false, quit = false;
    std::exception_ptr err;
    void submit(std::function<void()> f)
    {
        std::unique_lock<std::mutex> lk(m);
        if (!th.joinable()) th = std::thread([this] { loop(); });
        job = std::move(f);
        has_job = busy = true;
        err = nullptr;
        cv.notify_all();
    }
    void wait() // rethrows what the job threw
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return !busy; });
        if (err) {
            std::exception_ptr e = err;
            err = nullptr;
            std::rethrow_exception(e);
        }
    }
    void loop()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [this] { return has_job || quit; });
                if (quit) return;
                f = std::move(job);
                has_job = false;
            }
            std::exception_ptr e;
            try {
                f();
            } catch (...) {
                e = std::current_exception();
            }
            std::unique_lock<std::mutex> lk(m);
            err = e;
            busy = false;
            cv.notify_all();
        }
    }
    ~Worker()
    {
        {
            std::unique_lock<std::mutex> lk(m);
            quit = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

// pipeline objects cached in the context so that repeated ccd() calls allocate nothing
struct Pipeline {
    sccd_boxes vb, eb, fb; // boxes in element order (raw)
    sccd_broad_phase bp;
    sccd_broad_phase bp_ee; // edge-edge lists of ccd(): belongs to the helper context c->side
    Worker worker;
};
static Pipeline* pipeline_of(sccd_ctx* c)
{
    if (!c->pipeline) {
        auto* p = new Pipeline();
        p->vb.ctx = p->eb.ctx = p->fb.ctx = c;
        p->bp.ctx = c;
        c->pipeline = p;
    }
    return static_cast<Pipeline*>(c->pipeline);
}

void sccd_collect_profile(sccd_ctx* c)
{
    for (auto& pe : c->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(pe.b) == hipSuccess && hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess)
            c->prof_ms[pe.cls] += ms;
        c->event_pool.push_back(pe.a);
        c->event_pool.push_back(pe.b);
    }
    c->pending.clear();
}

// what the helper context ran (ccd(): the edge-edge lists, sweep and narrow kernel) belongs to this context's account
static void merge_side_profile(sccd_ctx* c)
{
    sccd_collect_profile(c);
    if (!c->side) return;
    SCCD_HIP(hipStreamSynchronize(c->side->stream));
    sccd_collect_profile(c->side);
    for (int k = 0; k < SCCD_PROF_COUNT; k++) {
        c->prof_ms[k] += c->side->prof_ms[k];
        c->prof_launches[k] += c->side->prof_launches[k];
        c->side->prof_ms[k] = 0;
        c->side->prof_launches[k] = 0;
    }
}

extern "C" {

const char* sccd_version(void) { return "sccd-hip 0.2 (gfx950)"; } // 0.2: default contract fused (SCCD_OPT_ARITH = 1), option id 12 retired

int sccd_create(int device, sccd_ctx** out)
{
    if (!out) return SCCD_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        g_create_error = "no HIP device available (libsccd_hip has no CPU fallback)";
        return SCCD_E_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        g_create_error = "device ordinal out of range";
        return SCCD_E_NO_DEVICE;
    }
    sccd_ctx* c = new sccd_ctx();
    c->device = device;
    const int rc = guarded(c, [&] {
        hipDeviceProp_t prop;
        SCCD_HIP(hipGetDeviceProperties(&prop, device));
        c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        SCCD_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
        c->scalars.ensure(4096);
        SCCD_HIP(hipMemsetAsync(c->scalars.p, 0, 4096, c->stream)); // (holds a counter that is never reset: sort.hip)
        c->h_scalars.ensure(16384);
    });
    if (rc != SCCD_OK) {
        g_create_error = c->err;
        delete c;
        return rc;
    }
    *out = c;
    return SCCD_OK;
}

void sccd_destroy(sccd_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    delete static_cast<Pipeline*>(c->pipeline);
    c->pipeline = nullptr;
    delete c->scratch_mesh;
    c->scratch_mesh = nullptr;
    sccd_collect_profile(c);
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->rb_event) (void)hipEventDestroy(c->rb_event);
    if (c->side_event) (void)hipEventDestroy(c->side_event);
    if (c->side) sccd_destroy(c->side);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* sccd_last_error(const sccd_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int sccd_set_stream(sccd_ctx* c, void* s)
{
    if (!c) return SCCD_E_INVALID;
    return guarded(c, [&] {
        SCCD_HIP(hipStreamSynchronize(c->stream));
        if (s) {
            if (c->own_stream) SCCD_HIP(hipStreamDestroy(c->stream));
            c->stream = (hipStream_t)s;
            c->own_stream = false;
        } else if (!c->own_stream) {
            SCCD_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
            c->own_stream = true;
        }
    });
}

void* sccd_get_stream(const sccd_ctx* c) { return c ? (void*)c->stream : nullptr; }

int sccd_synchronize(sccd_ctx* c)
{
    if (!c) return SCCD_E_INVALID;
    return guarded(c, [&] { SCCD_HIP(hipStreamSynchronize(c->stream)); });
}

int sccd_set_option(sccd_ctx* c, int opt, int64_t v)
{
    if (!c) return SCCD_E_INVALID;
    switch (opt) {
    case SCCD_OPT_ARITH: c->arith = v ? 1 : 0; break;
    case SCCD_OPT_NARROW_ALGO: c->narrow_algo = v ? 1 : 0; break;
    case SCCD_OPT_SWEEP_ALGO:
        if (v < 0 || v > 3) return SCCD_E_INVALID;
        c->sweep_algo = (int)v;
        break;
    case SCCD_OPT_SORT_AXIS:
        if (v < -1 || v > 2) return SCCD_E_INVALID;
        c->sort_axis = (int)v;
        break;
    case SCCD_OPT_SHARD_RANK: c->shard_rank = (int)v; break;
    case SCCD_OPT_SHARD_COUNT:
        if (v < 1) return SCCD_E_INVALID;
        c->shard_count = (int)v;
        break;
    case SCCD_OPT_OVERLAP_CAPACITY: c->overlap_capacity = v; break;
    case SCCD_OPT_PROFILE: c->profile = (int)v; break; // 0 off, 1 every class, else (mask of classes) << 1
    case SCCD_OPT_MAX_OVERLAP_CUTOFF: c->max_overlap_cutoff = v; break;
    case SCCD_OPT_MEMORY_LIMIT_MB: c->memory_limit_mb = v; break;
    case SCCD_OPT_SCALAR: c->scalar_f32 = v ? 1 : 0; break;
    case SCCD_OPT_LIMIT_LEVEL_ORDER: c->limit_level_order = v ? 1 : 0; break;
    case SCCD_OPT_PASSES_APART: c->passes_apart = v ? 1 : 0; break;
    case SCCD_OPT_SPEC_HITS:
    case SCCD_OPT_SPEC_MISSES: // (counters: any value resets both, here and on the helper context)
        c->spec_hits = c->spec_misses = 0;
        if (c->side) c->side->spec_hits = c->side->spec_misses = 0;
        break;
    case 12: c->err = "option 12 is retired (SCCD_OPT_MAX_ITER_FAST of 0.1): see SCCD_OPT_LIMIT_LEVEL_ORDER"; return SCCD_E_INVALID;
    default: c->err = "unknown option"; return SCCD_E_INVALID;
    }
    return SCCD_OK;
}

int64_t sccd_get_option(const sccd_ctx* c, int opt)
{
    if (!c) return 0;
    switch (opt) {
    case SCCD_OPT_ARITH: return c->arith;
    case SCCD_OPT_NARROW_ALGO: return c->narrow_algo;
    case SCCD_OPT_SWEEP_ALGO: return c->sweep_algo;
    case SCCD_OPT_SORT_AXIS: return c->sort_axis;
    case SCCD_OPT_SHARD_RANK: return c->shard_rank;
    case SCCD_OPT_SHARD_COUNT: return c->shard_count;
    case SCCD_OPT_OVERLAP_CAPACITY: return c->overlap_capacity;
    case SCCD_OPT_PROFILE: return c->profile;
    case SCCD_OPT_MAX_OVERLAP_CUTOFF: return c->max_overlap_cutoff;
    case SCCD_OPT_MEMORY_LIMIT_MB: return c->memory_limit_mb;
    case SCCD_OPT_SCALAR: return c->scalar_f32;
    case SCCD_OPT_LIMIT_LEVEL_ORDER: return c->limit_level_order;
    case SCCD_OPT_PASSES_APART: return c->passes_apart;
    case SCCD_OPT_SPEC_HITS: return c->spec_hits + (c->side ? c->side->spec_hits : 0);
    case SCCD_OPT_SPEC_MISSES: return c->spec_misses + (c->side ? c->side->spec_misses : 0);
    default: return 0;
    }
}

int sccd_get_profile(sccd_ctx* c, double ms[SCCD_PROF_COUNT], int64_t launches[SCCD_PROF_COUNT])
{
    if (!c) return SCCD_E_INVALID;
    return guarded(c, [&] {
        SCCD_HIP(hipStreamSynchronize(c->stream));
        merge_side_profile(c);
        for (int k = 0; k < SCCD_PROF_COUNT; k++) {
            if (ms) ms[k] = c->prof_ms[k];
            if (launches) launches[k] = c->prof_launches[k];
        }
    });
}

int sccd_reset_profile(sccd_ctx* c)
{
    if (!c) return SCCD_E_INVALID;
    return guarded(c, [&] {
        SCCD_HIP(hipStreamSynchronize(c->stream));
        sccd_collect_profile(c);
        for (int k = 0; k < SCCD_PROF_COUNT; k++) {
            c->prof_ms[k] = 0;
            c->prof_launches[k] = 0;
        }
        if (c->side) {
            SCCD_HIP(hipStreamSynchronize(c->side->stream));
            sccd_collect_profile(c->side);
            for (int k = 0; k < SCCD_PROF_COUNT; k++) {
                c->side->prof_ms[k] = 0;
                c->side->prof_launches[k] = 0;
            }
        }
    });
}

void sccd_free(void* p) { std::free(p); }

} // extern "C"

// ------------------------------------------------------------------------------------------
// mesh
static void copy_in(sccd_ctx* c, void* dst, const void* src, size_t bytes, int src_on_device)
{
    if (bytes == 0) return;
    SCCD_HIP(hipMemcpyAsync(dst, src, bytes, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                            c->stream));
}

static void mesh_set_vertices(sccd_mesh* m, const double* V0, const double* V1, int src_on_device)
{
    sccd_ctx* c = m->ctx;
    const size_t nb = sizeof(double) * 3 * (size_t)m->nV;
    const double *d0 = V0, *d1 = V1;
    if (!src_on_device) {
        c->tmp0.ensure(nb);
        c->tmp1.ensure(nb);
        copy_in(c, c->tmp0.p, V0, nb, 0);
        copy_in(c, c->tmp1.p, V1, nb, 0);
        d0 = c->tmp0.as<double>();
        d1 = c->tmp1.as<double>();
    }
    launch_pack_vertices(c, d0, d1, m->nV, m->V.as<double>());
    SCCD_HIP(hipStreamSynchronize(c->stream)); // borrowed inputs may go away after return
}

// (Re)fills a mesh from the caller's matrices.  Index matrices are validated on the device while they are packed
// (pack_edges_k / pack_faces_k: an out-of-range vertex index would otherwise turn into wild gathers in the box builders
// and the narrow phase -- the reference asserts nothing and would fault); the verdict comes back with the one
// synchronisation this function ends with anyway (borrowed inputs may go away after return).
constexpr size_t MESH_VERDICT_MIRROR = 11264; // the deferred verdict's slot in the pinned mirror (common.hpp: h_scalars)
static void mesh_fill(sccd_ctx* c, sccd_mesh* m, const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                      const int32_t* F, int nF, int src_on_device, bool defer_verdict = false)
{
    SCCD_REQUIRE(nV >= 0 && nE >= 0 && nF >= 0, "mesh: negative size");
    SCCD_REQUIRE((nV == 0 || (V0 && V1)) && (nE == 0 || E) && (nF == 0 || F), "mesh: null matrix");
    SCCD_REQUIRE(nV > 0 || (nE == 0 && nF == 0), "mesh: edge or face index out of range"); // (no vertex to index)
    m->ctx = c;
    m->nV = nV;
    m->nE = nE;
    m->nF = nF;
    m->V.ensure(sizeof(double) * 6 * (size_t)std::max(nV, 1));
    m->E.ensure(sizeof(int2) * (size_t)std::max(nE, 1));
    m->F.ensure(sizeof(int4) * (size_t)std::max(nF, 1));
    // staging: [the verdict word | raw E | raw F] (host sources)
    c->tmp2.ensure(sizeof(int32_t) * (2 * (size_t)nE + 3 * (size_t)nF + 4));
    unsigned* const d_bad = c->tmp2.as<unsigned>();
    const int32_t *dE = E, *dF = F;
    const double *d0 = V0, *d1 = V1;
    if (!src_on_device) {
        // the four uploads back to back (a copy from pageable memory blocks the host: anything issued between two of them
        // -- the pack kernels used to be -- costs the link 15-50 us of idle time), then everything that works on them
        const size_t nb = sizeof(double) * 3 * (size_t)nV;
        c->tmp0.ensure(nb);
        c->tmp1.ensure(nb);
        int32_t* t = c->tmp2.as<int32_t>() + 4;
        copy_in(c, c->tmp0.p, V0, nb, 0);
        copy_in(c, c->tmp1.p, V1, nb, 0);
        copy_in(c, t, E, sizeof(int32_t) * 2 * (size_t)nE, 0);
        copy_in(c, t + 2 * (size_t)nE, F, sizeof(int32_t) * 3 * (size_t)nF, 0);
        dE = t;
        dF = t + 2 * (size_t)nE;
        d0 = c->tmp0.as<double>();
        d1 = c->tmp1.as<double>();
    }
    SCCD_HIP(hipMemsetAsync(d_bad, 0, sizeof(unsigned), c->stream));
    launch_pack_vertices(c, d0, d1, nV, m->V.as<double>());
    launch_pack_edges(c, dE, nE, nV, m->E.as<int2>(), d_bad);
    launch_pack_faces(c, dF, nF, nV, m->F.as<int4>(), d_bad);
    if (defer_verdict) {
        // ccd() from host matrices: the step is enqueued right behind the packing, and the verdict is looked at when the call
        // has synchronised anyway (mesh_deferred_verdict; the clamped indices keep the step harmless meanwhile)
        SCCD_HIP(hipMemcpyAsync(c->h_scalars.as<char>() + MESH_VERDICT_MIRROR, d_bad, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        return;
    }
    unsigned bad = 0;
    {
        ReadBack rb(c); // (also the synchronisation this function owes its caller: borrowed inputs may go away after return)
        rb.add(&bad, d_bad, sizeof bad);
        rb.sync();
    }
    SCCD_REQUIRE(bad == 0, "mesh: edge or face index out of range");
}
static void mesh_deferred_verdict(sccd_ctx* c)
{
    SCCD_HIP(hipStreamSynchronize(c->stream));
    unsigned bad = 0;
    std::memcpy(&bad, c->h_scalars.as<char>() + MESH_VERDICT_MIRROR, sizeof bad);
    SCCD_REQUIRE(bad == 0, "mesh: edge or face index out of range");
}

extern "C" int sccd_mesh_create(sccd_ctx* c, const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                                const int32_t* F, int nF, int src_on_device, sccd_mesh** out)
{
    if (!c || !out) return SCCD_E_INVALID;
    *out = nullptr;
    return guarded(c, [&] {
        std::unique_ptr<sccd_mesh> m(new sccd_mesh());
        mesh_fill(c, m.get(), V0, V1, nV, E, nE, F, nF, src_on_device);
        *out = m.release();
    });
}

// The mesh behind the drivers that take HOST matrices (ccd(), ccd() with collisions, ipc_ccd_strategy()): owned by the
// context and refilled call after call -- three allocations and three frees per call were 1 ms of a 5.6 ms ccd() on the
// 1M-triangle cloth.
static sccd_mesh* scratch_mesh_from_host(sccd_ctx* c, const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                                         const int32_t* F, int nF, bool defer_verdict = false)
{
    if (!c->scratch_mesh) c->scratch_mesh = new sccd_mesh();
    mesh_fill(c, c->scratch_mesh, V0, V1, nV, E, nE, F, nF, 0, defer_verdict);
    return c->scratch_mesh;
}

extern "C" int sccd_mesh_assign(sccd_mesh* m, const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                                const int32_t* F, int nF, int src_on_device)
{
    if (!m) return SCCD_E_INVALID;
    return guarded(m->ctx, [&] { mesh_fill(m->ctx, m, V0, V1, nV, E, nE, F, nF, src_on_device); });
}

extern "C" int sccd_mesh_update_vertices(sccd_mesh* m, const double* V0, const double* V1, int src_on_device)
{
    if (!m) return SCCD_E_INVALID;
    return guarded(m->ctx, [&] {
        SCCD_REQUIRE(m->nV == 0 || (V0 && V1), "mesh: null matrix");
        mesh_set_vertices(m, V0, V1, src_on_device);
    });
}

extern "C" void sccd_mesh_destroy(sccd_mesh* m)
{
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    delete m;
}

// ------------------------------------------------------------------------------------------
// boxes

extern "C" int sccd_build_vertex_boxes(sccd_ctx* c, const double* V0, const double* V1, int nV, double r,
                                       sccd_aabb* out)
{
    if (!c) return SCCD_E_INVALID;
    return guarded(c, [&] {
        SCCD_REQUIRE(nV >= 0 && (nV == 0 || (V0 && V1 && out)), "build_vertex_boxes: bad arguments");
        if (nV == 0) return;
        const size_t nb = sizeof(double) * 3 * (size_t)nV;
        c->tmp0.ensure(nb);
        c->tmp1.ensure(nb);
        c->tmp2.ensure(sizeof(double) * 6 * (size_t)nV);
        c->np_scratch0.ensure(sizeof(sccd_aabb) * (size_t)nV);
        copy_in(c, c->tmp0.p, V0, nb, 0);
        copy_in(c, c->tmp1.p, V1, nb, 0);
        launch_pack_vertices(c, c->tmp0.as<double>(), c->tmp1.as<double>(), nV, c->tmp2.as<double>());
        launch_vertex_boxes(c, c->tmp2.as<double>(), nV, r, c->np_scratch0.as<sccd_aabb>());
        SCCD_HIP(hipMemcpyAsync(out, c->np_scratch0.p, sizeof(sccd_aabb) * (size_t)nV, hipMemcpyDeviceToHost,
                                c->stream));
        SCCD_HIP(hipStreamSynchronize(c->stream));
    });
}

static int build_elem_boxes(sccd_ctx* c, const sccd_aabb* vb, int nV, const int32_t* M, int nM, int cols,
                            sccd_aabb* out)
{
    if (!c) return SCCD_E_INVALID;
    return guarded(c, [&] {
        SCCD_REQUIRE(nV >= 0 && nM >= 0 && (nM == 0 || (vb && M && out)), "build_*_boxes: bad arguments");
        if (nM == 0) return;
        SCCD_REQUIRE(nV > 0, "build_*_boxes: vertex index out of range");
        c->tmp0.ensure(sizeof(sccd_aabb) * (size_t)std::max(nV, 1));
        c->tmp1.ensure(sizeof(int32_t) * (size_t)cols * (size_t)nM + 16); // (+ the verdict word of the index check, at the end)
        c->tmp2.ensure(sizeof(int4) * (size_t)nM);
        unsigned* const d_bad = reinterpret_cast<unsigned*>(c->tmp1.as<char>() + ((sizeof(int32_t) * (size_t)cols * (size_t)nM + 3) & ~(size_t)3));
        SCCD_HIP(hipMemsetAsync(d_bad, 0, sizeof(unsigned), c->stream));
        c->np_scratch0.ensure(sizeof(sccd_aabb) * (size_t)nM);
        copy_in(c, c->tmp0.p, vb, sizeof(sccd_aabb) * (size_t)nV, 0);
        copy_in(c, c->tmp1.p, M, sizeof(int32_t) * (size_t)cols * (size_t)nM, 0);
        if (cols == 2) {
            launch_pack_edges(c, c->tmp1.as<int32_t>(), nM, nV, c->tmp2.as<int2>(), d_bad);
            launch_edge_boxes(c, c->tmp0.as<sccd_aabb>(), c->tmp2.as<int2>(), nM, c->np_scratch0.as<sccd_aabb>());
        } else {
            launch_pack_faces(c, c->tmp1.as<int32_t>(), nM, nV, c->tmp2.as<int4>(), d_bad);
            launch_face_boxes(c, c->tmp0.as<sccd_aabb>(), c->tmp2.as<int4>(), nM, c->np_scratch0.as<sccd_aabb>());
        }
        // (indices are validated on the device while they are packed -- clamped, so the builders read nothing wild)
        unsigned bad = 0;
        SCCD_HIP(hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, c->stream));
        SCCD_HIP(hipStreamSynchronize(c->stream));
        SCCD_REQUIRE(bad == 0, "build_*_boxes: vertex index out of range");
        SCCD_HIP(hipMemcpyAsync(out, c->np_scratch0.p, sizeof(sccd_aabb) * (size_t)nM, hipMemcpyDeviceToHost,
                                c->stream));
        SCCD_HIP(hipStreamSynchronize(c->stream));
    });
}

extern "C" int sccd_build_edge_boxes(sccd_ctx* c, const sccd_aabb* vb, int nV, const int32_t* E, int nE,
                                     sccd_aabb* out)
{
    return build_elem_boxes(c, vb, nV, E, nE, 2, out);
}
extern "C" int sccd_build_face_boxes(sccd_ctx* c, const sccd_aabb* vb, int nV, const int32_t* F, int nF,
                                     sccd_aabb* out)
{
    return build_elem_boxes(c, vb, nV, F, nF, 3, out);
}

extern "C" int sccd_boxes_create(sccd_ctx* c, const sccd_aabb* boxes, int n, int src_on_device, sccd_boxes** out)
{
    if (!c || !out) return SCCD_E_INVALID;
    *out = nullptr;
    return guarded(c, [&] {
        SCCD_REQUIRE(n >= 0 && (n == 0 || boxes), "boxes_create: bad arguments");
        std::unique_ptr<sccd_boxes> b(new sccd_boxes());
        b->ctx = c;
        b->n = n;
        b->raw.ensure(sizeof(sccd_aabb) * (size_t)std::max(n, 1));
        copy_in(c, b->raw.p, boxes, sizeof(sccd_aabb) * (size_t)n, src_on_device);
        SCCD_HIP(hipStreamSynchronize(c->stream));
        *out = b.release();
    });
}

// vertex boxes -> (edge, face) boxes, all on the device (ccd.cu:112-121 without the host trip)
constexpr int LAZY_STATS_STRIDE = 8; // a lazy list's grid statistics look at every 8th element
// lazy_ef: the edge and face lists are only DESCRIBED (multi-GPU ccd(): a rank builds the boxes of its window of cells, in
// the fill pass -- internal.hpp sccd_boxes::lazy); their grid statistics come from a sample every rank takes alike
static void boxes_from_mesh(sccd_ctx* c, const sccd_mesh* m, double r, Pipeline* pl, bool want_v, bool want_e,
                            bool want_f, bool lazy_ef = false)
{
    (void)want_v;
    ProfScope ps(c, SCCD_PROF_BOXES);
    // the builders also produce the bounds / extent sums the grid needs (no second pass over the boxes)
    auto begin_stats = [&](sccd_boxes& b) {
        b.stats.ensure(SCCD_STATS_BYTES);
        b.have_stats = false;
    };
    pl->vb.n = m->nV;
    pl->vb.kind = BOX_VERTEX;
    pl->vb.raw.ensure(sizeof(sccd_aabb) * (size_t)std::max(m->nV, 1));
    begin_stats(pl->vb);
    pl->vb.n_part = launch_vertex_boxes(c, m->V.as<double>(), m->nV, r, pl->vb.raw.as<sccd_aabb>(), pl->vb.stats_head(),
                                        pl->vb.stats_part());
    pl->vb.have_stats = true;
    if (want_e) {
        pl->eb.n = m->nE;
        pl->eb.kind = BOX_EDGE;
        pl->eb.raw.ensure(sizeof(sccd_aabb) * (size_t)std::max(m->nE, 1));
        begin_stats(pl->eb);
    }
    if (want_f) {
        pl->fb.n = m->nF;
        pl->fb.kind = BOX_FACE;
        pl->fb.raw.ensure(sizeof(sccd_aabb) * (size_t)std::max(m->nF, 1));
        begin_stats(pl->fb);
    }
    pl->eb.lazy = pl->fb.lazy = false;
    if (lazy_ef) {
        for (sccd_boxes* b : { want_e ? &pl->eb : nullptr, want_f ? &pl->fb : nullptr }) {
            if (!b || b->n == 0) continue;
            b->lazy = true;
            b->lazy_vb = pl->vb.raw.as<sccd_aabb>();
            b->lazy_elems = b == &pl->eb ? (const void*)m->E.as<int2>() : (const void*)m->F.as<int4>();
            b->n_part = launch_elem_stats(c, b, LAZY_STATS_STRIDE, b->stats_head(), b->stats_part());
            b->have_stats = true;
        }
        return;
    }
    if (want_e && want_f && m->nE > 0 && m->nF > 0) { // both in one launch
        launch_edge_face_boxes(c, pl->vb.raw.as<sccd_aabb>(), m->E.as<int2>(), m->nE, pl->eb.raw.as<sccd_aabb>(), pl->eb.stats_head(),
                               pl->eb.stats_part(), &pl->eb.n_part, m->F.as<int4>(), m->nF, pl->fb.raw.as<sccd_aabb>(),
                               pl->fb.stats_head(), pl->fb.stats_part(), &pl->fb.n_part);
        pl->eb.have_stats = pl->fb.have_stats = true;
        return;
    }
    if (want_e) {
        pl->eb.n_part = launch_edge_boxes(c, pl->vb.raw.as<sccd_aabb>(), m->E.as<int2>(), m->nE,
                                          pl->eb.raw.as<sccd_aabb>(), pl->eb.stats_head(), pl->eb.stats_part());
        pl->eb.have_stats = true;
    }
    if (want_f) {
        pl->fb.n_part = launch_face_boxes(c, pl->vb.raw.as<sccd_aabb>(), m->F.as<int4>(), m->nF,
                                          pl->fb.raw.as<sccd_aabb>(), pl->fb.stats_head(), pl->fb.stats_part());
        pl->fb.have_stats = true;
    }
}

// statistics of a list that was uploaded rather than built here: one pass, cached in the object
static void ensure_stats(sccd_ctx* c, const sccd_boxes* b)
{
    if (b->have_stats) return;
    b->stats.ensure(SCCD_STATS_BYTES);
    b->n_part = launch_box_stats(c, b->raw.as<sccd_aabb>(), b->n, b->stats_head(), b->stats_part());
    b->have_stats = true;
}

static sccd_boxes* clone_boxes(sccd_ctx* c, const sccd_boxes& s)
{
    std::unique_ptr<sccd_boxes> b(new sccd_boxes());
    b->ctx = c;
    b->n = s.n;
    b->kind = s.kind; // (device-resident, opaque: the caller cannot have changed the ids)
    b->raw.ensure(sizeof(sccd_aabb) * (size_t)std::max(s.n, 1));
    copy_in(c, b->raw.p, s.raw.p, sizeof(sccd_aabb) * (size_t)s.n, 1);
    return b.release();
}

extern "C" int sccd_boxes_from_mesh(sccd_ctx* c, const sccd_mesh* m, double r, sccd_boxes** vb, sccd_boxes** eb,
                                    sccd_boxes** fb)
{
    if (!c || !m) return SCCD_E_INVALID;
    return guarded(c, [&] {
        Pipeline* pl = pipeline_of(c);
        boxes_from_mesh(c, m, r, pl, true, eb != nullptr, fb != nullptr);
        if (vb) *vb = clone_boxes(c, pl->vb);
        if (eb) *eb = clone_boxes(c, pl->eb);
        if (fb) *fb = clone_boxes(c, pl->fb);
        SCCD_HIP(hipStreamSynchronize(c->stream));
    });
}

extern "C" int sccd_boxes_size(const sccd_boxes* b) { return b ? b->n : 0; }

extern "C" int sccd_boxes_download(const sccd_boxes* b, sccd_aabb* out)
{
    if (!b || !out) return SCCD_E_INVALID;
    return guarded(b->ctx, [&] {
        if (b->n == 0) return;
        SCCD_HIP(hipMemcpyAsync(out, b->raw.p, sizeof(sccd_aabb) * (size_t)b->n, hipMemcpyDeviceToHost,
                                b->ctx->stream));
        SCCD_HIP(hipStreamSynchronize(b->ctx->stream));
    });
}

extern "C" void sccd_boxes_destroy(sccd_boxes* b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    delete b;
}

// ------------------------------------------------------------------------------------------
// broad phase
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

// cell size = SCCD_CELL_FACTOR x mean box extent per minor axis (grid_setup_k)
static double cell_factor()
{
    const char* e = std::getenv("SCCD_CELL_FACTOR");
    const double f = e ? std::atof(e) : 4.0;
    return f > 0 ? f : 1e300; // <= 0 switches the grid off (one cell)
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
// the sorted records of the lists of a build whose (key, index) pairs are sorted, each list in its own arrays
static void lists_records(sccd_ctx* c, const sccd_boxes* A, const sccd_boxes* B, const GridParams* gp, SortedList* LA,
                          SortedList* LB, const uint32_t* d_tot = nullptr, int expect_bits = 0)
{
    ProfScope ps(c, SCCD_PROF_BOXES);
    if (!B) {
        launch_entry_records(c, A->raw.as<sccd_aabb>(), LA->key.as<uint32_t>(), LA->idx.as<uint32_t>(), LA->m, gp, 0, nullptr,
                             0, false, false, LA, d_tot, expect_bits);
        return;
    }
    SCCD_REQUIRE(!d_tot, "broad phase: device-side counts serve the one-list and the merged two-list build");
    if (LA->m == 0 || LB->m == 0) return;
    launch_entry_records_two(c, A->raw.as<sccd_aabb>(), LA->key.as<uint32_t>(), LA->idx.as<uint32_t>(), LA->m,
                             B->raw.as<sccd_aabb>(), LB->key.as<uint32_t>(), LB->idx.as<uint32_t>(), LB->m, /*b_tagged=*/false, gp, LA, LB);
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
        launch_entry_records_two(c, A->raw.as<sccd_aabb>(), keys, idx, (int)ma, B->raw.as<sccd_aabb>(), keys + ma, idx + ma, (int)mb,
                                 /*b_tagged=*/true, gp, LA, LB, d_tot, key_bits, d_extq);
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

// the grid parameters and, right behind them, the two list totals of a build: ONE copy brings both back (two copies in a row
// cost a 12 us bubble between them)
struct GridReadBack {
    GridParams gp;
    uint32_t total[2];
    uint32_t place; // (device only: the shared placement cursor of a merged two-list fill)
    uint32_t ext_q; // (device only: list A's largest extent along the sort axis, quantised -- the one-class two-list sweep)
};
static bool one_class_env()
{
    static const bool on = !(std::getenv("SCCD_ONE_CLASS") && std::atoi(std::getenv("SCCD_ONE_CLASS")) == 0);
    return on;
}
static bool speculate_env()
{
    static const bool on = !(std::getenv("SCCD_SPECULATE") && std::atoi(std::getenv("SCCD_SPECULATE")) == 0);
    return on;
}
static bool over_budget(uint32_t t, int n) { return (int64_t)t > std::max<int64_t>(3 * (int64_t)n, (int64_t)n + 4096); }

static void bp_build(sccd_broad_phase* bp, const sccd_boxes* A, const sccd_boxes* B)
{
    sccd_ctx* c = bp->ctx;
    SCCD_REQUIRE(A != nullptr, "BroadPhase::build: boxes are null");
    bp->A = A;
    bp->B = B;
    bp->built = true;
    bp->cursor = 0;
    bp->n_overlaps = 0;
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
        // lazy lists live on the device-window path only
        static const bool dw_env = !(std::getenv("SCCD_DEVICE_WINDOW") && std::atoi(std::getenv("SCCD_DEVICE_WINDOW")) == 0);
        const bool scan_env = std::getenv("SCCD_BUILD") && std::string(std::getenv("SCCD_BUILD")) == "scan";
        // (one GPU: the one-pass append build computes a lazy list's boxes in the fill just as well -- SCCD_LAZY_ONE)
        static const bool lazy_one = std::getenv("SCCD_LAZY_ONE") && std::atoi(std::getenv("SCCD_LAZY_ONE")) != 0;
        const bool merged_off = std::getenv("SCCD_MERGED_SORT") && std::atoi(std::getenv("SCCD_MERGED_SORT")) == 0;
        if (!(((c->shard_count > 1 && dw_env) || (c->shard_count == 1 && lazy_one && !merged_off && c->max_overlap_cutoff == 0)) && !scan_env && c->sort_axis >= 0)) {
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
    const double cf = cell_factor();
    bp->cell_lo = 0;
    bp->cell_hi = 1 << 30;
    bp->row_shard = false;
    unsigned long long window_est = 0; // entries of this rank's cell window, estimated from the sampled histogram
    // Two lists are sorted in ONE go: the entries of list B carry a tag bit on top of the key, so the sorted array
    // is list A followed by list B (one histogram and one set of radix passes instead of two; SCCD_MERGED_SORT=0
    // sorts them apart).  Only the one-pass append build can do it.
    static const bool merged_env = !(std::getenv("SCCD_MERGED_SORT") && std::atoi(std::getenv("SCCD_MERGED_SORT")) == 0);
    const bool scan_build_env = std::getenv("SCCD_BUILD") && std::string(std::getenv("SCCD_BUILD")) == "scan";
    const bool want_merged = merged_env && B != nullptr && !scan_build_env;
    // ONE sweep class for vertices x faces: every pair is found from the FACE's row, whose window reaches back over the
    // vertices that start before it (entry_record_body) -- a vertex box is a point's path: tiny along the sort axis -- instead
    // of a second class with the vertices as rows: each list is read once, not twice (SCCD_ONE_CLASS=0: two classes)
    const bool one_class = one_class_env() && want_merged && A->kind == BOX_VERTEX && B->kind == BOX_FACE && c->sweep_algo != 1;
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
        static const bool device_window_env = !(std::getenv("SCCD_DEVICE_WINDOW") && std::atoi(std::getenv("SCCD_DEVICE_WINDOW")) == 0);
        const bool device_window = c->shard_count > 1 && shrink == 0 && !device_window_tried && device_window_env && !scan_build_env;
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
        // SCCD_BUILD=scan selects count -> device-wide prefix scan -> fill (entries in box order: a
        // reproducible entry order, 0.15 ms slower per step on the 1M-triangle cloth)
        const char* build_env = std::getenv("SCCD_BUILD");
        const bool scan_build = build_env && std::string(build_env) == "scan";
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
                        && c->sweep_algo != 1 && !(std::getenv("SCCD_SORT") && std::string(std::getenv("SCCD_SORT")) == "classic")) {
                        const uint32_t ba = gs.total[0] + std::max<uint32_t>(4096u, gs.total[0] / 32u);
                        const uint32_t bb = B ? gs.total[1] + std::max<uint32_t>(4096u, gs.total[1] / 32u) : 0u;
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
    const bool ok = built.gp.key_bits == gs.key_bits                  // the sort ran the right passes
        && ta > 0 && (!two || tb > 0)                                 // (an empty side ends a build early)
        && ta <= bp->spec_bound[0] && tb <= bp->spec_bound[1]         // records and sweep saw every entry
        && (unsigned long long)ta + tb <= bp->spec_sorted             // ... and so did the sort
        && std::max(ta, tb) <= bp->spec_cap                           // the fill dropped nothing
        && (bp->spec_window                                           // no coarser grid was due, nor a split by rows
                ? !(hwin.total_est > (unsigned long long)std::max<int64_t>(3 * n_total, n_total + 4096)) && hwin.n_cells >= 4 * bp->ctx->shard_count
                : !over_budget(ta, bp->A->n) && !(two && over_budget(tb, bp->B->n)));
    bp->speculative = false;
    (ok ? bp->ctx->spec_hits : bp->ctx->spec_misses) += 1;
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

static void bp_detect_partial(sccd_broad_phase* bp, int phase = 0)
{
    sccd_ctx* c = bp->ctx;
    if (!bp->built) throw SccdError { SCCD_E_NOT_BUILT, "Must initialize build broad phase before detecting overlaps!" };
    bp->n_overlaps = 0;
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
        SCCD_HIP(hipMemsetAsync(d_cnt, 0, sizeof(SweepCounters), c->stream)); // pairs and candidate tests of THIS attempt
        {
            ProfScope ps(c, SCCD_PROF_SWEEP);
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
        if (phase == 1) return;
    launched:
        SweepCounters h;
        GridReadBack built; // (speculative build: the grid and the entry counts it really had)
        ShardWindow hwin {}; // (... of a rank of a multi-GPU job: the cell window it was dealt on the device)
        {
            ReadBack rb(c);
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
                return;
            }
            chunk_hi = bp->total_rows; // (a speculative build is swept in one chunk: bp_build)
        }
        {
            unsigned long long cs = 0;
            for (int k = 0; k < 32; k++) cs += h.cand_parts[k];
            bp->candidates = bp->candidates_done + (int64_t)cs; // (a chunk swept again after an overflow counts once)
            static const bool diag = std::getenv("SCCD_SWEEP_DIAG") && std::atoi(std::getenv("SCCD_SWEEP_DIAG")) != 0;
            if (diag)
                std::fprintf(stderr, "[sweep] rows %lld pairs %llu tests %llu | filter blocks %llu groups %llu confirm rounds %llu segments staged %llu\n",
                             (long long)(chunk_hi - chunk_lo), (unsigned long long)h.n_pairs, cs, h.diag[0], h.diag[1], h.diag[2], h.diag[3]);
        }
        if ((int64_t)h.n_pairs <= bp->capacity) {
            bp->n_overlaps = (int64_t)h.n_pairs;
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

// ------------------------------------------------------------------------------------------
// narrow phase
static NarrowCounters* narrow_counters(sccd_ctx* c)
{
    return reinterpret_cast<NarrowCounters*>(c->scalars.as<char>() + 2048);
}

struct NarrowResult {
    unsigned long long n_checks;
};

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

// copy_out_collisions (narrow_phase.cu:84-103): the queries with toi < 1, appended as (aid, bid, toi).  The filter runs on
// the device (ballot + one atomic per wave); only the records that survive cross the bus.
__global__ void collisions_compact_k(const int2* __restrict__ pairs, const double* __restrict__ per_query, long long n,
                                     sccd_collision* __restrict__ out, long long* __restrict__ out_idx,
                                     unsigned long long* __restrict__ n_out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const double t = i < n ? per_query[i] : 2.0;
    const bool hit = t < 1;
    const unsigned long long mask = __ballot(hit);
    if (mask == 0) return;
    const int leader = (int)__builtin_ctzll(mask);
    unsigned long long base = 0;
    if (lane_id() == leader) base = atomicAdd(n_out, (unsigned long long)popc64(mask));
    base = __shfl(base, leader, 64);
    if (hit) {
        const int2 p = pairs[i];
        const unsigned long long at = base + (unsigned long long)mbcnt64(mask);
        out[at] = sccd_collision { p.x, p.y, t };
        out_idx[at] = i;
    }
}
static void copy_out_collisions(sccd_ctx* c, const int2* d_pairs, const double* d_pq, int64_t n, std::vector<sccd_collision>& acc)
{
    if (n <= 0) return;
    DevBuf out, idx, cnt;
    out.ensure(sizeof(sccd_collision) * (size_t)n);
    idx.ensure(sizeof(long long) * (size_t)n);
    cnt.ensure(sizeof(unsigned long long));
    SCCD_HIP(hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(collisions_compact_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_pairs, d_pq,
                       (long long)n, out.as<sccd_collision>(), idx.as<long long>(), cnt.as<unsigned long long>());
    SCCD_HIP(hipGetLastError());
    unsigned long long k = 0;
    SCCD_HIP(hipMemcpyAsync(&k, cnt.p, sizeof k, hipMemcpyDeviceToHost, c->stream));
    SCCD_HIP(hipStreamSynchronize(c->stream));
    if (k == 0) return;
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
        DevBuf pq;
        if (collisions && n > 0) {
            pq.ensure(sizeof(double) * (size_t)n);
            d_pq = pq.as<double>();
        }
        run_narrow(c, m, d_pairs, n, is_vf, max_iter, tol, ms, allow_zero_toi, toi, d_pq);
        if (collisions && n > 0) {
            std::vector<sccd_collision> acc;
            copy_out_collisions(c, d_pairs, d_pq, n, acc);
            *collisions = collisions_to_c(acc);
            if (n_collisions) *n_collisions = (int64_t)acc.size();
        }
    });
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
    if (built) {} // (ccd() had the lists built already, by the helper)
    else if (vf) bp_build(bp, &pl->vb, &pl->fb);
    else bp_build(bp, &pl->eb, nullptr);
    bool started = swept; // (... and the first sweep enqueued as well: bp_detect_partial(bp, 1))
    while (bp->cursor < bp->total_rows) {
        narrow_counters_upload(c, narrow_counters(c), *toi); // ahead of the sweep: one copy less between sweep and narrow phase
        bp_detect_partial(bp, started ? 2 : 0);
        started = false;
        if (before_narrow && *before_narrow) {
            (*before_narrow)();
            *before_narrow = nullptr; // once
        }
        const NarrowResult r = run_narrow(c, m, bp->overlaps.as<int2>(), bp->n_overlaps, vf ? 1 : 0, max_iter, tol, ms,
                                          allow_zero_toi, toi, nullptr);
        if (st) {
            (vf ? st->n_vf_pairs : st->n_ee_pairs) += bp->n_overlaps;
            (vf ? st->n_vf_checks : st->n_ee_checks) += (int64_t)r.n_checks;
        }
    }
    if (st) (vf ? st->n_vf_candidates : st->n_ee_candidates) = bp->candidates;
}

static void ccd_on_mesh(sccd_ctx* c, const sccd_mesh* m, double ms, int max_iter, double tol, int allow_zero_toi,
                        double* toi_out, sccd_stats* st)
{
    Pipeline* pl = pipeline_of(c);
    if (st) std::memset(st, 0, sizeof *st);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (st && c->profile == 1) {
        SCCD_HIP(hipEventCreate(&e0));
        SCCD_HIP(hipEventCreate(&e1));
        SCCD_HIP(hipEventRecord(e0, c->stream));
    }
    double before[SCCD_PROF_COUNT];
    if (st && c->profile == 1) {
        merge_side_profile(c);
        std::memcpy(before, c->prof_ms, sizeof before);
    }
    // (a rank of a multi-GPU job builds the edge and face boxes of its window of cells only: see boxes_from_mesh)
    static const bool lazy_env = !(std::getenv("SCCD_LAZY_BOXES") && std::atoi(std::getenv("SCCD_LAZY_BOXES")) == 0);
    static const bool lazy_one_env = std::getenv("SCCD_LAZY_ONE") && std::atoi(std::getenv("SCCD_LAZY_ONE")) != 0;
    const bool lazy_ef = lazy_env && (c->shard_count > 1 || lazy_one_env) && c->sort_axis >= 0 && c->max_overlap_cutoff == 0;
    boxes_from_mesh(c, m, ms, pl, true, true, true, lazy_ef); // inflation radius = min_distance (ccd.cu:112)
    double toi = 1; // ccd.cu:125
    // The edge-edge lists do not depend on the vertex-face pass: a helper context (own stream, scratch, pinned mirror)
    // builds them on a worker thread meanwhile.  Both builds are chains of short, latency-bound kernels with host
    // round trips in between, so two of them interleave almost for free (2.23 instead of 2.34 ms per step on the
    // 1M-triangle cloth; 1.99 instead of 2.11 with the round-2 kernels).  On by default since the whole GPU suite
    // and the soak run with it; SCCD_OVERLAP=0 keeps the passes apart.
    static const bool overlap_env = !(std::getenv("SCCD_OVERLAP") && std::atoi(std::getenv("SCCD_OVERLAP")) == 0);
    bool helper = false, presweep_done = false;
    static const bool presweep_env = !(std::getenv("SCCD_PRESWEEP") && std::atoi(std::getenv("SCCD_PRESWEEP")) == 0);
    if (overlap_env && !c->passes_apart && m->nE > 0) {
        if (!c->side) {
            if (sccd_create(c->device, &c->side) != SCCD_OK) throw SccdError { SCCD_E_NOMEM, "ccd: cannot create the helper context" };
            SCCD_HIP(hipEventCreateWithFlags(&c->side_event, hipEventDisableTiming));
            pl->bp_ee.ctx = c->side;
            // SCCD_HELPER_PRIORITY=low|high: the helper's stream with the lowest / highest queue priority (measurements)
            if (const char* pe = std::getenv("SCCD_HELPER_PRIORITY")) {
                int least = 0, greatest = 0;
                SCCD_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
                hipStream_t s = nullptr;
                SCCD_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, std::string(pe) == "high" ? greatest : least));
                SCCD_HIP(hipStreamDestroy(c->side->stream));
                c->side->stream = s;
            }
        }
        sccd_ctx* const sc = c->side;
        sc->sort_axis = c->sort_axis;
        sc->sweep_algo = c->sweep_algo;
        sc->shard_rank = c->shard_rank;
        sc->shard_count = c->shard_count;
        sc->overlap_capacity = c->overlap_capacity;
        sc->max_overlap_cutoff = c->max_overlap_cutoff;
        sc->memory_limit_mb = c->memory_limit_mb;
        sc->profile = c->profile;
        SCCD_HIP(hipEventRecord(c->side_event, c->stream)); // the boxes are complete behind this point
        const int device = c->device;
        hipEvent_t const ev = c->side_event;
        sccd_broad_phase* const bp_ee = &pl->bp_ee;
        const sccd_boxes* const eb = &pl->eb;
        pl->worker.submit([=] {
            SCCD_HIP(hipSetDevice(device));
            SCCD_HIP(hipStreamWaitEvent(sc->stream, ev, 0));
            bp_build(bp_ee, eb, nullptr);
        });
        helper = true;
    }
    // The edge-edge SWEEP runs beside the vertex-face NARROW phase: it is enqueued on the helper's stream right before
    // that kernel is launched, with half a CU's worth of blocks (they are resident first, the narrow kernel's blocks
    // take the rest and, being ticket-driven, make do with what they get).  The sweep waits on dependent gathers most
    // of the time, the narrow phase is bound by vector issue: sharing the CUs, the two take little longer than the
    // narrow phase alone.  SCCD_PRESWEEP=0 keeps them apart.
    std::function<void()> start_ee_sweep;
    if (helper && presweep_env)
        start_ee_sweep = [&] {
            pl->worker.wait(); // the lists are built (long since: the build is shorter than the vertex-face broad phase)
            // ... behind whatever this context's stream holds now (the vertex-face sweep): ordered on the DEVICE, so the
            // edge-edge sweep starts the moment that sweep ends -- not a host round trip later
            SCCD_HIP(hipEventRecord(c->side_event, c->stream));
            SCCD_HIP(hipStreamWaitEvent(c->side->stream, c->side_event, 0));
            static const int side_blocks = std::getenv("SCCD_PRESWEEP_BLOCKS") ? std::max(1, std::atoi(std::getenv("SCCD_PRESWEEP_BLOCKS"))) : 2;
            c->side->sweep_blocks_per_cu = side_blocks;
            try {
                bp_detect_partial(&pl->bp_ee, 1);
            } catch (...) {
                c->side->sweep_blocks_per_cu = 0;
                throw;
            }
            c->side->sweep_blocks_per_cu = 0;
            presweep_done = true;
        };
    // ... and the edge-edge NARROW kernel starts on the helper's stream the moment that sweep is done, beside the tail of
    // the vertex-face kernel (a wave-step of a deep query is a long dependent chain: the last part of a narrow launch
    // keeps few lanes busy).  The two kernels share ONE running TOI (the vertex-face launch's word), so each prunes with
    // what the other finds -- the final minimum does not depend on the order (Appendix A.20).  Only when both passes are
    // served by the walk kernel in one chunk each; SCCD_NARROW_BESIDE=0 turns it off.
    static const bool beside_env = !(std::getenv("SCCD_NARROW_BESIDE") && std::atoi(std::getenv("SCCD_NARROW_BESIDE")) == 0);
    bool both_done = false;
    try {
        // (a check limit: each pass proves on its own that the limit did not matter -- narrow.hip, the certificate -- which needs
        // the pass's own running TOI: the passes stay in sequence)
        if (helper && presweep_env && beside_env && c->max_overlap_cutoff == 0 && max_iter < 0) {
            sccd_ctx* const sc = c->side;
            sc->arith = c->arith;
            sc->scalar_f32 = c->scalar_f32;
            sc->narrow_algo = c->narrow_algo;
            sc->limit_level_order = c->limit_level_order;
            bp_build(&pl->bp, &pl->vb, &pl->fb);
            narrow_counters_upload(c, narrow_counters(c), toi);
            // the vertex-face sweep is enqueued, the edge-edge sweep behind it (on the helper's stream, by an event), and only
            // then does the host wait for the vertex-face pairs: the narrow kernel it launches next finds the edge-edge
            // sweep's blocks resident already and takes the rest of the chip
            bp_detect_partial(&pl->bp, 1);
            start_ee_sweep();
            start_ee_sweep = nullptr;
            bp_detect_partial(&pl->bp, 2);
            const bool vf_one_chunk = pl->bp.cursor >= pl->bp.total_rows;
            const NarrowParams pv = narrow_params(c, m, pl->bp.overlaps.as<int2>(), pl->bp.n_overlaps, 1, max_iter, tol, ms, allow_zero_toi);
            if (vf_one_chunk && narrow_uses_walk_kernel(c, pv, false)) {
                double toi_vf = toi, toi_ee = toi;
                narrow_phase_begin(c, pv, narrow_counters(c), &toi_vf, nullptr); // (asynchronous)
                bp_detect_partial(&pl->bp_ee, 2);                                 // waits for the edge-edge sweep
                if (pl->bp_ee.cursor >= pl->bp_ee.total_rows) {
                    NarrowParams pe = narrow_params(sc, m, pl->bp_ee.overlaps.as<int2>(), pl->bp_ee.n_overlaps, 0, max_iter, tol, ms, allow_zero_toi);
                    pe.toi_word = &narrow_counters(c)->toi_bits;
                    narrow_counters_upload(sc, narrow_counters(sc), toi_ee);
                    c->np_peer_stream = sc->stream;
                    try {
                        narrow_phase_begin(sc, pe, narrow_counters(sc), &toi_ee, nullptr);
                        narrow_phase_end(c, pv, narrow_counters(c), &toi_vf, nullptr);
                    } catch (...) {
                        c->np_peer_stream = nullptr;
                        throw;
                    }
                    c->np_peer_stream = nullptr;
                    const NarrowResult rv = narrow_result(c);
                    narrow_phase_end(sc, pe, narrow_counters(sc), &toi_ee, nullptr);
                    const NarrowResult re = narrow_result(sc);
                    toi = std::min(toi_vf, toi_ee);
                    if (st) {
                        st->n_vf_pairs += pl->bp.n_overlaps;
                        st->n_vf_checks += (int64_t)rv.n_checks;
                        st->n_vf_candidates = pl->bp.candidates;
                        st->n_ee_pairs += pl->bp_ee.n_overlaps;
                        st->n_ee_checks += (int64_t)re.n_checks;
                        st->n_ee_candidates = pl->bp_ee.candidates;
                    }
                    both_done = true;
                } else { // (the edge-edge overlaps come in chunks: finish the vertex-face pass, then chunk by chunk as usual)
                    narrow_phase_end(c, pv, narrow_counters(c), &toi_vf, nullptr);
                    const NarrowResult rv = narrow_result(c);
                    toi = toi_vf;
                    if (st) {
                        st->n_vf_pairs += pl->bp.n_overlaps;
                        st->n_vf_checks += (int64_t)rv.n_checks;
                        st->n_vf_candidates = pl->bp.candidates;
                    }
                    // the first edge-edge chunk is swept already: its narrow phase, then the rest of the loop
                    const NarrowResult re = run_narrow(c, m, pl->bp_ee.overlaps.as<int2>(), pl->bp_ee.n_overlaps, 0, max_iter, tol, ms,
                                                       allow_zero_toi, &toi, nullptr);
                    if (st) {
                        st->n_ee_pairs += pl->bp_ee.n_overlaps;
                        st->n_ee_checks += (int64_t)re.n_checks;
                    }
                    presweep_done = false; // (consumed)
                }
            } else { // not this time: the vertex-face pass as usual (its lists are built and its first chunk swept)
                const NarrowResult rv = run_narrow(c, m, pl->bp.overlaps.as<int2>(), pl->bp.n_overlaps, 1, max_iter, tol, ms,
                                                   allow_zero_toi, &toi, nullptr);
                if (st) {
                    st->n_vf_pairs += pl->bp.n_overlaps;
                    st->n_vf_checks += (int64_t)rv.n_checks;
                }
                ccd_pass(c, m, pl, &pl->bp, true, ms, max_iter, tol, allow_zero_toi, &toi, st, /*built=*/true);
            }
        } else {
            ccd_pass(c, m, pl, &pl->bp, true, ms, max_iter, tol, allow_zero_toi, &toi, st, false, false, &start_ee_sweep);
        }
    } catch (...) {
        if (helper) {
            try {
                pl->worker.wait();
            } catch (...) {
            }
            (void)hipStreamSynchronize(c->side->stream);
        }
        throw;
    }
    if (both_done) {
        // (both passes are behind us)
    } else if (helper) {
        pl->worker.wait();
        ccd_pass(c, m, pl, &pl->bp_ee, false, ms, max_iter, tol, allow_zero_toi, &toi, st, /*built=*/true, /*swept=*/presweep_done);
    } else {
        ccd_pass(c, m, pl, &pl->bp, false, ms, max_iter, tol, allow_zero_toi, &toi, st);
    }
    *toi_out = toi;
    if (st && c->profile == 1) {
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
        st->ms_sweep = (c->prof_ms[SCCD_PROF_SWEEP] - before[SCCD_PROF_SWEEP])
            + (c->prof_ms[SCCD_PROF_RANGES] - before[SCCD_PROF_RANGES]);
        st->ms_narrow = (c->prof_ms[SCCD_PROF_NARROW_VF] - before[SCCD_PROF_NARROW_VF])
            + (c->prof_ms[SCCD_PROF_NARROW_EE] - before[SCCD_PROF_NARROW_EE]);
    }
}

extern "C" int sccd_ccd_mesh(sccd_ctx* c, const sccd_mesh* m, double ms, int max_iter, double tol, int allow_zero_toi,
                             double* toi, sccd_stats* stats)
{
    if (!c || !m || !toi) return SCCD_E_INVALID;
    return guarded(c, [&] {
        SCCD_REQUIRE(m->ctx == c, "ccd: mesh belongs to another context");
        ccd_on_mesh(c, m, ms, max_iter, tol, allow_zero_toi, toi, stats);
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
        ccd_on_mesh(c, m, ms, max_iter, tol, allow_zero_toi, &t, stats);
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
    if (vf) bp_build(&pl->bp, &pl->vb, &pl->fb);
    else bp_build(&pl->bp, &pl->eb, nullptr);
    DevBuf pq;
    while (pl->bp.cursor < pl->bp.total_rows) {
        bp_detect_partial(&pl->bp);
        const int64_t n = pl->bp.n_overlaps;
        if (n > 0) pq.ensure(sizeof(double) * (size_t)n);
        run_narrow(c, m, pl->bp.overlaps.as<int2>(), n, vf ? 1 : 0, max_iter, tol, ms, allow_zero_toi, toi,
                   n > 0 ? pq.as<double>() : nullptr);
        copy_out_collisions(c, pl->bp.overlaps.as<int2>(), pq.as<double>(), n, acc);
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

extern "C" int sccd_dev_alloc(sccd_ctx* c, size_t bytes, void** d_ptr)
{
    if (!c || !d_ptr) return SCCD_E_INVALID;
    *d_ptr = nullptr;
    return guarded(c, [&] {
        if (bytes == 0) return;
        SCCD_HIP(hipSetDevice(c->device));
        if (hipMalloc(d_ptr, bytes) != hipSuccess) {
            (void)hipGetLastError();
            throw SccdError { SCCD_E_NOMEM, "device allocation failed" };
        }
    });
}
extern "C" int sccd_dev_free(sccd_ctx* c, void* d_ptr)
{
    if (!c) return SCCD_E_INVALID;
    return guarded(c, [&] {
        if (d_ptr) SCCD_HIP(hipFree(d_ptr));
    });
}
extern "C" int sccd_dev_upload(sccd_ctx* c, void* d_dst, const void* h_src, size_t bytes)
{
    if (!c || (bytes && (!d_dst || !h_src))) return SCCD_E_INVALID;
    return guarded(c, [&] {
        if (bytes == 0) return;
        SCCD_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
        SCCD_HIP(hipStreamSynchronize(c->stream));
    });
}
extern "C" int sccd_dev_copy(sccd_ctx* c, void* d_dst, const void* d_src, size_t bytes)
{
    if (!c || (bytes && (!d_dst || !d_src))) return SCCD_E_INVALID;
    return guarded(c, [&] {
        if (bytes == 0) return;
        SCCD_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, c->stream));
        SCCD_HIP(hipStreamSynchronize(c->stream));
    });
}
extern "C" int sccd_dev_download(sccd_ctx* c, void* h_dst, const void* d_src, size_t bytes)
{
    if (!c || (bytes && (!h_dst || !d_src))) return SCCD_E_INVALID;
    return guarded(c, [&] {
        if (bytes == 0) return;
        SCCD_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
        SCCD_HIP(hipStreamSynchronize(c->stream));
    });
}

// partial_ipc_ccd_strategy<run_vf> (ipc_ccd_strategy.cu:12-92)
static void ipc_pass(sccd_ctx* c, const sccd_mesh* m, Pipeline* pl, bool vf, double ms, int max_iter, double tol,
                     double* earliest)
{
    if (vf) bp_build(&pl->bp, &pl->vb, &pl->fb);
    else bp_build(&pl->bp, &pl->eb, nullptr);
    while (pl->bp.cursor < pl->bp.total_rows) {
        bp_detect_partial(&pl->bp);
        const double before = *earliest;
        run_narrow(c, m, pl->bp.overlaps.as<int2>(), pl->bp.n_overlaps, vf ? 1 : 0, max_iter, tol, ms,
                   /*allow_zero_toi=*/1, earliest, nullptr);
        if (*earliest < 1e-6) { // :72-91: conservative re-run without minimum separation
            *earliest = before;
            run_narrow(c, m, pl->bp.overlaps.as<int2>(), pl->bp.n_overlaps, vf ? 1 : 0, /*max_iter=*/-1, tol,
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

extern "C" int sccd_selftest_lds_gather(sccd_ctx* c, int n_waves, int n_active, int64_t* n_bad)
{
    if (!c || !n_bad) return SCCD_E_INVALID;
    *n_bad = -1;
    return guarded(c, [&] {
        SCCD_REQUIRE(n_waves >= 1 && n_waves <= 4096 && n_active >= 0 && n_active <= 64, "selftest: bad arguments");
        const int nrec = 5000, per_wave = 2 * 3 * 64 * 2 + 64 * 2;
        {
            const int wpb = narrow_selftest_waves_per_block(); // whole blocks of the kernel's own shape
            n_waves = (n_waves + wpb - 1) / wpb * wpb;
        }
        std::vector<double> hV((size_t)nrec * 6);
        for (size_t i = 0; i < hV.size(); i++) hV[i] = (double)i + 0.5;
        std::vector<int> perm((size_t)n_waves * 64);
        for (size_t i = 0; i < perm.size(); i++) perm[i] = (int)((i * 2654435761ull + 11) % nrec);
        DevBuf dV, dP, dO;
        dV.ensure(hV.size() * 8);
        dP.ensure(perm.size() * 4);
        dO.ensure((size_t)n_waves * per_wave * 8);
        copy_in(c, dV.p, hV.data(), hV.size() * 8, 0);
        copy_in(c, dP.p, perm.data(), perm.size() * 4, 0);
        narrow_selftest_lds_gather(c, dV.as<double>(), dP.as<int>(), n_waves, n_active, dO.as<double>());
        std::vector<double> o((size_t)n_waves * per_wave);
        SCCD_HIP(hipMemcpy(o.data(), dO.p, o.size() * 8, hipMemcpyDeviceToHost));
        int64_t bad = 0;
        for (int w = 0; w < n_waves; w++) {
            const double* ow = o.data() + (size_t)w * per_wave;
            for (int p = 0; p < 3; p++)
                for (int l = 0; l < 64; l++) {
                    const double base = (double)perm[(size_t)w * 64 + l] * 6 + p * 2 + 0.5;
                    const double w0 = l < n_active ? base : -1.0, w1 = l < n_active ? base + 1 : -1.0;
                    bad += ow[2 * (p * 64 + l)] != w0;     // B1: the gathered pieces
                    bad += ow[2 * (p * 64 + l) + 1] != w1;
                    const int i = p * 64 + l;              // B0: what the plain LDS traffic left
                    bad += ow[2 * 3 * 64 + 2 * i] != (i < 64 ? 8.0 : 0.0);
                    bad += ow[2 * 3 * 64 + 2 * i + 1] != (double)i;
                }
            for (int l = 0; l < 64; l++) {
                double acc = 0;
                for (int r = 0; r < 8; r++) acc += (double)((l + 7 * r) % (3 * 64));
                bad += ow[4 * 3 * 64 + 2 * l] != (double)(l + 1000 * w);
                bad += ow[4 * 3 * 64 + 2 * l + 1] != acc;
            }
        }
        *n_bad = bad;
    });
}

extern "C" int sccd_sort_pairs_u32(sccd_ctx* c, uint32_t* d_keys, uint32_t* d_vals, int64_t n)
{
    if (!c) return SCCD_E_INVALID;
    return guarded(c, [&] {
        ProfScope ps(c, SCCD_PROF_SORT);
        radix_sort_pairs_u32(c, d_keys, d_vals, n, 32);
    });
}
