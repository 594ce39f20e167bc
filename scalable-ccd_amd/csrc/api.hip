// api.hip -- the C ABI of include/sccd.h, part 1: context, options, profiling, HBM-resident meshes and box lists, device
// memory helpers, self-tests.  (build.hip: BroadPhase; drivers.hip: narrow_phase, ccd(), ipc_ccd_strategy.)  Host code only;
// kernels live in boxes / scan / sort / sweep / narrow .hip.
#include "api_internal.hpp"

// ------------------------------------------------------------------------------------------
thread_local std::string g_create_error;

Pipeline* pipeline_of(sccd_ctx* c)
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
void merge_side_profile(sccd_ctx* c)
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

// ReadBack's gather kernel (common.hpp): four waves copy every item into the pinned mailbox, then the sequence word.  Eight
// vector registers at most (tests/test_kernel_resources.py): what a SIMD running three narrow-phase waves has left, so the
// block is placed beside them instead of waiting for one to retire.  (Four waves: the counters of a narrow launch are 2.6 KB,
// a dependent load -> store round per 256 bytes with one.)
__global__ __launch_bounds__(256) void readback_gather_k(ReadBackItems it, unsigned* __restrict__ box, unsigned long long* seq_word,
                                                        unsigned long long seq)
{
    const unsigned lane = threadIdx.x;
    for (int i = 0; i < it.n; i++) {
        const unsigned* __restrict__ src = it.src[i];
        unsigned* dst = box + it.off_words[i];
        const unsigned nw = it.n_words[i];
        for (unsigned w = lane; w < nw; w += 256) __hip_atomic_store(dst + w, __builtin_nontemporal_load(src + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // Every lane's stores have been acknowledged before the word the host polls is written.  NOT a system-scope fence: that
    // writes back and invalidates the whole L2 (buffer_wbl2 / buffer_inv sc0 sc1) for stores that never were in it -- the
    // mailbox is uncached on the device -- and the kernels behind this one found their vertices and pairs gone from the
    // cache (first version: the 1M-triangle step 1.17 -> 1.26 ms).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (round 5, ADVICE r04: the word itself is a RELEASE store at system scope -- buffer_wbl2 sc0 sc1 in front of it, the write-back
    // half of a fence without the invalidate that hurt: 0.8446 against 0.8458 ms per step over three A/B rounds, eight registers still)
    // (... and, in front of it, the device's real-time clock: the END of a ccd() step for the host's account of where a slow step's
    // time went -- sccd_ctx::step_stamp, SCCD_OPT_DEVICE_SPAN_NS)
    if (lane == 0) {
        __hip_atomic_store(seq_word + 9, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(seq_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
void readback_gather_launch(sccd_ctx* c, const ReadBackItems& it, unsigned long long seq)
{
    hipLaunchKernelGGL(readback_gather_k, dim3(1), dim3(256), 0, c->stream, it, reinterpret_cast<unsigned*>(c->mailbox_dev),
                       reinterpret_cast<unsigned long long*>(c->mailbox_dev + SCCD_MAILBOX_BYTES), seq);
    SCCD_HIP(hipGetLastError());
}

extern "C" {

const char* sccd_version(void) { return "sccd-hip 0.4 (gfx950)"; } // 0.4: sccd_abi_*, SCCD_OPT_DEVICE_SPAN_NS / _HOST_WAITS, the cull under check limits; 0.3: sccd_stats grew (n_*_culled), SCCD_OPT_CULL; 0.2: default contract fused, option id 12 retired
size_t sccd_abi_sizeof_stats(void) { return sizeof(sccd_stats); }
int sccd_abi_prof_count(void) { return SCCD_PROF_COUNT; }

int sccd_create(int device, sccd_ctx** out)
{
    if (!out) return SCCD_E_INVALID;
    *out = nullptr;
    (void)lab_env(); // the laboratory switches are read here, once (common.hpp)
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
        int khz = 0; // (the clock behind s_memrealtime: sccd_ctx::device_span_ns)
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) c->wall_clock_khz = khz;
        else (void)hipGetLastError();
        SCCD_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
        c->scalars.ensure(4096);
        SCCD_HIP(hipMemsetAsync(c->scalars.p, 0, 4096, c->stream)); // (holds a counter that is never reset: sort.hip)
        c->h_scalars.ensure(16384);
        // the read-back mailbox: host-coherent (fine-grained) pinned memory, written by readback_gather_k, polled by ReadBack
        c->mailbox.ensure(SCCD_MAILBOX_BYTES + 256, hipHostMallocCoherent);
        std::memset(c->mailbox.p, 0, SCCD_MAILBOX_BYTES + 256);
        void* dev = nullptr;
        SCCD_HIP(hipHostGetDevicePointer(&dev, c->mailbox.p, 0));
        c->mailbox_dev = static_cast<char*>(dev);
        c->verdict.ensure(4096, hipHostMallocCoherent); // (the counters of a narrow launch, 2 KB, and the sequence word behind them at 2048)
        std::memset(c->verdict.p, 0, 4096);
        SCCD_HIP(hipHostGetDevicePointer(&dev, c->verdict.p, 0));
        c->verdict_dev = static_cast<char*>(dev);
    });
    if (rc != SCCD_OK) {
        g_create_error = c->err;
        delete c;
        return rc;
    }
    __atomic_add_fetch(&live_context_count(), 1, __ATOMIC_RELAXED);
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
    if (c->side_event) (void)hipEventDestroy(c->side_event);
    if (c->side_event2) (void)hipEventDestroy(c->side_event2);
    if (c->side_event3) (void)hipEventDestroy(c->side_event3);
    if (c->side_event4) (void)hipEventDestroy(c->side_event4);
    if (c->records_gate.ev) (void)hipEventDestroy(c->records_gate.ev);
    if (c->side) sccd_destroy(c->side);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    __atomic_sub_fetch(&live_context_count(), 1, __ATOMIC_RELAXED);
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
    case SCCD_OPT_CELL_FACTOR_MILLI: c->cell_factor_milli = (int)v; break;
    case SCCD_OPT_BUILD_SCAN: c->build_scan = v ? 1 : 0; break;
    case SCCD_OPT_CULL: c->cull_on = v < 0 ? 0 : (v > 2 ? 2 : (int)v); break; // 0 off, 1 where it pays (by mesh size), 2 always
    case SCCD_OPT_TWO_HALVES: c->two_halves = v < 0 ? 0 : (v > 2 ? 2 : (int)v); break; // 0 off, 1 where it pays (mesh size, history), 2 always
    case SCCD_OPT_TOI_GUESS:
        c->toi_guess_on = v ? 1 : 0;
        c->toi_guess = 1.0; // (forget what was learnt)
        c->toi_last = -1.0;
        break;
    case SCCD_OPT_TOI_GUESS_HITS:
    case SCCD_OPT_TOI_GUESS_MISSES: c->toi_guess_hits = c->toi_guess_misses = 0; break;
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
    case SCCD_OPT_CELL_FACTOR_MILLI: return c->cell_factor_milli;
    case SCCD_OPT_BUILD_SCAN: return c->build_scan;
    case SCCD_OPT_CULL: return c->cull_on;
    case SCCD_OPT_TWO_HALVES: return c->two_halves;
    case SCCD_OPT_ALLOC_COUNT: return devbuf_alloc_count();
    case SCCD_OPT_TOI_GUESS: return c->toi_guess_on;
    case SCCD_OPT_TOI_GUESS_HITS: return c->toi_guess_hits;
    case SCCD_OPT_TOI_GUESS_MISSES: return c->toi_guess_misses;
    case SCCD_OPT_SPEC_HITS: return c->spec_hits + (c->side ? c->side->spec_hits : 0);
    case SCCD_OPT_SPEC_MISSES: return c->spec_misses + (c->side ? c->side->spec_misses : 0);
    case SCCD_OPT_DEVICE_SPAN_NS: return c->device_span_ns;
    case SCCD_OPT_HOST_WAITS: return c->host_waits + (c->side ? c->side->host_waits : 0);
    case SCCD_OPT_READ_BACKS: return c->read_backs + (c->side ? c->side->read_backs : 0);
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
void copy_in(sccd_ctx* c, void* dst, const void* src, size_t bytes, int src_on_device)
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
void mesh_deferred_verdict(sccd_ctx* c)
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
sccd_mesh* scratch_mesh_from_host(sccd_ctx* c, const double* V0, const double* V1, int nV, const int32_t* E, int nE,
                                         const int32_t* F, int nF, bool defer_verdict)
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
// after_vertices (the step's two streams): recorded behind the vertex boxes; the EDGE boxes are then left to the caller's helper
// stream (edge_boxes_on) -- the face boxes here and the edge boxes there run side by side, and each build chain starts when
// its own list is done, not when both are
void boxes_from_mesh(sccd_ctx* c, const sccd_mesh* m, double r, Pipeline* pl, bool want_v, bool want_e,
                            bool want_f, bool lazy_ef, hipEvent_t after_vertices)
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
    if (after_vertices) SCCD_HIP(hipEventRecord(after_vertices, c->stream));
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
        const bool both = want_e && want_f && pl->eb.n > 0 && pl->fb.n > 0;
        for (sccd_boxes* b : { want_e ? &pl->eb : nullptr, want_f ? &pl->fb : nullptr }) {
            if (!b || b->n == 0) continue;
            b->lazy = true;
            b->lazy_vb = pl->vb.raw.as<sccd_aabb>();
            b->lazy_elems = b == &pl->eb ? (const void*)m->E.as<int2>() : (const void*)m->F.as<int4>();
            if (!both) b->n_part = launch_elem_stats(c, b, LAZY_STATS_STRIDE, b->stats_head(), b->stats_part());
            b->have_stats = true;
        }
        if (both) launch_elem_stats_two(c, &pl->eb, &pl->fb, LAZY_STATS_STRIDE, &pl->eb.n_part, &pl->fb.n_part); // (one launch: the same partials)
        return;
    }
    if (after_vertices) want_e = false; // (edge_boxes_on: the buffers are ready, the launch is the helper's)
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

void edge_boxes_on(sccd_ctx* sc, const sccd_mesh* m, Pipeline* pl)
{
    ProfScope ps(sc, SCCD_PROF_BOXES);
    pl->eb.n_part = launch_edge_boxes(sc, pl->vb.raw.as<sccd_aabb>(), m->E.as<int2>(), m->nE, pl->eb.raw.as<sccd_aabb>(),
                                      pl->eb.stats_head(), pl->eb.stats_part());
    pl->eb.have_stats = true;
}

// statistics of a list that was uploaded rather than built here: one pass, cached in the object
void ensure_stats(sccd_ctx* c, const sccd_boxes* b)
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
