// common.hpp -- context, device buffers, error plumbing and wave64 helpers shared by every
// translation unit of libsccd_hip.so.  gfx950 only: wavefront = 64 lanes everywhere.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/sccd.h"

#define SCCD_WAVE 64

// ------------------------------------------------------------------------------------------
// error plumbing: HIP failures become SCCD_E_HIP / SCCD_E_NOMEM with the text kept in the
// context (the reference throws std::runtime_error from gpuErrchk, cuda/utils/assert.cuh:18-27)
struct SccdError {
    int code;
    std::string msg;
};

#define SCCD_HIP(expr)                                                                         \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            (void)hipGetLastError(); /* the runtime's last-error is sticky: do not leave it to the next call */ \
            throw SccdError { _e == hipErrorOutOfMemory ? SCCD_E_NOMEM : SCCD_E_HIP,           \
                              std::string(#expr) + ": " + hipGetErrorString(_e) + " ("         \
                                  + __FILE__ + ":" + std::to_string(__LINE__) + ")" };         \
        }                                                                                      \
    } while (0)

#define SCCD_REQUIRE(cond, text)                                                               \
    do {                                                                                       \
        if (!(cond)) throw SccdError { SCCD_E_INVALID, std::string(text) };                    \
    } while (0)

// ------------------------------------------------------------------------------------------
// device allocations made by DevBuf::ensure since the library was loaded (SCCD_OPT_ALLOC_COUNT: a step that allocates is a slow
// step -- hipFree + hipMalloc are milliseconds -- and a caller, or bench.py --jitter, can tell such steps from the others)
inline long long& devbuf_alloc_count()
{
    static long long n = 0; // (diagnostic: plain increments, contexts on several threads may lose a count)
    return n;
}
// live contexts of this process (sccd_create / sccd_destroy; a context's helper counts).  The sort's block-index tiles
// (sort.hip os_pass_k) are argued for TWO concurrent sorts -- a ccd() step's two streams; with more contexts than one call's pair
// alive, the passes go back to atomic tickets, which need no argument about dispatch order.
inline int& live_context_count()
{
    static int n = 0; // (guarded by the callers' own serialisation of create / destroy per thread; a stale read only picks the safe path later)
    return n;
}
// grow-only device buffer (no hipMalloc in the steady state of repeated ccd() calls)
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    // contents are NOT preserved on growth
    void ensure(size_t bytes)
    {
        if (bytes <= cap) return;
        release();
        size_t want = bytes + bytes / 8 + 256;
        devbuf_alloc_count() += 1;
        SCCD_HIP(hipMalloc(&p, want));
        cap = want;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct PinnedBuf {
    void* p = nullptr;
    size_t cap = 0;
    ~PinnedBuf()
    {
        if (p) (void)hipHostFree(p);
    }
    void ensure(size_t bytes, unsigned flags = hipHostMallocDefault)
    {
        if (bytes <= cap) return;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        SCCD_HIP(hipHostMalloc(&p, bytes, flags));
        cap = bytes;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// ------------------------------------------------------------------------------------------
// ccd()'s two build chains (drivers.hip): a point of the caller's chain (the end of its records kernel) that a kernel of the helper's
// chain is ordered behind, on the device.  One thread enqueues both chains, the caller's first: `recorded` says the event is in.
struct StageGate {
    hipEvent_t ev = nullptr;
    bool recorded = false;
};
// The laboratory switches that are left (tools/README.md), read ONCE, when the first context is made: diagnostics and the handful a
// test or a soak uses.  Everything else that used to be an environment variable is either gone with the path it selected (round 6:
// SCCD_OVERLAP, SCCD_PRESWEEP, SCCD_NARROW_BESIDE, SCCD_NARROW_ORDER, SCCD_EE_EARLY, SCCD_EARLY_VERDICT, SCCD_SPLIT_BOXES,
// SCCD_EREC_LATE, SCCD_SYNC, SCCD_READBACK -- the step is enqueued by one thread and read back once) or an option of the context
// (SCCD_OPT_PASSES_APART, SCCD_OPT_CELL_FACTOR_MILLI, SCCD_OPT_BUILD_SCAN).
struct LabEnv {
    int np_diag = 0, sweep_diag = 0;   // SCCD_NP_DIAG / SCCD_SWEEP_DIAG: counters and cycle stamps of the narrow / sweep kernels
    bool speculate = true;             // SCCD_SPECULATE=0: every build waits for its entry counts (no speculative build), every ccd() starts from 1
    bool sort_tickets = false;         // SCCD_SORT_TICKETS=1: sort tiles by atomic ticket even when every tile has its block
    long long level_budget_mb = 0;     // SCCD_LEVEL_BUDGET_MB: budget per level buffer of the level-synchronous narrow phase (soaks on shared machines)
    int np_waves = 3;                  // SCCD_NP_WAVES=1|2: the plain walk kernel's grid fills that many waves per SIMD at most (it is built for three)
    bool cull_slabs = true;            // SCCD_CULL_SLABS=0: the projection cull looks at the whole step whatever the launches ask (round 5's first cull)
    int spec_break_every = 0;          // SCCD_SPEC_BREAK=N: every N-th speculative build is declared a failed guess (measures what a miss costs)
    static int num(const char* name, int dflt)
    {
        const char* e = std::getenv(name);
        return e ? std::atoi(e) : dflt;
    }
    LabEnv()
    {
        np_diag = num("SCCD_NP_DIAG", 0);
        sweep_diag = num("SCCD_SWEEP_DIAG", 0);
        speculate = num("SCCD_SPECULATE", 1) != 0;
        sort_tickets = num("SCCD_SORT_TICKETS", 0) != 0;
        level_budget_mb = num("SCCD_LEVEL_BUDGET_MB", 0);
        spec_break_every = num("SCCD_SPEC_BREAK", 0);
        cull_slabs = num("SCCD_CULL_SLABS", 1) != 0;
        np_waves = (int)num("SCCD_NP_WAVES", 3);
    }
};
inline const LabEnv& lab_env()
{
    static const LabEnv e;
    return e;
}

// ------------------------------------------------------------------------------------------
struct sccd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    std::string err;

    // options (sccd.h SCCD_OPT_*)
    int64_t spec_hits = 0, spec_misses = 0; // speculative builds whose guess held / broke (api.hip speculation_settle; SCCD_OPT_SPEC_*)
    int arith = 1; // fused multiply-adds where nvcc -fmad=true (the reference builds with --use_fast_math, CMakeLists.txt:219-225) fuses; 0: strict
    int narrow_algo = 0;
    int sweep_algo = 0;
    int sort_axis = 0;
    int shard_rank = 0;
    int shard_count = 1;
    int64_t overlap_capacity = 0;
    int profile = 0;
    int scalar_f32 = 0; // SCCD_OPT_SCALAR: 1 = the reference's float build
    int sweep_blocks_per_cu = 0; // 0 = the sweep kernel's own choice (a full CU); ccd()'s helper context sweeps with half
    int passes_apart = 0;      // SCCD_OPT_PASSES_APART
    int limit_level_order = 0; // SCCD_OPT_LIMIT_LEVEL_ORDER: 1 = every check limit on the level-synchronous kernels (cross-check)
    // ccd() on a mesh whose previous step found an impact starts from a BOUND a quarter above that step's TOI instead of 1
    // (drivers.hip ccd_on_mesh: exact -- a result below the bound is the result; a result AT the bound proves nothing and the
    // step is redone from 1); SCCD_OPT_TOI_GUESS = 0 turns it off
    int toi_guess_on = 1;
    double toi_guess = 1.0;
    const void* toi_guess_mesh = nullptr;
    int toi_guess_n[3] = { 0, 0, 0 };
    int64_t toi_guess_hits = 0, toi_guess_misses = 0;
    int toi_guess_rest = 0, toi_guess_backoff = 4; // steps without a bound after a miss (doubling up to 64, halved by a hit)
    double toi_last = -1.0;    // what the last ccd() on toi_guess_mesh returned (-1: nothing yet)
    int two_halves_off = 0;    // this call only (ccd_on_mesh): the last call on the same mesh found nothing before 0.5 -- one launch per pass
    int two_halves = 1;        // SCCD_OPT_TWO_HALVES: a plain narrow launch from a TOI above 0.5 is two launches over the halves of time (narrow_walk.inc)
    int cull_on = 1;           // SCCD_OPT_CULL: ccd()'s passes drop the pairs that provably have no impact before the bisection (narrow_cull.inc)
    int cell_factor_milli = 0; // SCCD_OPT_CELL_FACTOR_MILLI: grid cell size in thousandths of the mean box extent (0: the default, 4000; < 0: one cell)
    int build_scan = 0;        // SCCD_OPT_BUILD_SCAN: 1 = count -> device-wide scan -> fill (entries in box order) instead of the one-pass append
    // narrow_counters_upload() already put {zeros, this TOI} into the narrow phase's counters (it rides ahead of
    // the sweep in ccd(), so that the narrow kernel can start right behind the sweep's read-back)
    bool np_uploaded = false;
    // the next build's grid kernel also starts the narrow counters from this TOI (ccd(): drivers.hip); and: the build zeroed
    // the sweep counters, the next sweep need not (launch_grid_setup / bp_detect_partial)
    bool np_init_pending = false, sweep_cnt_cleared = false;
    double np_init_toi = 0;
    const unsigned long long* np_init_peer = nullptr; // ... and leaves this address in the counters' peer_word (NarrowCounters; null: none)
    hipEvent_t side_event2 = nullptr; // ccd(): "the helper's stream has reached its sweep" (drivers.hip)
    hipEvent_t side_event4 = nullptr; // ccd() with a check limit: "the second pass's stream has reached its sweep" (drivers.hip)
    hipEvent_t side_event3 = nullptr; // ccd(): "the helper's sweep and cull are done" (their counters are read through this context's stream)
    // ccd(): the helper's records kernel (edge list) is ordered behind the END of this context's two-list records kernel (vertices +
    // faces) -- build.hip: records_gate_signal / records_gate_wait (from a mesh size on: drivers.hip)
    StageGate records_gate;                    // the caller's context owns it
    StageGate* records_gate_signal = nullptr;  // caller's context: its own gate while a ccd() call uses it
    StageGate* records_gate_wait = nullptr;    // helper context: the gate in front of its one-list records kernel
    double np_uploaded_toi = 0;
    // a narrow-phase launch of ANOTHER context that shares this one's TOI word is running on that stream: before this
    // context resets its counters (fallback paths) it waits for it
    hipStream_t np_peer_stream = nullptr;
    // a narrow-phase call with a check limit that runs on the fast kernel (narrow.hip: the certificate): the TOI it started from
    bool np_limit_fast = false;
    double np_toi_init = 0;
    unsigned np_limit_cap = 0; // ... and the room its (query, time) records were given
    bool np_cert_in_verdict = false; // ... and: the certificate's inputs come with the launch's verdict (run_walk: np_cert_k ran behind it)
    // per-query output WITH a check limit, served by the fast kernel without the limit: narrow_phase_end redoes the queries that
    // reported an impact -- and only those -- in the reference's level order with the limit (narrow.hip)
    bool np_pq_limit = false;
    int64_t max_overlap_cutoff = 0;
    int64_t memory_limit_mb = 0;

    // profiling: accumulated per kernel class
    double prof_ms[SCCD_PROF_COUNT] = { 0 };
    int64_t prof_launches[SCCD_PROF_COUNT] = { 0 };
    struct PendingEvent {
        int cls;
        hipEvent_t a, b;
    };
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;

    // scratch shared by the pipeline stages
    DevBuf sort_tmp_keys, sort_tmp_vals, sort_hist, sort_status;
    // ccd() with the collision list: per-query TOIs of a pass, the compacted records and their query numbers (kept between calls:
    // five allocations and frees per pass were most of a 9.7 ms call on the 1M-triangle cloth)
    DevBuf col_pq, col_out, col_idx;
    DevBuf scalars;      // small device-side counters block
    PinnedBuf h_scalars; // pinned mirror for async read-back
    // The read-back mailbox (ReadBack): host-coherent pinned memory the GATHER KERNEL writes -- [0, 8 KB) the items, one
    // sequence word behind them.  mailbox_dev is the same memory as the device addresses it.
    PinnedBuf mailbox;
    char* mailbox_dev = nullptr;
    // THE EARLY VERDICT of a narrow phase in two halves of time (narrow_walk.inc): np_verdict_k, the kernel between the two
    // launches, also leaves the pass's counters and a sequence word HERE (host-coherent pinned memory, like the mailbox); if the first
    // half found its impact the host has its result then and does not wait for what is still enqueued behind that kernel -- the
    // second half's cull, its launch and the read-back, three launches that find nothing to do.
    PinnedBuf verdict;
    char* verdict_dev = nullptr;
    unsigned long long verdict_seq = 0; // the number the next verdict carries
    bool verdict_armed = false;         // narrow_phase_begin enqueued one for narrow_phase_end to look at
    unsigned long long rb_seq = 0;
    // THE DEVICE'S OWN ACCOUNT OF A ccd() STEP (SCCD_OPT_DEVICE_SPAN_NS): the first kernel of the step stores the device's real-time
    // clock into the mailbox's tail (boxes.hip vertex_boxes_k), every read-back kernel and early verdict stores it again behind its
    // items (api.hip readback_gather_k, narrow_walk.inc np_verdict_k); the span between the first and the LAST of them, in this
    // context and its helper, is what the device spent on the step -- beside the host's clock it tells a step the chip was slow on
    // from a step the host was late for (bench.py device_span_ms)
    bool step_stamp_armed = false;          // the next vertex-box launch carries the first stamp
    unsigned long long step_t_last = 0;     // the latest end stamp a read-back of this context has seen since the step began
    long long device_span_ns = -1;          // of the last ccd() call on a mesh (-1: none yet)
    int wall_clock_khz = 100000;            // the rate of that clock (hipDeviceAttributeWallClockRate)
    long long read_backs = 0;               // SCCD_OPT_READ_BACKS: ReadBack::sync calls (a launch + a polled word each) since the context was made
    long long host_waits = 0;               // SCCD_OPT_HOST_WAITS: read-backs and early verdicts the host has waited for since the context was made (its helper counts its own)
    DevBuf np_scratch0, np_scratch1, np_scratch2, np_scratch3, np_scratch4, np_scratch3_ovf, np_scratch5;
    DevBuf np_cull_list; // sccd_narrow_phase: what the projection cull keeps of a caller's pair list (+ its two counters)
    DevBuf tmp0, tmp1, tmp2;
    struct sccd_mesh* scratch_mesh = nullptr; // the mesh behind the host-matrix drivers (api.hip: scratch_mesh_from_host)
    void* pipeline = nullptr; // cached pipeline objects (api.hip)
    // ccd(): the edge-edge lists are built by a helper context (own stream, scratch and pinned mirror) on a worker
    // thread while this context does the vertex-face pass; side_event orders the helper's stream behind the boxes
    sccd_ctx* side = nullptr;
    hipEvent_t side_event = nullptr;
};

// RAII profile scope: records a hipEvent pair on the context's stream around a kernel class
struct ProfScope {
    sccd_ctx* c;
    int cls;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(sccd_ctx* ctx, int k) : c(ctx), cls(k)
    {
        // (two event records per scope are not free: ~0.15 ms of a 2.3 ms ccd() step with every class on)
        if (!c->profile || (c->profile != 1 && !((c->profile >> 1) & (1 << k)))) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!c->event_pool.empty()) {
                e = c->event_pool.back();
                c->event_pool.pop_back();
            } else {
                SCCD_HIP(hipEventCreate(&e));
            }
            return e;
        };
        a = get();
        b = get();
        SCCD_HIP(hipEventRecord(a, c->stream));
    }
    ~ProfScope()
    {
        if (!c->profile || !a) return;
        (void)hipEventRecord(b, c->stream);
        c->pending.push_back({ cls, a, b });
        c->prof_launches[cls]++;
    }
};

void sccd_collect_profile(sccd_ctx* c); // api.cpp


// Small device -> host reads (counters, the grid, the TOI: a few hundred bytes, each on a call's critical path).  As copies they cost
// a copy kernel PER ITEM plus an event the host polls: ~14 us per read-back, and a copy kernel needs more registers than a CU full of
// narrow-phase waves has left (168 x 3 of 512 per SIMD lane: 8 are free), so a read-back issued beside the narrow phase sat in its
// queue until the first of those waves retired (165 us: round 3).  Since round 4 ONE single-wave kernel of <= 8 vector registers
// (readback_gather_k, api.hip) gathers every item into host-coherent pinned memory (sccd_ctx::mailbox), waits for its stores and stores
// a sequence number behind the items; the host polls that word.  No event, no copy engine, and the kernel fits beside resident
// narrow-phase waves.  (The default ccd() step needs none of these any more: its passes' verdicts bring everything -- drivers.hip.)
constexpr size_t SCCD_MAILBOX_BYTES = 8192; // items; the sequence word sits right behind
struct ReadBackItems { // by value into the gather kernel
    const unsigned* src[8];
    unsigned off_words[8], n_words[8];
    int n;
};
void readback_gather_launch(sccd_ctx* c, const ReadBackItems& it, unsigned long long seq); // api.hip
struct ReadBack {
    sccd_ctx* c;
    size_t off = 0;
    struct Item {
        void* dst;
        size_t off, bytes;
    };
    Item items[8];
    ReadBackItems dev {};
    int n = 0;
    explicit ReadBack(sccd_ctx* ctx) : c(ctx) {}
    void add(void* dst, const void* src_dev, size_t bytes)
    {
        if (n >= 8 || off + bytes > SCCD_MAILBOX_BYTES) throw SccdError { SCCD_E_INVALID, "ReadBack: too many items" };
        if ((bytes & 3) || (reinterpret_cast<uintptr_t>(src_dev) & 3)) throw SccdError { SCCD_E_INVALID, "ReadBack: items are whole aligned words" };
        dev.src[n] = static_cast<const unsigned*>(src_dev);
        dev.off_words[n] = (unsigned)(off / 4);
        dev.n_words[n] = (unsigned)(bytes / 4);
        items[n++] = Item { dst, off, bytes };
        off += (bytes + 15) & ~(size_t)15;
    }
    void sync()
    {
        c->host_waits += 1;
        c->read_backs += 1;
        const char* const from = c->mailbox.as<char>();
        dev.n = n;
        const unsigned long long want = ++c->rb_seq;
        readback_gather_launch(c, dev, want);
        const unsigned long long* const word = reinterpret_cast<const unsigned long long*>(from + SCCD_MAILBOX_BYTES);
        // (polled: a blocking wait puts the thread to sleep and the wake-up alone costs tens of microseconds.  A stream that drained --
        // or failed -- without the word having arrived must not hang the caller: looked at now and then)
        for (unsigned spins = 1; __atomic_load_n(word, __ATOMIC_ACQUIRE) != want; spins++) {
            __builtin_ia32_pause();
            if ((spins & 0xFFFFu) != 0) continue;
            const hipError_t e = hipStreamQuery(c->stream);
            if (e == hipErrorNotReady) continue;
            SCCD_HIP(e);
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) != want) throw SccdError { SCCD_E_HIP, "ReadBack: the stream drained without the mailbox word" };
        }
        for (int i = 0; i < n; i++) std::memcpy(items[i].dst, from + items[i].off, items[i].bytes);
        unsigned long long t_end; // (the end stamp the gather kernel left in front of its sequence word: sccd_ctx::step_t_last)
        std::memcpy(&t_end, from + SCCD_MAILBOX_BYTES + 72, sizeof t_end);
        if (t_end > c->step_t_last) c->step_t_last = t_end;
        n = 0;
        off = 0;
    }
};

// ------------------------------------------------------------------------------------------
// device-side helpers
#if defined(__HIPCC__)

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// number of set bits of `mask` below this lane
__device__ __forceinline__ int mbcnt64(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }

// Order LDS writes of this wave before later LDS reads by its other lanes (single-wave
// producer/consumer).  The LDS executes one wave's instructions in issue order, so only the
// COMPILER must not reorder: wavefront-scope fences emit no instruction.  (A workgroup-scope
// fence here costs an s_waitcnt vmcnt(0), i.e. a wait for every global store/load in flight.)
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class T> __device__ __forceinline__ T wave_bcast(T v, int src_lane)
{
    return __shfl(v, src_lane, 64);
}

__device__ __forceinline__ unsigned readfirst_u32(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    return v;
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ int wave_incl_scan(int v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (l >= o) v += t;
    }
    return v;
}

// ---- wave reductions on the DPP path --------------------------------------------------------------
// __shfl_xor / __shfl_up compile to ds_bpermute_b32: an LDS round trip (~150 cycles) per step, six dependent steps per
// reduction.  The same reductions as six DPP moves inside the vector ALU (row_shr 1 / 2 / 4 / 8 inside each row of 16,
// then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3): an INCLUSIVE SCAN over the 64 lanes, its last
// lane the reduction.  Lanes that would read past their row keep `identity` (old operand, bound_ctrl off).
// EVERY lane of the wave must be active.
#define SCCD_DPP_STEP(OP, CTRL, ROWMASK) v = OP(v, (unsigned)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROWMASK, 0xf, false))
#define SCCD_DPP_SCAN(OP)            \
    SCCD_DPP_STEP(OP, 0x111, 0xf);   \
    SCCD_DPP_STEP(OP, 0x112, 0xf);   \
    SCCD_DPP_STEP(OP, 0x114, 0xf);   \
    SCCD_DPP_STEP(OP, 0x118, 0xf);   \
    SCCD_DPP_STEP(OP, 0x142, 0xa);   \
    SCCD_DPP_STEP(OP, 0x143, 0xc)
__device__ __forceinline__ unsigned dpp_op_min(unsigned a, unsigned b) { return min(a, b); }
__device__ __forceinline__ unsigned dpp_op_max(unsigned a, unsigned b) { return max(a, b); }
__device__ __forceinline__ unsigned dpp_op_add(unsigned a, unsigned b) { return a + b; }
// wave-uniform results (scalar registers)
__device__ __forceinline__ unsigned wave_min_u32_dpp(unsigned v)
{
    const unsigned identity = 0xFFFFFFFFu;
    SCCD_DPP_SCAN(dpp_op_min);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_max_u32_dpp(unsigned v)
{
    const unsigned identity = 0u;
    SCCD_DPP_SCAN(dpp_op_max);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// inclusive prefix sum over the 64 lanes; *total (wave-uniform) = the sum
__device__ __forceinline__ unsigned wave_incl_scan_dpp(unsigned v, unsigned* total)
{
    const unsigned identity = 0u;
    SCCD_DPP_SCAN(dpp_op_add);
    *total = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
    return v;
}

// ---- order-preserving 32-bit sort key of a double -------------------------------------------
// K(x) = top 32 bits of the monotone u64 image of x (+0.0 canonicalises -0.0).  Monotone
// non-decreasing: a <= b  =>  K(a) <= K(b).  The sweep only needs that (DESIGN.md "Sort key").
__device__ __forceinline__ unsigned key32(double x)
{
    x = x + 0.0;
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    b ^= (b >> 63) ? 0xFFFFFFFFFFFFFFFFull : 0x8000000000000000ull;
    return (unsigned)(b >> 32);
}

// nextafter(x, -DBL_MAX) / nextafter(x, +DBL_MAX) of the reference (scalar.hpp:31-49) by bit
// arithmetic: no libm call, identical results including zeros, subnormals and infinities.
__device__ __forceinline__ double nextafter_up(double x)
{
    if (x != x) return x;
    const double dmax = 1.7976931348623157e308;
    if (x == dmax) return x;      // x == y
    if (x > dmax) return dmax;    // +inf steps down towards y
    if (x == 0.0) return __longlong_as_double(1ll); // smallest positive subnormal
    long long b = __double_as_longlong(x);
    b += (x > 0.0) ? 1 : -1;
    return __longlong_as_double(b);
}
__device__ __forceinline__ double nextafter_down(double x)
{
    if (x != x) return x;
    const double dmax = 1.7976931348623157e308;
    if (x == -dmax) return x;
    if (x < -dmax) return -dmax;
    if (x == 0.0) return __longlong_as_double((long long)0x8000000000000001ull);
    long long b = __double_as_longlong(x);
    b += (x > 0.0) ? -1 : 1;
    return __longlong_as_double(b);
}

#endif // __HIPCC__
