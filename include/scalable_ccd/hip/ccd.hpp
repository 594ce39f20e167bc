// scalable_ccd/hip/ccd.hpp -- header-only C++17 mirror of the reference's CUDA host API on top of
// the C ABI of libsccd_hip.so (include/sccd.h).  Same names, argument order and error behaviour
// as the reference, in namespace scalable_ccd::hip instead of scalable_ccd::cuda:
//
//   reference (src/scalable_ccd/cuda/...)                         here
//   ------------------------------------------------------------  -------------------------------
//   Scalar, AABB                 scalar.hpp:13-19, aabb.cuh:12-93  Scalar, AABB (= sccd_aabb, 64 B)
//   build_vertex/edge/face_boxes broad_phase/aabb.cuh:156-188      same signatures (Matrix views)
//   DeviceAABBs                  broad_phase/aabb.cuh:122-150      DeviceAABBs
//   BroadPhase                   broad_phase/broad_phase.cuh:15-92 BroadPhase
//   DeviceMatrix<T>              utils/device_matrix.cuh:10-66     DeviceMatrix<T>
//   thrust::device_vector<int2>  (overlap list)                    DeviceVector<int2> (size(), data())
//   MemoryHandler                memory_handler.hpp:7-44           MemoryHandler (same fields)
//   narrow_phase<is_vf>          narrow_phase/narrow_phase.cuh:30  narrow_phase<is_vf>, same argument list
//   ccd                          ccd.cuh:26-38                     ccd
//   ipc_ccd_strategy             ipc_ccd_strategy.hpp:17-24        ipc_ccd_strategy
//
// Eigen is not required: matrices are passed as column-major views (ConstMatrixView), which is
// exactly Eigen::MatrixXd / MatrixXi storage; with Eigen available the overloads at the bottom
// accept Eigen matrices directly.  Every failure throws std::runtime_error like gpuErrchk
// (utils/assert.cuh:18-27).
#pragma once

#include "../../sccd.h"

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <algorithm>
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define SCCD_HIP_HAVE_EIGEN 1
#endif
#endif

namespace scalable_ccd::hip {

using Scalar = double; // SCALABLE_CCD_USE_DOUBLE=ON (CMakeLists.txt:69)
using AABB = ::sccd_aabb;

/// Column-major matrix view (Eigen default storage): element (r, c) = data[r + c * rows].
template <class T> struct ConstMatrixView {
    const T* data = nullptr;
    int rows = 0, cols = 0;
    ConstMatrixView() = default;
    ConstMatrixView(const T* d, int r, int c) : data(d), rows(r), cols(c) { }
#ifdef SCCD_HIP_HAVE_EIGEN
    ConstMatrixView(const Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic>& m)
        : data(m.data()), rows((int)m.rows()), cols((int)m.cols()) { }
#endif
};
using MatrixXdView = ConstMatrixView<double>;
using MatrixXiView = ConstMatrixView<int32_t>;

/// One device + stream + scratch memory.  The reference keeps this state in globals
/// (device_init_id, __constant__ CONFIG); here it is explicit and defaulted.
/// A Context is SINGLE-THREADED: one call at a time (the reference's globals are no different), and narrow_phase() /
/// ccd() calls on one Context are not re-entrant -- the packed mesh behind the four-DeviceMatrix argument list
/// (call_mesh) is ONE object owned by the Context, refilled by every such call; a call that fails on bad indices leaves
/// it holding clamped indices until the next call refills it.  Threads that run CCD concurrently each make their own Context.
class Context {
public:
    explicit Context(int device = 0)
    {
        // (the library writes sccd_stats and the profile arrays into memory sized by THIS header: refuse one built from another)
        if (sccd_abi_sizeof_stats() != sizeof(sccd_stats) || sccd_abi_prof_count() != SCCD_PROF_COUNT)
            throw std::runtime_error(std::string("libsccd_hip.so (") + sccd_version() + ") was built from another include/sccd.h");
        if (sccd_create(device, &m_ctx) != SCCD_OK)
            throw std::runtime_error(std::string("sccd_create: ") + sccd_last_error(nullptr));
    }
    ~Context()
    {
        sccd_mesh_destroy(m_call_mesh);
        sccd_destroy(m_ctx);
    }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    sccd_ctx* get() const { return m_ctx; }
    void check(int rc) const
    {
        if (rc != SCCD_OK) throw std::runtime_error(sccd_last_error(m_ctx));
    }
    void set_option(int option, int64_t value) { check(sccd_set_option(m_ctx, option, value)); }
    static Context& default_context()
    {
        static Context ctx(0);
        return ctx;
    }
    /// The packed mesh behind narrow_phase() calls that hand over four DeviceMatrix (the reference's argument list): kept
    /// by the context and refilled call after call -- no allocation per call.
    sccd_mesh* call_mesh(const double* V0, const double* V1, int nV, const int32_t* E, int nE, const int32_t* F, int nF)
    {
        if (!m_call_mesh) check(sccd_mesh_create(m_ctx, V0, V1, nV, E, nE, F, nF, /*src_on_device=*/1, &m_call_mesh));
        else check(sccd_mesh_assign(m_call_mesh, V0, V1, nV, E, nE, F, nF, /*src_on_device=*/1));
        return m_call_mesh;
    }

private:
    sccd_ctx* m_ctx = nullptr;
    sccd_mesh* m_call_mesh = nullptr;
};

/// Two packed ints, the element type of the overlap list (CUDA's int2 in the reference).
struct int2 {
    int x, y;
};

/// Device memory with the interface the reference uses of thrust::device_vector<T>: size(), data(), resize(), clear().
/// Owning (allocated through the C ABI) or, for BroadPhase::overlaps(), a view of library-owned memory.
template <class T> class DeviceVector {
public:
    DeviceVector() = default;
    explicit DeviceVector(size_t n, Context& ctx = Context::default_context()) : m_ctx(&ctx) { resize(n); }
    DeviceVector(const std::vector<T>& host, Context& ctx = Context::default_context()) : m_ctx(&ctx) { assign(host.data(), host.size()); }
    ~DeviceVector() { release(); }
    DeviceVector(const DeviceVector&) = delete;
    DeviceVector& operator=(const DeviceVector&) = delete;
    DeviceVector(DeviceVector&& o) noexcept { swap(o); }
    DeviceVector& operator=(DeviceVector&& o) noexcept
    {
        swap(o);
        return *this;
    }
    size_t size() const { return m_size; }
    bool empty() const { return m_size == 0; }
    T* data() { return m_data; }
    const T* data() const { return m_data; }
    void clear() { m_size = 0; }
    /// keeps the first min(size(), n) elements, like thrust::device_vector::resize
    void resize(size_t n)
    {
        if (n > m_cap) {
            T* grown = nullptr;
            ctx().check(sccd_dev_alloc(ctx().get(), n * sizeof(T), reinterpret_cast<void**>(&grown)));
            if (m_data && m_size) ctx().check(sccd_dev_copy(ctx().get(), grown, m_data, m_size * sizeof(T)));
            release();
            m_data = grown;
            m_cap = n;
            m_owned = true;
        }
        m_size = n;
    }
    void assign(const T* host, size_t n)
    {
        resize(n);
        if (n) ctx().check(sccd_dev_upload(ctx().get(), m_data, host, n * sizeof(T)));
    }
    std::vector<T> to_host() const
    {
        std::vector<T> h(m_size);
        if (m_size) ctx().check(sccd_dev_download(ctx().get(), h.data(), m_data, m_size * sizeof(T)));
        return h;
    }
    /// (BroadPhase) point at library-owned memory
    void view(const T* d, size_t n, Context& c)
    {
        release();
        m_ctx = &c;
        m_data = const_cast<T*>(d);
        m_size = n;
        m_cap = 0;
        m_owned = false;
    }
    Context& ctx() const { return m_ctx ? *m_ctx : Context::default_context(); }

private:
    void release()
    {
        if (m_owned && m_data) sccd_dev_free(ctx().get(), m_data);
        m_data = nullptr;
        m_size = m_cap = 0;
        m_owned = false;
    }
    void swap(DeviceVector& o)
    {
        std::swap(m_ctx, o.m_ctx);
        std::swap(m_data, o.m_data);
        std::swap(m_size, o.m_size);
        std::swap(m_cap, o.m_cap);
        std::swap(m_owned, o.m_owned);
    }
    Context* m_ctx = nullptr;
    T* m_data = nullptr;
    size_t m_size = 0, m_cap = 0;
    bool m_owned = false;
};

/// A column-major matrix stored on the device (utils/device_matrix.cuh:10-66).
template <typename T> class DeviceMatrix {
public:
    DeviceMatrix() = default;
    DeviceMatrix(const size_t rows, const size_t cols, Context& ctx = Context::default_context())
        : m_rows(rows), m_cols(cols), m_data(rows * cols, ctx) { }
    DeviceMatrix(const ConstMatrixView<T>& mat, Context& ctx = Context::default_context())
        : m_rows((size_t)mat.rows), m_cols((size_t)mat.cols), m_data(0, ctx)
    {
        m_data.assign(mat.data, m_rows * m_cols);
    }
    void operator=(const ConstMatrixView<T>& mat)
    {
        m_rows = (size_t)mat.rows;
        m_cols = (size_t)mat.cols;
        m_data.assign(mat.data, m_rows * m_cols);
    }
#ifdef SCCD_HIP_HAVE_EIGEN
    DeviceMatrix(const Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic>& mat, Context& ctx = Context::default_context())
        : DeviceMatrix(ConstMatrixView<T>(mat), ctx) { }
#endif
    T* data() { return m_data.data(); }
    const T* data() const { return m_data.data(); }
    size_t size() const { return m_data.size(); }
    size_t rows() const { return m_rows; }
    size_t cols() const { return m_cols; }
    Context& ctx() const { return m_data.ctx(); }

private:
    size_t m_rows = 0, m_cols = 0;
    DeviceVector<T> m_data;
};

/// MemoryHandler (memory_handler.hpp:7-44): the knobs of the reference's memory policy.  Here the library sizes its
/// buffers itself (counted overflow -> exact-size rerun); what a caller can set keeps its meaning:
///   memory_limit_GB      budget of the overlap list   (SCCD_OPT_MEMORY_LIMIT_MB)
///   MAX_OVERLAP_CUTOFF   boxes swept per detect_overlaps_partial() call, 0 = all (SCCD_OPT_MAX_OVERLAP_CUTOFF)
///   MAX_OVERLAP_SIZE     initial capacity of the overlap list in pairs, 0 = automatic (SCCD_OPT_OVERLAP_CAPACITY)
/// real_count is written back after every detect_overlaps_partial(); the other fields are kept for source
/// compatibility and have no effect.
struct MemoryHandler {
    size_t MAX_OVERLAP_CUTOFF = 0;
    size_t MAX_OVERLAP_SIZE = 0;
    size_t MAX_UNIT_SIZE = 0;
    size_t MAX_QUERIES = 0;
    size_t per_overlap_memory_size = 256 + 3 * sizeof(int2); // sizeof(CCDData) + 3 * sizeof(int2), memory_handler.hpp:18
    int real_count = 0;
    int memory_limit_GB = 0;
};

// --- boxes (aabb.cuh:156-188) -------------------------------------------------------------------

inline void build_vertex_boxes(const MatrixXdView& vertices_t0, const MatrixXdView& vertices_t1,
                               std::vector<AABB>& vertex_boxes, double inflation_radius = 0,
                               Context& ctx = Context::default_context())
{
    if (vertices_t0.rows != vertices_t1.rows || vertices_t0.cols != 3 || vertices_t1.cols != 3)
        throw std::runtime_error("build_vertex_boxes: vertices must both be n x 3");
    vertex_boxes.resize((size_t)vertices_t0.rows);
    ctx.check(sccd_build_vertex_boxes(ctx.get(), vertices_t0.data, vertices_t1.data, vertices_t0.rows,
                                      inflation_radius, vertex_boxes.data()));
}
inline void build_vertex_boxes(const MatrixXdView& vertices, std::vector<AABB>& vertex_boxes,
                               double inflation_radius = 0, Context& ctx = Context::default_context())
{
    build_vertex_boxes(vertices, vertices, vertex_boxes, inflation_radius, ctx);
}
inline void build_edge_boxes(const std::vector<AABB>& vertex_boxes, const MatrixXiView& edges,
                             std::vector<AABB>& edge_boxes, Context& ctx = Context::default_context())
{
    if (edges.rows > 0 && edges.cols != 2) throw std::runtime_error("build_edge_boxes: edges must be m x 2");
    edge_boxes.resize((size_t)edges.rows);
    ctx.check(sccd_build_edge_boxes(ctx.get(), vertex_boxes.data(), (int)vertex_boxes.size(), edges.data,
                                    edges.rows, edge_boxes.data()));
}
inline void build_face_boxes(const std::vector<AABB>& vertex_boxes, const MatrixXiView& faces,
                             std::vector<AABB>& face_boxes, Context& ctx = Context::default_context())
{
    if (faces.rows > 0 && faces.cols != 3) throw std::runtime_error("build_face_boxes: faces must be k x 3");
    face_boxes.resize((size_t)faces.rows);
    ctx.check(sccd_build_face_boxes(ctx.get(), vertex_boxes.data(), (int)vertex_boxes.size(), faces.data,
                                    faces.rows, face_boxes.data()));
}

/// Boxes resident on the device (DeviceAABBs, aabb.cuh:122-150).
struct DeviceAABBs {
    DeviceAABBs() = default;
    explicit DeviceAABBs(const std::vector<AABB>& boxes, Context& ctx = Context::default_context()) : m_ctx(&ctx)
    {
        ctx.check(sccd_boxes_create(ctx.get(), boxes.data(), (int)boxes.size(), 0, &m_boxes));
    }
    ~DeviceAABBs() { sccd_boxes_destroy(m_boxes); }
    DeviceAABBs(const DeviceAABBs&) = delete;
    DeviceAABBs& operator=(const DeviceAABBs&) = delete;
    size_t size() const { return (size_t)sccd_boxes_size(m_boxes); }
    sccd_boxes* get() const { return m_boxes; }

private:
    Context* m_ctx = nullptr;
    sccd_boxes* m_boxes = nullptr;
};

/// class BroadPhase (broad_phase.cuh:15-92).
class BroadPhase {
public:
    BroadPhase() : BroadPhase(std::make_shared<MemoryHandler>()) { }
    explicit BroadPhase(std::shared_ptr<MemoryHandler> _memory_handler, Context& ctx = Context::default_context())
        : memory_handler(std::move(_memory_handler)), m_ctx(&ctx)
    {
        ctx.check(sccd_broad_phase_create(ctx.get(), &m_bp));
    }
    explicit BroadPhase(Context& ctx) : BroadPhase(std::make_shared<MemoryHandler>(), ctx) { }
    ~BroadPhase() { sccd_broad_phase_destroy(m_bp); }
    BroadPhase(const BroadPhase&) = delete;
    BroadPhase& operator=(const BroadPhase&) = delete;

    /// Forget the boxes and the overlaps (broad_phase.cu:103-119; the reference also resets the caller's memory
    /// handler and clears the CALLER's boxes here -- two quirks that are not reproduced).
    void clear()
    {
        m_a.reset();
        m_b.reset();
        d_overlaps.view(nullptr, 0, *m_ctx);
        sccd_broad_phase_destroy(m_bp);
        m_bp = nullptr;
        m_ctx->check(sccd_broad_phase_create(m_ctx->get(), &m_bp));
    }
    void build(const std::shared_ptr<DeviceAABBs> boxes)
    {
        if (!boxes) throw std::runtime_error("BroadPhase::build: boxes are null");
        m_a = boxes;
        m_b.reset();
        const HandlerScope handler_scope(m_ctx, memory_handler.get());
        m_ctx->check(sccd_broad_phase_build(m_bp, boxes->get(), nullptr));
    }
    void build(const std::shared_ptr<DeviceAABBs> boxesA, const std::shared_ptr<DeviceAABBs> boxesB)
    {
        if (!boxesA || !boxesB) throw std::runtime_error("BroadPhase::build: boxes are null");
        m_a = boxesA;
        m_b = boxesB;
        const HandlerScope handler_scope(m_ctx, memory_handler.get());
        m_ctx->check(sccd_broad_phase_build(m_bp, boxesA->get(), boxesB->get()));
    }
    /// One sweep step; the overlaps stay on the device, valid until the next call (broad_phase.cuh:41-44).
    const DeviceVector<int2>& detect_overlaps_partial()
    {
        const int32_t* p = nullptr;
        int64_t n = 0;
        const HandlerScope handler_scope(m_ctx, memory_handler.get());
        m_ctx->check(sccd_broad_phase_detect_overlaps_partial(m_bp, &p, &n));
        d_overlaps.view(reinterpret_cast<const int2*>(p), (size_t)n, *m_ctx);
        if (memory_handler) memory_handler->real_count = (int)n;
        return d_overlaps;
    }
    std::vector<std::pair<int, int>> detect_overlaps()
    {
        int32_t* p = nullptr;
        int64_t n = 0;
        const HandlerScope handler_scope(m_ctx, memory_handler.get());
        m_ctx->check(sccd_broad_phase_detect_overlaps(m_bp, &p, &n));
        std::vector<std::pair<int, int>> out((size_t)n);
        for (int64_t i = 0; i < n; i++) out[(size_t)i] = { p[2 * i], p[2 * i + 1] };
        sccd_free(p);
        return out;
    }
    bool is_complete() const { return sccd_broad_phase_is_complete(m_bp) != 0; }
    /// The boxes handed to build() (list A; unsorted, as given) -- broad_phase.cuh:60.
    std::shared_ptr<DeviceAABBs> boxes() { return m_a; }
    size_t num_boxes() const { return (size_t)sccd_broad_phase_num_boxes(m_bp); }
    /// The overlaps of the last detect_overlaps_partial(), on the device (broad_phase.cuh:66-68).
    const DeviceVector<int2>& overlaps() { return d_overlaps; }

    int threads_per_block = 32; // accepted for source compatibility; the sweep kernel picks its own launch shape

private:
    /// The handler belongs to THIS BroadPhase (broad_phase.cuh:76): its three settings apply for the duration of one call
    /// and the context's own options come back afterwards (a shared context must not inherit one object's limits).
    struct HandlerScope {
        Context* c = nullptr;
        int64_t saved[3] = { 0, 0, 0 };
        HandlerScope(Context* ctx, const MemoryHandler* h) : c(h ? ctx : nullptr)
        {
            if (!c) return;
            saved[0] = sccd_get_option(c->get(), SCCD_OPT_MEMORY_LIMIT_MB);
            saved[1] = sccd_get_option(c->get(), SCCD_OPT_MAX_OVERLAP_CUTOFF);
            saved[2] = sccd_get_option(c->get(), SCCD_OPT_OVERLAP_CAPACITY);
            if (h->memory_limit_GB > 0) c->set_option(SCCD_OPT_MEMORY_LIMIT_MB, (int64_t)h->memory_limit_GB * 1024);
            if (h->MAX_OVERLAP_CUTOFF > 0) c->set_option(SCCD_OPT_MAX_OVERLAP_CUTOFF, (int64_t)h->MAX_OVERLAP_CUTOFF);
            if (h->MAX_OVERLAP_SIZE > 0) c->set_option(SCCD_OPT_OVERLAP_CAPACITY, (int64_t)h->MAX_OVERLAP_SIZE);
        }
        ~HandlerScope()
        {
            if (!c) return;
            (void)sccd_set_option(c->get(), SCCD_OPT_MEMORY_LIMIT_MB, saved[0]);
            (void)sccd_set_option(c->get(), SCCD_OPT_MAX_OVERLAP_CUTOFF, saved[1]);
            (void)sccd_set_option(c->get(), SCCD_OPT_OVERLAP_CAPACITY, saved[2]);
        }
        HandlerScope(const HandlerScope&) = delete;
        HandlerScope& operator=(const HandlerScope&) = delete;
    };
    std::shared_ptr<MemoryHandler> memory_handler;
    Context* m_ctx;
    sccd_broad_phase* m_bp = nullptr;
    std::shared_ptr<DeviceAABBs> m_a, m_b; // shared ownership as in the reference
    DeviceVector<int2> d_overlaps;          // view of the library's overlap buffer
};

/// sort_and_sweep() of the reference's CPU API (broad_phase/sort_and_sweep.hpp:28-42,
/// sort_and_sweep.cpp:197-240), served by the device path: sweeps along `sort_axis` and hands
/// back the arg-max-variance axis for the next call.  Boxes are taken by value like there.
inline void sort_and_sweep(std::vector<AABB> boxes, int& sort_axis, std::vector<std::pair<int, int>>& overlaps,
                           Context& ctx = Context::default_context())
{
    overlaps.clear();
    if (boxes.empty()) return;
    const int64_t saved = sccd_get_option(ctx.get(), SCCD_OPT_SORT_AXIS);
    ctx.set_option(SCCD_OPT_SORT_AXIS, sort_axis);
    try {
        auto d = std::make_shared<DeviceAABBs>(boxes, ctx);
        BroadPhase bp(ctx);
        bp.build(d);
        overlaps = bp.detect_overlaps();
        ctx.check(sccd_boxes_variance_axis(ctx.get(), d->get(), nullptr, &sort_axis));
    } catch (...) {
        ctx.set_option(SCCD_OPT_SORT_AXIS, saved);
        throw;
    }
    ctx.set_option(SCCD_OPT_SORT_AXIS, saved);
}
inline void sort_and_sweep(std::vector<AABB> boxesA, std::vector<AABB> boxesB, int& sort_axis,
                           std::vector<std::pair<int, int>>& overlaps, Context& ctx = Context::default_context())
{
    overlaps.clear();
    if (boxesA.empty() || boxesB.empty()) return;
    const int64_t saved = sccd_get_option(ctx.get(), SCCD_OPT_SORT_AXIS);
    ctx.set_option(SCCD_OPT_SORT_AXIS, sort_axis);
    try {
        auto a = std::make_shared<DeviceAABBs>(boxesA, ctx);
        auto b = std::make_shared<DeviceAABBs>(boxesB, ctx);
        BroadPhase bp(ctx);
        bp.build(a, b);
        overlaps = bp.detect_overlaps();
        ctx.check(sccd_boxes_variance_axis(ctx.get(), a->get(), b->get(), &sort_axis));
    } catch (...) {
        ctx.set_option(SCCD_OPT_SORT_AXIS, saved);
        throw;
    }
    ctx.set_option(SCCD_OPT_SORT_AXIS, saved);
}

/// sort_along_axis() (broad_phase/sort_and_sweep.hpp:11, sort_and_sweep.cpp:126-141): orders host boxes by min[axis].  The device
/// path sorts its own keys; this is for callers that use the reference's two-step form (sort, then sweep<>()).
inline void sort_along_axis(const int axis, std::vector<AABB>& boxes)
{
    if (axis < 0 || axis > 2) throw std::runtime_error("sort_along_axis: axis must be 0, 1 or 2");
    std::sort(boxes.begin(), boxes.end(), [axis](const AABB& a, const AABB& b) { return a.min[axis] < b.min[axis]; });
}
/// sweep<is_two_lists>() (sort_and_sweep.hpp:18-22, sort_and_sweep.cpp:143-195): the boxes of one list, or of two lists MERGED
/// with the first list's element ids flipped to -id - 1 (sort_and_sweep.cpp:228-237); pairs come back as (min id, max id), or
/// as (first-list id, second-list id) with the flip undone (:104-109); sort_axis in: the axis swept, out: the arg-max-variance
/// axis of the box centres.  Served by the device path, which orders the boxes itself.
template <bool is_two_lists>
void sweep(std::vector<AABB>& boxes, int& sort_axis, std::vector<std::pair<int, int>>& overlaps,
           Context& ctx = Context::default_context())
{
    overlaps.clear();
    if (boxes.empty()) return;
    if constexpr (!is_two_lists) {
        sort_and_sweep(boxes, sort_axis, overlaps, ctx);
    } else {
        std::vector<AABB> first, second;
        for (const AABB& box : boxes) {
            if (box.element_id < 0) {
                first.push_back(box);
                first.back().element_id = -box.element_id - 1;
            } else {
                second.push_back(box);
            }
        }
        sort_and_sweep(std::move(first), std::move(second), sort_axis, overlaps, ctx);
    }
}

/// The four DeviceMatrix objects of ccd() (ccd.cu:103-106) as one device-resident mesh.
class DeviceMesh {
public:
    DeviceMesh(const MatrixXdView& V0, const MatrixXdView& V1, const MatrixXiView& E, const MatrixXiView& F,
               Context& ctx = Context::default_context())
        : m_ctx(&ctx)
    {
        if (V0.rows != V1.rows || V0.cols != 3 || V1.cols != 3 || (E.rows > 0 && E.cols != 2) || (F.rows > 0 && F.cols != 3))
            throw std::runtime_error("mesh: V must be n x 3, E m x 2, F k x 3"); // ccd.cu:94-98
        ctx.check(sccd_mesh_create(ctx.get(), V0.data, V1.data, V0.rows, E.data, E.rows, F.data, F.rows, 0, &m_mesh));
    }
    ~DeviceMesh() { sccd_mesh_destroy(m_mesh); }
    DeviceMesh(const DeviceMesh&) = delete;
    DeviceMesh& operator=(const DeviceMesh&) = delete;
    sccd_mesh* get() const { return m_mesh; }
    Context& context() const { return *m_ctx; }

private:
    Context* m_ctx;
    sccd_mesh* m_mesh = nullptr;
};

namespace detail {
    template <bool is_vf>
    void narrow_phase_on_mesh(Context& ctx, const sccd_mesh* mesh, const int32_t* pairs, int64_t n, int pairs_on_device,
                              const int max_iter, const Scalar tol, const Scalar ms, const bool allow_zero_toi,
                              std::vector<std::tuple<int, int, Scalar>>* collisions, Scalar& toi)
    {
        sccd_collision* col = nullptr;
        int64_t ncol = 0;
        ctx.check(sccd_narrow_phase(ctx.get(), mesh, pairs, n, pairs_on_device, is_vf ? 1 : 0, max_iter, tol, ms,
                                    allow_zero_toi ? 1 : 0, &toi, collisions ? &col : nullptr, collisions ? &ncol : nullptr));
        if (collisions) {
            for (int64_t i = 0; i < ncol; i++) collisions->emplace_back(col[i].aid, col[i].bid, col[i].toi);
            sccd_free(col);
        }
    }
    struct MeshOfMatrices { // the library's packed mesh (sccd_mesh) of four device matrices, for one call (owned by the context)
        sccd_mesh* m = nullptr;
        MeshOfMatrices(Context& ctx, const DeviceMatrix<Scalar>& V0, const DeviceMatrix<Scalar>& V1,
                       const DeviceMatrix<int>& E, const DeviceMatrix<int>& F)
        {
            if (V0.rows() != V1.rows() || (V0.rows() && (V0.cols() != 3 || V1.cols() != 3)) || (E.rows() && E.cols() != 2)
                || (F.rows() && F.cols() != 3))
                throw std::runtime_error("narrow_phase: V must be n x 3, E m x 2, F k x 3");
            m = ctx.call_mesh(V0.data(), V1.data(), (int)V0.rows(), E.data(), (int)E.rows(), F.data(), (int)F.rows());
        }
    };
} // namespace detail

/// narrow_phase<is_vf>() with the reference's argument list (narrow_phase.cuh:30-46): four device matrices, the
/// overlaps ON THE DEVICE (BroadPhase::overlaps()), `threads` and `memory_handler` accepted and unused (the kernel
/// picks its own launch shape and needs no CCDData pool), toi in/out (>= 0, narrow_phase.cu:126).
template <bool is_vf>
void narrow_phase(const DeviceMatrix<Scalar>& d_vertices_t0, const DeviceMatrix<Scalar>& d_vertices_t1,
                  const DeviceMatrix<int>& d_edges, const DeviceMatrix<int>& d_faces, const DeviceVector<int2>& d_overlaps,
                  const int threads, const int max_iter, const Scalar tol, const Scalar ms, const bool allow_zero_toi,
                  std::shared_ptr<MemoryHandler> memory_handler, Scalar& toi)
{
    (void)threads;
    (void)memory_handler;
    Context& ctx = d_vertices_t0.ctx();
    detail::MeshOfMatrices mesh(ctx, d_vertices_t0, d_vertices_t1, d_edges, d_faces);
    detail::narrow_phase_on_mesh<is_vf>(ctx, mesh.m, reinterpret_cast<const int32_t*>(d_overlaps.data()),
                                        (int64_t)d_overlaps.size(), 1, max_iter, tol, ms, allow_zero_toi, nullptr, toi);
}
/// The SCALABLE_CCD_TOI_PER_QUERY overload (narrow_phase.cuh:42-44): also appends (aid, bid, toi) of the queries with toi < 1.
template <bool is_vf>
void narrow_phase(const DeviceMatrix<Scalar>& d_vertices_t0, const DeviceMatrix<Scalar>& d_vertices_t1,
                  const DeviceMatrix<int>& d_edges, const DeviceMatrix<int>& d_faces, const DeviceVector<int2>& d_overlaps,
                  const int threads, const int max_iter, const Scalar tol, const Scalar ms, const bool allow_zero_toi,
                  std::shared_ptr<MemoryHandler> memory_handler, std::vector<std::tuple<int, int, Scalar>>& collisions,
                  Scalar& toi)
{
    (void)threads;
    (void)memory_handler;
    Context& ctx = d_vertices_t0.ctx();
    detail::MeshOfMatrices mesh(ctx, d_vertices_t0, d_vertices_t1, d_edges, d_faces);
    detail::narrow_phase_on_mesh<is_vf>(ctx, mesh.m, reinterpret_cast<const int32_t*>(d_overlaps.data()),
                                        (int64_t)d_overlaps.size(), 1, max_iter, tol, ms, allow_zero_toi, &collisions, toi);
}
/// Convenience forms on a packed DeviceMesh with HOST pairs (not in the reference).
template <bool is_vf>
void narrow_phase(const DeviceMesh& mesh, const std::vector<std::pair<int, int>>& overlaps, const int max_iter,
                  const Scalar tol, const Scalar minimum_separation_distance, const bool allow_zero_toi, Scalar& toi,
                  std::vector<std::tuple<int, int, Scalar>>* collisions = nullptr)
{
    static_assert(sizeof(std::pair<int, int>) == 2 * sizeof(int32_t), "pair<int,int> must be two packed ints");
    detail::narrow_phase_on_mesh<is_vf>(mesh.context(), mesh.get(), reinterpret_cast<const int32_t*>(overlaps.data()),
                                        (int64_t)overlaps.size(), 0, max_iter, tol, minimum_separation_distance,
                                        allow_zero_toi, collisions, toi);
}

/// ccd() (ccd.cuh:26-38): earliest time of impact in [0, 1], 1 = none.
inline Scalar ccd(const MatrixXdView& vertices_t0, const MatrixXdView& vertices_t1, const MatrixXiView& edges,
                  const MatrixXiView& faces, const Scalar minimum_separation_distance, const int max_iterations,
                  const Scalar tolerance, const bool allow_zero_toi, const int memory_limit_GB = 0,
                  Context& ctx = Context::default_context())
{
    if (vertices_t0.rows != vertices_t1.rows || vertices_t0.cols != 3 || vertices_t1.cols != 3
        || (edges.rows > 0 && edges.cols != 2) || (faces.rows > 0 && faces.cols != 3))
        throw std::runtime_error("ccd: V must be n x 3, E m x 2, F k x 3"); // ccd.cu:94-98
    Scalar toi = 1;
    ctx.check(sccd_ccd(ctx.get(), vertices_t0.data, vertices_t1.data, vertices_t0.rows, edges.data, edges.rows,
                       faces.data, faces.rows, minimum_separation_distance, max_iterations, tolerance,
                       allow_zero_toi ? 1 : 0, memory_limit_GB, &toi));
    return toi;
}

/// ccd() of a SCALABLE_CCD_TOI_PER_QUERY build (ccd.cuh:26-38 with the `collisions` argument, ccd.cu:14-78):
/// also returns (aid, bid, toi) of every query with toi < 1 -- vertex-face pairs first, then edge-edge.  One call
/// into the library (sccd_ccd_collisions): the overlap pairs stay on the device between the phases.
inline Scalar ccd(const MatrixXdView& vertices_t0, const MatrixXdView& vertices_t1, const MatrixXiView& edges,
                  const MatrixXiView& faces, const Scalar minimum_separation_distance, const int max_iterations,
                  const Scalar tolerance, const bool allow_zero_toi,
                  std::vector<std::tuple<int, int, Scalar>>& collisions, const int memory_limit_GB = 0,
                  Context& ctx = Context::default_context())
{
    if (vertices_t0.rows != vertices_t1.rows || vertices_t0.cols != 3 || vertices_t1.cols != 3
        || (edges.rows > 0 && edges.cols != 2) || (faces.rows > 0 && faces.cols != 3))
        throw std::runtime_error("ccd: V must be n x 3, E m x 2, F k x 3"); // ccd.cu:94-98
    Scalar toi = 1;
    sccd_collision* col = nullptr;
    int64_t ncol = 0;
    ctx.check(sccd_ccd_collisions(ctx.get(), vertices_t0.data, vertices_t1.data, vertices_t0.rows, edges.data, edges.rows,
                                  faces.data, faces.rows, minimum_separation_distance, max_iterations, tolerance,
                                  allow_zero_toi ? 1 : 0, memory_limit_GB, &toi, &col, &ncol));
    for (int64_t i = 0; i < ncol; i++) collisions.emplace_back(col[i].aid, col[i].bid, col[i].toi);
    sccd_free(col);
    return toi;
}

/// ipc_ccd_strategy() (ipc_ccd_strategy.hpp:17-24).
inline Scalar ipc_ccd_strategy(const MatrixXdView& V0, const MatrixXdView& V1, const MatrixXiView& E,
                               const MatrixXiView& F, const Scalar min_distance, const int max_iter,
                               const Scalar tolerance, Context& ctx = Context::default_context())
{
    Scalar toi = 1;
    ctx.check(sccd_ipc_ccd_strategy(ctx.get(), V0.data, V1.data, V0.rows, E.data, E.rows, F.data, F.rows,
                                    min_distance, max_iter, tolerance, &toi));
    return toi;
}

} // namespace scalable_ccd::hip
