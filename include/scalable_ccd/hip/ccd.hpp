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
//   DeviceMatrix<T> x 4          utils/device_matrix.cuh:10-66     DeviceMesh (V0, V1, E, F together)
//   narrow_phase<is_vf>          narrow_phase/narrow_phase.cuh:30  narrow_phase<is_vf>
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
class Context {
public:
    explicit Context(int device = 0)
    {
        if (sccd_create(device, &m_ctx) != SCCD_OK)
            throw std::runtime_error(std::string("sccd_create: ") + sccd_last_error(nullptr));
    }
    ~Context() { sccd_destroy(m_ctx); }
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

private:
    sccd_ctx* m_ctx = nullptr;
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
    explicit BroadPhase(Context& ctx = Context::default_context()) : m_ctx(&ctx)
    {
        ctx.check(sccd_broad_phase_create(ctx.get(), &m_bp));
    }
    ~BroadPhase() { sccd_broad_phase_destroy(m_bp); }
    BroadPhase(const BroadPhase&) = delete;
    BroadPhase& operator=(const BroadPhase&) = delete;

    void build(const std::shared_ptr<DeviceAABBs> boxes)
    {
        if (!boxes) throw std::runtime_error("BroadPhase::build: boxes are null");
        m_a = boxes;
        m_b.reset();
        m_ctx->check(sccd_broad_phase_build(m_bp, boxes->get(), nullptr));
    }
    void build(const std::shared_ptr<DeviceAABBs> boxesA, const std::shared_ptr<DeviceAABBs> boxesB)
    {
        if (!boxesA || !boxesB) throw std::runtime_error("BroadPhase::build: boxes are null");
        m_a = boxesA;
        m_b = boxesB;
        m_ctx->check(sccd_broad_phase_build(m_bp, boxesA->get(), boxesB->get()));
    }
    /// Device pointer to int32[n][2] + n; valid until the next call (broad_phase.cuh:41-44).
    std::pair<const int32_t*, int64_t> detect_overlaps_partial()
    {
        const int32_t* p = nullptr;
        int64_t n = 0;
        m_ctx->check(sccd_broad_phase_detect_overlaps_partial(m_bp, &p, &n));
        return { p, n };
    }
    std::vector<std::pair<int, int>> detect_overlaps()
    {
        int32_t* p = nullptr;
        int64_t n = 0;
        m_ctx->check(sccd_broad_phase_detect_overlaps(m_bp, &p, &n));
        std::vector<std::pair<int, int>> out((size_t)n);
        for (int64_t i = 0; i < n; i++) out[(size_t)i] = { p[2 * i], p[2 * i + 1] };
        sccd_free(p);
        return out;
    }
    bool is_complete() const { return sccd_broad_phase_is_complete(m_bp) != 0; }
    size_t num_boxes() const { return (size_t)sccd_broad_phase_num_boxes(m_bp); }

private:
    Context* m_ctx;
    sccd_broad_phase* m_bp = nullptr;
    std::shared_ptr<DeviceAABBs> m_a, m_b; // shared ownership as in the reference
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

/// narrow_phase<is_vf>() (narrow_phase.cuh:30-46).  `overlaps` are host pairs; toi is in/out.
template <bool is_vf>
void narrow_phase(const DeviceMesh& mesh, const std::vector<std::pair<int, int>>& overlaps, const int max_iter,
                  const Scalar tol, const Scalar minimum_separation_distance, const bool allow_zero_toi, Scalar& toi,
                  std::vector<std::tuple<int, int, Scalar>>* collisions = nullptr)
{
    static_assert(sizeof(std::pair<int, int>) == 2 * sizeof(int32_t), "pair<int,int> must be two packed ints");
    sccd_collision* col = nullptr;
    int64_t ncol = 0;
    mesh.context().check(sccd_narrow_phase(
        mesh.context().get(), mesh.get(), reinterpret_cast<const int32_t*>(overlaps.data()), (int64_t)overlaps.size(),
        0, is_vf ? 1 : 0, max_iter, tol, minimum_separation_distance, allow_zero_toi ? 1 : 0, &toi,
        collisions ? &col : nullptr, collisions ? &ncol : nullptr));
    if (collisions) {
        for (int64_t i = 0; i < ncol; i++) collisions->emplace_back(col[i].aid, col[i].bid, col[i].toi);
        sccd_free(col);
    }
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
/// also returns (aid, bid, toi) of every query with toi < 1 -- vertex-face pairs first, then edge-edge.
inline Scalar ccd(const MatrixXdView& vertices_t0, const MatrixXdView& vertices_t1, const MatrixXiView& edges,
                  const MatrixXiView& faces, const Scalar minimum_separation_distance, const int max_iterations,
                  const Scalar tolerance, const bool allow_zero_toi,
                  std::vector<std::tuple<int, int, Scalar>>& collisions, const int memory_limit_GB = 0,
                  Context& ctx = Context::default_context())
{
    collisions.clear();
    DeviceMesh mesh(vertices_t0, vertices_t1, edges, faces, ctx);
    std::vector<AABB> vertex_boxes, edge_boxes, face_boxes;
    build_vertex_boxes(vertices_t0, vertices_t1, vertex_boxes, minimum_separation_distance, ctx); // ccd.cu:112
    build_edge_boxes(vertex_boxes, edges, edge_boxes, ctx);
    build_face_boxes(vertex_boxes, faces, face_boxes, ctx);
    const int64_t saved = sccd_get_option(ctx.get(), SCCD_OPT_MEMORY_LIMIT_MB);
    if (memory_limit_GB > 0) ctx.set_option(SCCD_OPT_MEMORY_LIMIT_MB, (int64_t)memory_limit_GB * 1024);
    Scalar toi = 1; // ccd.cu:125
    try {
        BroadPhase broad_phase(ctx);
        broad_phase.build(std::make_shared<DeviceAABBs>(vertex_boxes, ctx), std::make_shared<DeviceAABBs>(face_boxes, ctx));
        narrow_phase<true>(mesh, broad_phase.detect_overlaps(), max_iterations, tolerance, minimum_separation_distance,
                           allow_zero_toi, toi, &collisions);
        broad_phase.build(std::make_shared<DeviceAABBs>(edge_boxes, ctx));
        narrow_phase<false>(mesh, broad_phase.detect_overlaps(), max_iterations, tolerance, minimum_separation_distance,
                            allow_zero_toi, toi, &collisions);
    } catch (...) {
        ctx.set_option(SCCD_OPT_MEMORY_LIMIT_MB, saved);
        throw;
    }
    ctx.set_option(SCCD_OPT_MEMORY_LIMIT_MB, saved);
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
