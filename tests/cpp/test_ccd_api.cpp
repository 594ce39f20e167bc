// C++ parity test of the host API mirror (include/scalable_ccd/hip/ccd.hpp), shaped like the
// reference's own Catch2 tests (tests/test_broad_phase.cu:89-121, tests/test_narrow_phase.cu:41-65):
// load a two-frame mesh, build the boxes, run BroadPhase::detect_overlaps for vertex-face and
// edge-edge, run ccd(), and compare with the CPU oracle (oracle/sccd_oracle.h) instead of the
// absent sample-data ground truth.  Exit code 0 = all checks passed.
#include <scalable_ccd/hip/ccd.hpp>

#include "../../oracle/sccd_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <array>
#include <set>

using namespace scalable_ccd::hip;

static int g_fail = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::printf("CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
            g_fail++;                                                            \
        }                                                                        \
    } while (0)

static uint64_t g_state = 12345;
static double urand()
{ // splitmix64
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * 0x1.0p-53;
}

// n x n cloth dropping through a second, tilted cloth: column-major V, E, F like Eigen
static void make_scene(int n, std::vector<double>& V0, std::vector<double>& V1, std::vector<int32_t>& E,
                       std::vector<int32_t>& F, int& nV, int& nE, int& nF)
{
    const int per = n * n;
    nV = 2 * per;
    V0.assign((size_t)3 * nV, 0.0);
    V1.assign((size_t)3 * nV, 0.0);
    std::vector<std::array<int, 3>> faces;
    for (int s = 0; s < 2; s++)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                const int v = s * per + i * n + j;
                const double x = j / double(n - 1), y = i / double(n - 1);
                const double z0 = s == 0 ? 0.5 : 0.2 + 0.15 * x, dz = s == 0 ? -0.5 : 0.0;
                V0[v] = x;
                V0[v + nV] = y;
                V0[v + 2 * nV] = z0;
                V1[v] = x + (urand() - 0.5) * 0.01;
                V1[v + nV] = y + (urand() - 0.5) * 0.01;
                V1[v + 2 * nV] = z0 + dz + (urand() - 0.5) * 0.01;
                if (i + 1 < n && j + 1 < n) {
                    faces.push_back({ v, v + 1, v + n + 1 });
                    faces.push_back({ v, v + n + 1, v + n });
                }
            }
    nF = (int)faces.size();
    F.resize((size_t)3 * nF);
    std::set<std::pair<int, int>> es;
    for (int f = 0; f < nF; f++)
        for (int k = 0; k < 3; k++) {
            F[f + (size_t)k * nF] = faces[f][k];
            const int a = faces[f][k], b = faces[f][(k + 1) % 3];
            es.insert({ std::min(a, b), std::max(a, b) });
        }
    nE = (int)es.size();
    E.resize((size_t)2 * nE);
    int e = 0;
    for (auto& pr : es) {
        E[e] = pr.first;
        E[e + nE] = pr.second;
        e++;
    }
}

static std::vector<std::pair<int, int>> sorted_pairs(const int32_t* p, int64_t n)
{
    std::vector<std::pair<int, int>> v((size_t)n);
    for (int64_t i = 0; i < n; i++) v[(size_t)i] = { p[2 * i], p[2 * i + 1] };
    std::sort(v.begin(), v.end());
    return v;
}

// ---- a driver written against the REFERENCE's interfaces -- the call sequence of partial_ccd<run_vf> / ccd()
// (src/scalable_ccd/cuda/ccd.cu:14-146): MemoryHandler, BroadPhase(memory_handler), threads_per_block, build,
// is_complete / detect_overlaps_partial, narrow_phase<run_vf>(four DeviceMatrix, broad_phase.overlaps(), threads, ...,
// memory_handler, [collisions,] toi).  Only the namespace differs (sccd_ref below is scalable_ccd::cuda there); the
// logger / profiler / cudaDeviceSynchronize lines of the original are infrastructure outside the path and left out.
namespace sccd_ref = scalable_ccd::hip;
namespace as_in_reference {
using namespace sccd_ref;

template <bool run_vf, bool per_query>
void partial_ccd(const DeviceMatrix<Scalar>& d_vertices_t0, const DeviceMatrix<Scalar>& d_vertices_t1,
                 const DeviceMatrix<int>& d_edges, const DeviceMatrix<int>& d_faces,
                 const std::shared_ptr<DeviceAABBs> d_vertex_boxes, const std::shared_ptr<DeviceAABBs> d_edge_boxes,
                 const std::shared_ptr<DeviceAABBs> d_face_boxes, const Scalar min_distance, const int max_iterations,
                 const Scalar tolerance, const bool allow_zero_toi, std::vector<std::tuple<int, int, Scalar>>& collisions,
                 Scalar& toi, const int memory_limit_GB)
{
    constexpr int bp_threads = 32;
    constexpr int np_threads = 1024;
    std::shared_ptr<MemoryHandler> memory_handler = std::make_shared<MemoryHandler>();
    if (memory_limit_GB) memory_handler->memory_limit_GB = memory_limit_GB;
    BroadPhase broad_phase(memory_handler);
    broad_phase.threads_per_block = bp_threads;
    if constexpr (run_vf) broad_phase.build(d_vertex_boxes, d_face_boxes);
    else broad_phase.build(d_edge_boxes);
    while (!broad_phase.is_complete()) {
        broad_phase.detect_overlaps_partial();
        if constexpr (per_query)
            narrow_phase<run_vf>(d_vertices_t0, d_vertices_t1, d_edges, d_faces, broad_phase.overlaps(), np_threads,
                                 max_iterations, tolerance, min_distance, allow_zero_toi, memory_handler, collisions, toi);
        else
            narrow_phase<run_vf>(d_vertices_t0, d_vertices_t1, d_edges, d_faces, broad_phase.overlaps(), np_threads,
                                 max_iterations, tolerance, min_distance, allow_zero_toi, memory_handler, toi);
    }
}

template <bool per_query>
Scalar ccd(const MatrixXdView& vertices_t0, const MatrixXdView& vertices_t1, const MatrixXiView& edges, const MatrixXiView& faces,
           const Scalar min_distance, const int max_iterations, const Scalar tolerance, const bool allow_zero_toi,
           std::vector<std::tuple<int, int, Scalar>>& collisions, const int memory_limit_GB = 0)
{
    const DeviceMatrix<Scalar> d_vertices_t0(vertices_t0);
    const DeviceMatrix<Scalar> d_vertices_t1(vertices_t1);
    const DeviceMatrix<int> d_edges(edges);
    const DeviceMatrix<int> d_faces(faces);
    std::vector<AABB> vertex_boxes, edge_boxes, face_boxes;
    build_vertex_boxes(vertices_t0, vertices_t1, vertex_boxes, min_distance);
    build_edge_boxes(vertex_boxes, edges, edge_boxes);
    build_face_boxes(vertex_boxes, faces, face_boxes);
    const std::shared_ptr<DeviceAABBs> d_vertex_boxes = std::make_shared<DeviceAABBs>(vertex_boxes);
    const std::shared_ptr<DeviceAABBs> d_edge_boxes = std::make_shared<DeviceAABBs>(edge_boxes);
    const std::shared_ptr<DeviceAABBs> d_face_boxes = std::make_shared<DeviceAABBs>(face_boxes);
    Scalar toi = 1;
    partial_ccd<true, per_query>(d_vertices_t0, d_vertices_t1, d_edges, d_faces, d_vertex_boxes, d_edge_boxes, d_face_boxes,
                                 min_distance, max_iterations, tolerance, allow_zero_toi, collisions, toi, memory_limit_GB);
    partial_ccd<false, per_query>(d_vertices_t0, d_vertices_t1, d_edges, d_faces, d_vertex_boxes, d_edge_boxes, d_face_boxes,
                                  min_distance, max_iterations, tolerance, allow_zero_toi, collisions, toi, memory_limit_GB);
    return toi;
}
} // namespace as_in_reference

int main()
{
    std::vector<double> V0, V1;
    std::vector<int32_t> E, F;
    int nV, nE, nF;
    make_scene(24, V0, V1, E, F, nV, nE, nF);
    const MatrixXdView vertices_t0(V0.data(), nV, 3), vertices_t1(V1.data(), nV, 3);
    const MatrixXiView edges(E.data(), nE, 2), faces(F.data(), nF, 3);

    // ---- boxes: bit-identical to the CPU builders (tests/io.cpp:26-38 uses the same three calls)
    std::vector<AABB> vertex_boxes, edge_boxes, face_boxes;
    build_vertex_boxes(vertices_t0, vertices_t1, vertex_boxes);
    build_edge_boxes(vertex_boxes, edges, edge_boxes);
    build_face_boxes(vertex_boxes, faces, face_boxes);
    CHECK((int)vertex_boxes.size() == nV && (int)edge_boxes.size() == nE && (int)face_boxes.size() == nF);
    std::vector<orc_aabb> ovb((size_t)nV), oeb((size_t)nE), ofb((size_t)nF);
    orc_build_vertex_boxes(V0.data(), V1.data(), nV, 0.0, ovb.data());
    orc_build_edge_boxes(ovb.data(), E.data(), nE, oeb.data());
    orc_build_face_boxes(ovb.data(), F.data(), nF, ofb.data());
    static_assert(sizeof(orc_aabb) == sizeof(AABB), "box layouts must agree");
    CHECK(std::memcmp(ovb.data(), vertex_boxes.data(), sizeof(AABB) * (size_t)nV) == 0);
    CHECK(std::memcmp(oeb.data(), edge_boxes.data(), sizeof(AABB) * (size_t)nE) == 0);
    CHECK(std::memcmp(ofb.data(), face_boxes.data(), sizeof(AABB) * (size_t)nF) == 0);

    // ---- broad phase (tests/test_broad_phase.cu:94-104)
    BroadPhase broad_phase;
    bool threw = false;
    try {
        broad_phase.detect_overlaps();
    } catch (const std::runtime_error&) {
        threw = true; // broad_phase.cu:123-126
    }
    CHECK(threw);
    broad_phase.build(std::make_shared<DeviceAABBs>(vertex_boxes), std::make_shared<DeviceAABBs>(face_boxes));
    std::vector<std::pair<int, int>> vf_overlaps = broad_phase.detect_overlaps();
    CHECK(broad_phase.is_complete());
    broad_phase.build(std::make_shared<DeviceAABBs>(edge_boxes));
    std::vector<std::pair<int, int>> ee_overlaps = broad_phase.detect_overlaps();

    int axis = 0;
    int32_t* op = nullptr;
    int64_t on = orc_sort_and_sweep_two_lists(ovb.data(), nV, ofb.data(), nF, &axis, &op, 4);
    auto want_vf = sorted_pairs(op, on);
    orc_free(op);
    axis = 0;
    on = orc_sort_and_sweep(oeb.data(), nE, &axis, &op, 4);
    auto want_ee = sorted_pairs(op, on);
    orc_free(op);
    std::sort(vf_overlaps.begin(), vf_overlaps.end());
    std::sort(ee_overlaps.begin(), ee_overlaps.end());
    CHECK(vf_overlaps == want_vf);
    CHECK(ee_overlaps == want_ee);
    CHECK(!want_vf.empty() && !want_ee.empty());

    // ---- the CPU entry point of the reference (tests/test_broad_phase.cpp:46-55), on the device path:
    // same pairs along every sort axis, and the next axis the CPU code would hand back
    for (int ax_in = 0; ax_in < 3; ax_in++) {
        int sort_axis = ax_in, want_axis = ax_in;
        std::vector<std::pair<int, int>> got;
        sort_and_sweep(vertex_boxes, face_boxes, sort_axis, got);
        std::sort(got.begin(), got.end());
        CHECK(got == want_vf);
        on = orc_sort_and_sweep_two_lists(ovb.data(), nV, ofb.data(), nF, &want_axis, &op, 4);
        orc_free(op);
        CHECK(sort_axis == want_axis);
        sort_axis = want_axis = ax_in;
        sort_and_sweep(edge_boxes, sort_axis, got);
        std::sort(got.begin(), got.end());
        CHECK(got == want_ee);
        on = orc_sort_and_sweep(oeb.data(), nE, &want_axis, &op, 4);
        orc_free(op);
        CHECK(sort_axis == want_axis);
    }
    {
        // the reference's two-step form: sort_along_axis(), then sweep<is_two_lists>() on the sorted (and, for two lists,
        // merged and id-flipped) boxes -- sort_and_sweep.cpp:126-141,143-195,221-240
        int sort_axis = 0, want_axis = 0;
        std::vector<AABB> sorted_e = edge_boxes;
        sort_along_axis(sort_axis, sorted_e);
        CHECK(std::is_sorted(sorted_e.begin(), sorted_e.end(), [](const AABB& a, const AABB& b) { return a.min[0] < b.min[0]; }));
        std::vector<std::pair<int, int>> got;
        sweep<false>(sorted_e, sort_axis, got);
        std::sort(got.begin(), got.end());
        CHECK(got == want_ee);
        on = orc_sort_and_sweep(oeb.data(), nE, &want_axis, &op, 4);
        orc_free(op);
        CHECK(sort_axis == want_axis);
        std::vector<AABB> a = vertex_boxes, b = face_boxes, merged(vertex_boxes.size() + face_boxes.size());
        sort_along_axis(0, a);
        sort_along_axis(0, b);
        for (AABB& box : a) box.element_id = -box.element_id - 1;
        std::merge(a.begin(), a.end(), b.begin(), b.end(), merged.begin(), [](const AABB& x, const AABB& y) { return x.min[0] < y.min[0]; });
        sort_axis = want_axis = 0;
        sweep<true>(merged, sort_axis, got);
        std::sort(got.begin(), got.end());
        CHECK(got == want_vf);
        on = orc_sort_and_sweep_two_lists(ovb.data(), nV, ofb.data(), nF, &want_axis, &op, 4);
        orc_free(op);
        CHECK(sort_axis == want_axis);
    }
    {
        int sort_axis = 0;
        std::vector<std::pair<int, int>> got = { { 1, 2 } };
        sort_and_sweep(std::vector<AABB>(), sort_axis, got);
        CHECK(got.empty() && sort_axis == 0); // sort_and_sweep.cpp:203-207
    }

    // ---- narrow phase + ccd (tests/test_narrow_phase.cu:41-65)
    constexpr bool allow_zero_toi = true;
    constexpr Scalar min_distance = 0;
    constexpr int max_iterations = -1;
    constexpr Scalar tolerance = 1e-6;
    const Scalar toi = ccd(vertices_t0, vertices_t1, edges, faces, min_distance, max_iterations, tolerance, allow_zero_toi);
    double want_toi = 1;
    orc_ccd(V0.data(), V1.data(), nV, E.data(), nE, F.data(), nF, min_distance, max_iterations, tolerance, 1,
            ORC_ARITH_FMA, 4, &want_toi, nullptr, nullptr);
    CHECK(toi == want_toi);
    CHECK(toi < 1 && toi > 0);

    DeviceMesh mesh(vertices_t0, vertices_t1, edges, faces);
    Scalar t2 = 1;
    std::vector<std::tuple<int, int, Scalar>> collisions;
    narrow_phase<true>(mesh, vf_overlaps, max_iterations, tolerance, min_distance, allow_zero_toi, t2, &collisions);
    for (const auto& [i, j, _toi] : collisions) CHECK(t2 <= _toi); // tests/test_narrow_phase.cu:60-62
    narrow_phase<false>(mesh, ee_overlaps, max_iterations, tolerance, min_distance, allow_zero_toi, t2);
    CHECK(t2 == toi);
    {
        // ccd() with the per-query list (the TOI_PER_QUERY build's signature): same TOI, the union of both passes' lists
        std::vector<std::tuple<int, int, Scalar>> all, vf_only;
        const Scalar t3 = ccd(vertices_t0, vertices_t1, edges, faces, min_distance, max_iterations, tolerance, allow_zero_toi, all);
        CHECK(t3 == toi);
        Scalar t4 = 1;
        narrow_phase<true>(mesh, vf_overlaps, max_iterations, tolerance, min_distance, allow_zero_toi, t4, &vf_only);
        std::vector<std::tuple<int, int, Scalar>> ee_only;
        Scalar t5 = 1; // (per query: no pruning by the running minimum, so the seed does not matter)
        narrow_phase<false>(mesh, ee_overlaps, max_iterations, tolerance, min_distance, allow_zero_toi, t5, &ee_only);
        CHECK(all.size() == vf_only.size() + ee_only.size());
        CHECK(!all.empty());
        Scalar mn = 1;
        for (const auto& [i, j, _t] : all) mn = _t < mn ? _t : mn;
        CHECK(mn == toi);
    }

    {
        // the driver written against the reference's interfaces (above): same TOI, same collision records
        std::vector<std::tuple<int, int, Scalar>> none, mine, lib;
        CHECK(as_in_reference::ccd<false>(vertices_t0, vertices_t1, edges, faces, min_distance, max_iterations, tolerance, allow_zero_toi, none) == toi);
        CHECK(none.empty());
        CHECK(as_in_reference::ccd<true>(vertices_t0, vertices_t1, edges, faces, min_distance, max_iterations, tolerance, allow_zero_toi, mine, 1) == toi);
        CHECK(ccd(vertices_t0, vertices_t1, edges, faces, min_distance, max_iterations, tolerance, allow_zero_toi, lib) == toi);
        std::sort(mine.begin(), mine.end());
        std::sort(lib.begin(), lib.end());
        CHECK(mine == lib && !mine.empty());
        // BroadPhase surface of broad_phase.cuh:41-66: overlaps() is the device list of the last partial call
        auto handler = std::make_shared<MemoryHandler>();
        BroadPhase bp2(handler);
        auto d_eb = std::make_shared<DeviceAABBs>(edge_boxes);
        bp2.build(d_eb);
        CHECK(bp2.boxes() == d_eb && bp2.num_boxes() == (size_t)nE);
        const DeviceVector<int2>& ov = bp2.detect_overlaps_partial();
        CHECK(&ov == &bp2.overlaps() && ov.size() == ee_overlaps.size() && handler->real_count == (int)ee_overlaps.size());
        std::vector<int2> hv = ov.to_host();
        std::vector<std::pair<int, int>> got(hv.size());
        for (size_t i = 0; i < hv.size(); i++) got[i] = { hv[i].x, hv[i].y };
        std::sort(got.begin(), got.end());
        CHECK(got == ee_overlaps);
        bp2.clear();
        CHECK(bp2.num_boxes() == 0 && bp2.overlaps().size() == 0);
    }
    {
        // a MemoryHandler is per BroadPhase (broad_phase.cuh:76): its limits apply to that object's calls only -- the shared
        // default context keeps its own options afterwards
        Context& dc = Context::default_context();
        const int64_t lim0 = sccd_get_option(dc.get(), SCCD_OPT_MEMORY_LIMIT_MB), cut0 = sccd_get_option(dc.get(), SCCD_OPT_MAX_OVERLAP_CUTOFF),
                      cap0 = sccd_get_option(dc.get(), SCCD_OPT_OVERLAP_CAPACITY);
        auto handler = std::make_shared<MemoryHandler>();
        handler->memory_limit_GB = 1;
        handler->MAX_OVERLAP_CUTOFF = 4096;
        BroadPhase bp3(handler);
        bp3.build(std::make_shared<DeviceAABBs>(edge_boxes));
        std::vector<std::pair<int, int>> got = bp3.detect_overlaps(); // (in chunks of 4096 rows)
        std::sort(got.begin(), got.end());
        CHECK(got == ee_overlaps);
        CHECK(sccd_get_option(dc.get(), SCCD_OPT_MEMORY_LIMIT_MB) == lim0 && sccd_get_option(dc.get(), SCCD_OPT_MAX_OVERLAP_CUTOFF) == cut0
              && sccd_get_option(dc.get(), SCCD_OPT_OVERLAP_CAPACITY) == cap0);
        // DeviceVector::resize keeps its contents like thrust::device_vector
        std::vector<int> h(1000);
        for (int i = 0; i < 1000; i++) h[(size_t)i] = 7 * i + 1;
        DeviceVector<int> dv(h);
        dv.resize(100000);
        dv.resize(1000);
        CHECK(dv.to_host() == h);
    }

    threw = false;
    try {
        ccd(MatrixXdView(V0.data(), nV, 3), MatrixXdView(V1.data(), nV - 1, 3), edges, faces, 0, -1, 1e-6, true);
    } catch (const std::runtime_error&) {
        threw = true; // ccd.cu:94-95
    }
    CHECK(threw);

    std::printf("test_ccd_api: %d failure(s); toi=%.17g vf=%zu ee=%zu\n", g_fail, toi, vf_overlaps.size(), ee_overlaps.size());
    return g_fail ? 1 : 0;
}
