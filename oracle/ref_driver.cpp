// oracle/ref_driver.cpp -- test infrastructure.  Calls the REFERENCE's own CPU broad phase (compiled unmodified from
// /root/reference by oracle/ref_build.mk against real Eigen / oneTBB / spdlog) and dumps what it computes, so that
// tests/test_reference_build.py can diff the oracle (and the HIP library) against it:
//   build_vertex_boxes / build_edge_boxes / build_face_boxes   broad_phase/aabb.hpp (aabb.cpp:63-133)
//   sort_and_sweep (one list, two lists)                       broad_phase/sort_and_sweep.hpp (sort_and_sweep.cpp:198-240)
// Nothing of the product includes or links this file.
//
//   ref_driver mesh  <in.bin> <out.bin> [inflation_radius]   in: int32 nV nE nF | f64 V0[nV*3] V1[nV*3] (column-major, Eigen's
//                                                            layout) | int32 E[nE*2] F[nF*3] (column-major)
//       out: int32 nV nE nF | boxes V, E, F (each: f64 min[3] max[3], int32 vertex_ids[3] element_id = the 64-byte sccd_aabb)
//            | int32 axis_vf n_vf | pairs int32[n_vf][2] sorted | int32 axis_ee n_ee | pairs sorted
//   ref_driver boxes <in.bin> <out.bin>                      in: int32 n | n boxes (64 B each, as above)
//       out: int32 axis n_pairs | pairs sorted      (one list, sort axis 0 on entry: tests/test_broad_phase.cpp:46-55)
#include <scalable_ccd/broad_phase/aabb.hpp>
#include <scalable_ccd/broad_phase/sort_and_sweep.hpp>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

using scalable_ccd::AABB;

namespace {
struct Reader {
    std::FILE* f;
    template <class T> void get(T* p, size_t n)
    {
        if (n && std::fread(p, sizeof(T), n, f) != n) {
            std::fprintf(stderr, "ref_driver: short read\n");
            std::exit(3);
        }
    }
};
void put_box(std::FILE* f, const AABB& b)
{
    double g[6];
    for (int k = 0; k < 3; k++) {
        g[k] = (double)b.min[k];
        g[3 + k] = (double)b.max[k];
    }
    const int32_t ids[4] = { (int32_t)b.vertex_ids[0], (int32_t)b.vertex_ids[1], (int32_t)b.vertex_ids[2], (int32_t)b.element_id };
    std::fwrite(g, sizeof g, 1, f);
    std::fwrite(ids, sizeof ids, 1, f);
}
void put_pairs(std::FILE* f, int axis, std::vector<std::pair<int, int>> ov)
{
    std::sort(ov.begin(), ov.end());
    const int32_t head[2] = { (int32_t)axis, (int32_t)ov.size() };
    std::fwrite(head, sizeof head, 1, f);
    for (const auto& p : ov) {
        const int32_t q[2] = { (int32_t)p.first, (int32_t)p.second };
        std::fwrite(q, sizeof q, 1, f);
    }
}
} // namespace

int main(int argc, char** argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: ref_driver mesh|boxes <in.bin> <out.bin> [inflation_radius]\n");
        return 2;
    }
    const bool mesh = std::strcmp(argv[1], "mesh") == 0;
    std::FILE* in = std::fopen(argv[2], "rb");
    std::FILE* out = std::fopen(argv[3], "wb");
    if (!in || !out) {
        std::fprintf(stderr, "ref_driver: cannot open the files\n");
        return 2;
    }
    Reader r { in };
    if (mesh) {
        const double radius = argc > 4 ? std::atof(argv[4]) : 0.0;
        int32_t n[3];
        r.get(n, 3);
        Eigen::MatrixXd V0(n[0], 3), V1(n[0], 3);
        Eigen::MatrixXi E(n[1], 2), F(n[2], 3);
        r.get(V0.data(), (size_t)n[0] * 3);
        r.get(V1.data(), (size_t)n[0] * 3);
        r.get(E.data(), (size_t)n[1] * 2);
        r.get(F.data(), (size_t)n[2] * 3);
        std::vector<AABB> vb, eb, fb;
        scalable_ccd::build_vertex_boxes(V0, V1, vb, radius);
        scalable_ccd::build_edge_boxes(vb, E, eb);
        scalable_ccd::build_face_boxes(vb, F, fb);
        std::fwrite(n, sizeof n, 1, out);
        for (const auto* L : { &vb, &eb, &fb })
            for (const AABB& b : *L) put_box(out, b);
        std::vector<std::pair<int, int>> ov;
        int axis = 0; // tests/test_broad_phase.cpp:46
        scalable_ccd::sort_and_sweep(vb, fb, axis, ov);
        put_pairs(out, axis, ov);
        axis = 0;
        scalable_ccd::sort_and_sweep(eb, axis, ov);
        put_pairs(out, axis, ov);
    } else {
        int32_t n;
        r.get(&n, 1);
        std::vector<AABB> boxes((size_t)n);
        for (int32_t i = 0; i < n; i++) {
            double g[6];
            int32_t ids[4];
            r.get(g, 6);
            r.get(ids, 4);
            scalable_ccd::ArrayMax3 lo(3), hi(3);
            for (int k = 0; k < 3; k++) {
                lo[k] = (scalable_ccd::Scalar)g[k];
                hi[k] = (scalable_ccd::Scalar)g[3 + k];
            }
            boxes[i] = AABB(lo, hi);
            boxes[i].vertex_ids = { { ids[0], ids[1], ids[2] } };
            boxes[i].element_id = ids[3];
        }
        std::vector<std::pair<int, int>> ov;
        int axis = 0;
        scalable_ccd::sort_and_sweep(boxes, axis, ov);
        put_pairs(out, axis, ov);
    }
    std::fclose(in);
    std::fclose(out);
    return 0;
}
