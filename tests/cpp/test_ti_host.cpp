// Host-side check of the PRODUCTION arithmetic (csrc/ti_math.hpp, Scalar = double): what np_level_k (ti_step) and
// np_walk_k (nq_step on integer domain entries, per-query displacements, reciprocal tolerances, constants one
// coordinate at a time) compute, against the CPU oracle -- per-query constants, single inclusion-function
// evaluations, and whole queries walked depth-first, bit for bit.  Built by tests/test_ti_host.py with
// g++ -ffp-contract=off -mfma (no GPU needed; the kernels around this arithmetic are covered by the -m gpu tests).
#include "sccd_oracle.h"
#include "ti_math.hpp"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static double rnd()
{
    uint64_t z = (g_s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
static int fails = 0;
#define CHECK(c)                                                                    \
    do {                                                                            \
        if (!(c)) {                                                                 \
            if (fails < 20) std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            ++fails;                                                                \
        }                                                                           \
    } while (0)
static bool same(double a, double b) { return std::memcmp(&a, &b, 8) == 0 || (a == b); }

struct Dom {
    double lo[3], hi[3];
};
// level kernels: ti_step on (lo, hi) pairs; INVTOL = the reciprocal shortcut of the split rule
template <bool VF, int ARITH, bool INVTOL> static double walk_ti(const TIQuery& q, double ms, double tol, bool allow_zero, long* checks)
{
    std::vector<Dom> st;
    st.push_back(Dom { { 0, 0, 0 }, { 1, 1, 1 } });
    double toi = 1;
    while (!st.empty()) {
        const Dom d = st.back();
        st.pop_back();
        const TIStep s = ti_step<VF, ARITH, INVTOL>(q, d.lo, d.hi, ms, tol, allow_zero, toi);
        if (s.checked && ++*checks > 1000000) return -1.0;
        if (s.accept && d.lo[0] < toi) toi = d.lo[0];
        if (s.nk == 2) {
            Dom c = d;
            c.lo[s.split] = s.mid;
            st.push_back(c);
        }
        if (s.nk >= 1) {
            Dom c = d;
            c.hi[s.split] = s.mid;
            st.push_back(c);
        }
    }
    return toi;
}
// work-queue kernel: nq_step on integer domain entries, the query holding displacements
template <bool VF, int ARITH> static double walk_nq(const TIQuery& qd, double ms, double tol, bool allow_zero, long* checks)
{
    std::vector<NQDom> st;
    st.push_back(NQDom { 0u, 0u, 0u, 0u });
    double toi = 1;
    while (!st.empty()) {
        const NQDom d = st.back();
        st.pop_back();
        const NQStep s = nq_step<VF, ARITH>(qd, d, ms, tol, allow_zero, toi);
        if (s.checked && ++*checks > 1000000) return -1.0;
        if (s.accept && s.min_t < toi) toi = s.min_t;
        if (s.nk >= 1) { // narrow_walk.inc, section 3
            const unsigned nd = d.d + (1u << (8 * s.split));
            if (((nd >> (8 * s.split)) & 255u) > 31u) return -2.0; // NQ_OVF_INTERVAL: the kernel would hand over to level order
            const unsigned c0 = s.split == 0 ? 2u * d.k0 : d.k0, c1 = s.split == 1 ? 2u * d.k1 : d.k1, c2 = s.split == 2 ? 2u * d.k2 : d.k2;
            if (s.nk == 2) st.push_back(NQDom { c0 + (s.split == 0 ? 1u : 0u), c1 + (s.split == 1 ? 1u : 0u), c2 + (s.split == 2 ? 1u : 0u), nd });
            st.push_back(NQDom { c0, c1, c2, nd });
        }
    }
    return toi;
}

// work-queue kernel as it walks now: no stack -- pending halves are bits of the path (NQWalk); `donate_every` > 0 also
// gives the shallowest pending half away every so many checks and walks it afterwards as a sub-query of its own
// (what work sharing between lanes does)
template <bool VF, int ARITH>
static double walk_stackless(const TIQuery& qd, double ms, double tol, bool allow_zero, long* checks, int donate_every)
{
    NWQuery nq; // the walk kernel's query: tolerance LEVELS instead of tolerances
    std::memcpy(nq.v, qd.v, sizeof nq.v);
    for (int k = 0; k < 3; k++) {
        nq.err[k] = qd.err[k];
        nq.inv_tol[k] = qd.inv_tol[k];
    }
    nq.dlev = nw_tol_levels(qd.tol, qd.inv_ok);
    if (nq.dlev >> 24) return -3.0; // inexact reciprocals: level-synchronous path
    std::vector<NQDom> roots;
    roots.push_back(NQDom { 0u, 0u, 0u, 0u });
    double toi = 1;
    long since = 0;
    while (!roots.empty()) {
        NQDom d = roots.back();
        roots.pop_back();
        NQWalk w = { { 0, 0 }, { 0, 0 }, { 0, 0 } };
        bool live = true;
        while (live) {
            const NQStep s = nw_step<VF, ARITH>(nq, d, ms, tol, allow_zero, toi);
            if (s.checked && ++*checks > 1000000) return -1.0;
            if (s.accept && s.min_t < toi) toi = s.min_t;
            if (s.nk >= 1) {
                if ((((d.d + (1u << (8 * s.split))) >> (8 * s.split)) & 255u) > 31u) return -2.0;
                if (nq_depth(d) < 32u) {
                    NQWalk w2 = w, w3 = w;
                    const NQDom a = nq_descend32(w2, d, s.split, s.nk == 2), b = nq_descend(w3, d, s.split, s.nk == 2);
                    CHECK(std::memcmp(&a, &b, sizeof a) == 0 && std::memcmp(&w2, &w3, sizeof w2) == 0);
                }
                d = nq_descend(w, d, s.split, s.nk == 2);
            } else if (nqb_any(w.pend)) {
                if (nq_depth(d) <= 32u) { // the kernel's shallow form must agree with the general one
                    NQWalk w2 = w;
                    const NQDom a = nq_backtrack32(w2, d);
                    NQWalk w3 = w;
                    const NQDom b = nq_backtrack(w3, d);
                    CHECK(std::memcmp(&a, &b, sizeof a) == 0 && std::memcmp(&w2, &w3, sizeof w2) == 0);
                }
                d = nq_backtrack(w, d);
            } else {
                live = false;
            }
            if (live && donate_every > 0 && ++since >= donate_every && nqb_any(w.pend)) {
                since = 0;
                roots.push_back(nq_donate(w, d));
            }
        }
    }
    return toi;
}

template <bool VF, int ARITH>
static void run_case(int n_queries, double scale, double ms, bool allow_zero, long* hits, long* skipped, long* handed_over)
{
    const double tol = 1e-6;
    for (int it = 0; it < n_queries; it++) {
        double V0[12], V1[12]; // column-major 4 x 3
        auto set = [&](double* M, int i, double x, double y, double z) {
            M[i] = x * scale;
            M[i + 4] = y * scale;
            M[i + 8] = z * scale;
        };
        if (VF) {
            const double x = rnd(), y = rnd() * (1 - x);
            set(V0, 0, x, y, 0.2 + 0.5 * rnd());
            set(V1, 0, x + 0.1 * (rnd() - 0.5), y + 0.1 * (rnd() - 0.5), -0.2 - 0.5 * rnd());
            for (int j = 1; j < 4; j++) {
                const double px = (j == 2) ? 1.0 : 0.0, py = (j == 3) ? 1.0 : 0.0;
                set(V0, j, px + 0.05 * rnd(), py + 0.05 * rnd(), 0.05 * (rnd() - 0.5));
                set(V1, j, px + 0.05 * rnd(), py + 0.05 * rnd(), 0.05 * (rnd() - 0.5));
            }
        } else {
            set(V0, 0, rnd(), 0.0, 0.2 + 0.5 * rnd());
            set(V0, 1, rnd(), 1.0, 0.2 + 0.5 * rnd());
            set(V1, 0, rnd(), 0.0, -0.2 - 0.5 * rnd());
            set(V1, 1, rnd(), 1.0, -0.2 - 0.5 * rnd());
            set(V0, 2, 0.0, rnd(), 0.05 * (rnd() - 0.5));
            set(V0, 3, 1.0, rnd(), 0.05 * (rnd() - 0.5));
            set(V1, 2, 0.0, rnd(), 0.05 * (rnd() - 0.5));
            set(V1, 3, 1.0, rnd(), 0.05 * (rnd() - 0.5));
        }
        TIQuery q;
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 3; k++) {
                q.v[j][k] = V0[j + 4 * k];
                q.v[j + 4][k] = V1[j + 4 * k];
            }
        ti_tolerance<VF>(q.v, tol, q.tol);
        ti_error<VF>(q.v, ms > 0, q.err);
        ti_prepare_inv_tol(q);
        double otol[3], oerr[3];
        orc_query_constants(&q.v[0][0], VF ? 1 : 0, ms > 0, tol, otol, oerr);
        // the work-queue kernel derives the same constants one coordinate at a time (narrow_walk.inc, constants_of)
        double m[3] = { 0, 0, 0 }, er[3], tl[3];
        for (int k = 0; k < 3; k++) {
            double x[8];
            for (int j = 0; j < 8; j++) x[j] = q.v[j][k];
            ti_tolerance_dim<VF>(x, m);
            er[k] = ti_error_dim<VF>(x, ms > 0);
        }
        ti_tolerance_finish<VF>(m, tol, tl);
        for (int k = 0; k < 3; k++) {
            CHECK(same(q.tol[k], otol[k]) && same(tl[k], otol[k]));
            CHECK(same(q.err[k], oerr[k]) && same(er[k], oerr[k]));
        }
        TIQuery qd = q; // the queue kernel's query: displacements instead of end positions
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 3; k++) qd.v[j + 4][k] = q.v[j + 4][k] - q.v[j][k];
        for (int r = 0; r < 24; r++) { // single evaluations on random dyadic sub-domains
            double lo[3], hi[3], dom[6];
            for (int k = 0; k < 3; k++) {
                const int d = (int)(rnd() * 8);
                const int kk = (int)(rnd() * (1 << d));
                lo[k] = (double)kk / (double)(1 << d);
                hi[k] = (double)(kk + 1) / (double)(1 << d);
                dom[2 * k] = lo[k];
                dom[2 * k + 1] = hi[k];
            }
            double tt = 0, td = 0, ott = 0;
            bool bi = false, bd = false;
            int obi = 0;
            const bool a = ti_inclusion<VF, ARITH>(q.v, lo, hi, q.err, ms, tt, bi);
            const bool ad = ti_inclusion<VF, ARITH, true>(qd.v, lo, hi, q.err, ms, td, bd);
            const int b = orc_origin_in_inclusion_function(&q.v[0][0], dom, q.err, ms, VF ? 1 : 0, ARITH, &ott, &obi);
            CHECK(a == (b != 0) && ad == a);
            if (a && b) {
                CHECK(same(tt, ott) && same(td, ott));
                CHECK(bi == (obi != 0) && bd == bi);
            }
            // the walk kernel's evaluation from the extreme operands (monotone rounding): same verdict, same width
            double tm = 0;
            bool bm = false;
            const bool am = ti_inclusion_mm<VF, ARITH>(qd.v, lo, hi, q.err, ms, tm, bm);
            CHECK(am == a);
            if (am && a) CHECK(same(tm, ott) && bm == bi);
        }
        const int32_t E[4] = { 0, 2, 1, 3 }; // column-major 2 x 2: edges (0,1) and (2,3)
        const int32_t F[3] = { 1, 2, 3 };
        const int32_t pair_vf[2] = { 0, 0 }, pair_ee[2] = { 0, 1 };
        long c1 = 0, c2 = 0, c3 = 0;
        const double got1 = walk_ti<VF, ARITH, false>(q, ms, tol, allow_zero, &c1);
        if (got1 < 0) {
            ++*skipped;
            continue;
        }
        const double got2 = walk_ti<VF, ARITH, true>(q, ms, tol, allow_zero, &c2);
        const double got3 = walk_nq<VF, ARITH>(qd, ms, tol, allow_zero, &c3);
        double want = 1;
        orc_np_stats st;
        const int rc = orc_narrow_phase(V0, V1, 4, E, 2, F, 1, VF ? pair_vf : pair_ee, 1, VF ? 1 : 0, ms, -1, tol,
                                        allow_zero ? 1 : 0, ARITH, &want, nullptr, &st);
        if (rc != 0) {
            ++*skipped;
            continue;
        }
        CHECK(same(got1, want));
        CHECK(same(got2, want) && c2 == c1);
        if (got3 == -2.0) ++*handed_over; // deeper than 2^-31: np_walk_k hands the call to the level-synchronous kernel
        else CHECK(same(got3, want) && c3 == c1); // same traversal, same number of checks
        // the stackless walk visits the same domains in the same order as the explicit stack ...
        long c4 = 0, c5 = 0;
        const double got4 = walk_stackless<VF, ARITH>(qd, ms, tol, allow_zero, &c4, 0);
        if (got4 != -3.0) CHECK(same(got4, got3) && c4 == c3);
        // ... and with sub-trees given away every few checks (work sharing) still finds the same time of impact
        const double got5 = walk_stackless<VF, ARITH>(qd, ms, tol, allow_zero, &c5, 3 + it % 5);
        if (got3 != -2.0 && got5 >= 0) CHECK(same(got5, want));
        if (want < 1) ++*hits;
    }
}

int main()
{
    long hits = 0, skipped = 0, handed_over = 0;
    const int n = 12;
    for (double scale : { 1.0, 37.5, 0.01, 1234.5 })
        for (double ms : { 0.0, 1e-3 })
            for (bool az : { true, false }) {
                run_case<true, 0>(n, scale, ms * scale, az, &hits, &skipped, &handed_over);
                run_case<true, 1>(n, scale, ms * scale, az, &hits, &skipped, &handed_over);
                run_case<false, 0>(n, scale, ms * scale, az, &hits, &skipped, &handed_over);
                run_case<false, 1>(n, scale, ms * scale, az, &hits, &skipped, &handed_over);
            }
    const int total = 4 * 2 * 2 * 4 * n;
    std::printf("test_ti_host: %d failure(s), %ld colliding queries of %d, %ld skipped (> 1e6 checks), %ld beyond 2^-31\n", fails,
                hits, total, skipped, handed_over);
    return (fails == 0 && hits > total / 4 && skipped < total / 5) ? 0 : 1;
}
