// host reference of tests/gpu_probe/walk_probe.hip (built by the Makefile with g++)
#include "ti_math.hpp"
#define WP_QUAL
#include "walk_probe_ops.h"
#include <vector>
extern "C" void wp_host_generate(int n_seq, int n_ops, Op* ops, Out* href, unsigned* maxdepth)
{
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    *maxdepth = 0;
    for (int s = 0; s < n_seq; s++) {
        int dim_of_depth[96]; // the split dimension of a depth is the same whenever that depth is passed (the walk's premise)
        for (int d = 0; d < 96; d++) dim_of_depth[d] = (int)(rnd() % 3);
        NQDom dom = { 0, 0, 0, 0 };
        NQWalk w = { { 0, 0 }, { 0, 0 }, { 0, 0 } };
        for (int i = 0; i < n_ops; i++) {
            Op o;
            const unsigned r = (unsigned)(rnd() % 100);
            const unsigned depth = nq_depth(dom);
            o.kind = r < 62 ? 0 : (r < 90 ? 1 : 2);
            o.split = dim_of_depth[depth < 96 ? depth : 95];
            o.second = (int)(rnd() & 1);
            ops[(size_t)s * n_ops + i] = o;
            if (o.kind == 0) { if (((dom.d >> (8 * o.split)) & 255u) < 31u) dom = nq_descend(w, dom, o.split, o.second != 0); }
            else if (o.kind == 1) { if (nqb_any(w.pend)) dom = nq_backtrack(w, dom); }
            else { if (nqb_any(w.pend)) (void)nq_donate(w, dom); }
        }
        run_ops(ops + (size_t)s * n_ops, n_ops, NQDom{ 0, 0, 0, 0 }, href + (size_t)s * n_ops);
        for (int i = 0; i < n_ops; i++) { const unsigned d = nq_depth(href[(size_t)s * n_ops + i].dom); if (d > *maxdepth) *maxdepth = d; }
    }
}
