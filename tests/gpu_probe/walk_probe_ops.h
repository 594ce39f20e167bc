// shared by walk_probe.hip (device) and walk_probe_host.cpp (host reference)
#pragma once
struct Op { int kind, split, second; };
struct Out { NQDom dom; NQWalk w; };
WP_QUAL inline void run_ops(const Op* ops, int n_ops, NQDom dom, Out* out)
{
    NQWalk w = { { 0, 0 }, { 0, 0 }, { 0, 0 } };
    for (int i = 0; i < n_ops; i++) {
        const Op o = ops[i];
        if (o.kind == 0) { // descend (respecting the level limit)
            const unsigned lev = (dom.d >> (8 * o.split)) & 255u;
            if (lev < 31u) dom = nq_descend(w, dom, o.split, o.second != 0);
        } else if (o.kind == 1) {
            if (nqb_any(w.pend)) dom = nq_backtrack(w, dom);
        } else {
            if (nqb_any(w.pend)) (void)nq_donate(w, dom);
        }
        out[i].dom = dom;
        out[i].w = w;
    }
}
