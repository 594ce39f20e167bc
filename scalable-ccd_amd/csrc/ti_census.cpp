// ti_census.cpp -- ONE query's bisection in the reference's level order, on the host (plain C++, built with g++
// -ffp-contract=off: the same ti_math.hpp the kernels compile, operation for operation).
//
// Why the library needs it: a check limit (max_iter >= 0, what the IPC Toolkit passes: 10^7) is defined in the
// reference's LEVEL ORDER -- a query's domains are dropped once the launches have popped more than max_iter of them
// (root_finder.cu:287-305) -- while the fast kernel walks depth first.  narrow.hip therefore runs the fast kernel WITHOUT
// the limit, which gives T* = the earliest time of impact over all accepted domains, and then proves that the limit would
// not have changed that answer: every accept of the limited level-order run is an accept of the full bisection tree, so
// its result is >= T*; and it equals T* as soon as ONE domain with t = T* is reached, i.e. as soon as the query q* that
// holds such a domain is not cut off before it.  In the real run q* is pruned by the running TOI of ALL queries, which is
// never later than the TOI q* would reach alone -- alone it pops at least as many domains, level by level.  So: q* alone,
// in level order, WITH the limit (this file), reaching T*  ==>  the limited run of the whole call returns T*.  One query,
// some thousand checks: microseconds on a host core.  If the certificate fails (small limits), the call is redone on the
// level-synchronous kernels as before.
#include "ti_math.hpp"

#include <vector>

namespace {
struct Dom {
    double lo[3], hi[3];
};

template <bool VF, int ARITH>
double level_order(const double v[8][3], double ms, double tol, int max_iter, bool allow_zero_toi, double toi_init,
                   long long max_live, bool* gave_up)
{
    TIQuery q;
    for (int a = 0; a < 8; a++)
        for (int k = 0; k < 3; k++) q.v[a][k] = v[a][k];
    ti_tolerance<VF>(q.v, tol, q.tol);   // root_finder.cu:48-88
    ti_error<VF>(q.v, ms > 0, q.err);    // :90-135 (use_ms = ms > 0, narrow_phase.cu:128)
    std::vector<Dom> cur, nxt;
    cur.push_back(Dom { { 0, 0, 0 }, { 1, 1, 1 } });
    double toi = toi_init;
    long long count = 0; // the query's nbr_checks
    *gave_up = false;
    while (!cur.empty()) { // one launch per level, root_finder.cu:431-447
        // every thread of a launch reads the query's counter and the TOI as of the START of the level (the
        // serialisation np_level_k follows with a check limit: narrow.hip, LvlSnap)
        const double toi_level = toi;
        const long long before = count;
        nxt.clear();
        for (const Dom& d : cur) {
            ++count;                                      // :289
            if (d.lo[0] >= toi_level) continue;           // :295
            if (max_iter >= 0 && before > max_iter) continue; // :303
            const TIStep s = ti_step<VF, ARITH>(q, d.lo, d.hi, ms, tol, allow_zero_toi, toi_level);
            if (s.accept && d.lo[0] < toi) toi = d.lo[0];
            if (s.nk >= 1) {
                Dom c = d;
                c.hi[s.split] = s.mid;
                nxt.push_back(c);
                if (s.nk == 2) {
                    c = d;
                    c.lo[s.split] = s.mid;
                    nxt.push_back(c);
                }
            }
        }
        if ((long long)nxt.size() > max_live) {
            *gave_up = true;
            return toi;
        }
        cur.swap(nxt);
    }
    return toi;
}
} // namespace

// the earliest time of impact of ONE query bisected alone in level order with the check limit; *gave_up: a level held
// more than max_live domains (no answer)
double ti_census_level_order(const double v[8][3], int is_vf, int arith, double ms, double tol, int max_iter, int allow_zero_toi,
                             double toi_init, long long max_live, bool* gave_up)
{
    if (is_vf) {
        return arith ? level_order<true, 1>(v, ms, tol, max_iter, allow_zero_toi != 0, toi_init, max_live, gave_up)
                     : level_order<true, 0>(v, ms, tol, max_iter, allow_zero_toi != 0, toi_init, max_live, gave_up);
    }
    return arith ? level_order<false, 1>(v, ms, tol, max_iter, allow_zero_toi != 0, toi_init, max_live, gave_up)
                 : level_order<false, 0>(v, ms, tol, max_iter, allow_zero_toi != 0, toi_init, max_live, gave_up);
}
