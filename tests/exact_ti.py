"""An independent, EXACT evaluator of the continuous-collision function -- a second opinion on the oracle that is not
derived from the reference's root_finder.cu.  Written from the mathematical definition only (Wang et al., "A Large
Scale Benchmark and an Inclusion-Based Algorithm for Continuous Collision Detection", 2021, sections 3-4):

  vertex-face   F(t, u, v) = p(t) - ((1 - u - v) a(t) + u b(t) + v c(t)),   u, v >= 0, u + v <= 1
  edge-edge     F(t, u, v) = ((1 - u) p(t) + u q(t)) - ((1 - v) a(t) + v b(t)),   u, v in [0, 1]

with every vertex moving linearly, x(t) = (1 - t) x0 + t x1, t in [0, 1].  F is linear in each of t, u, v, so its
range over a box is spanned by its eight corner values.  Everything is computed in rational arithmetic (Python ints /
fractions.Fraction): no rounding, no error filter, no tolerance heuristics.

TEST INFRASTRUCTURE ONLY (imported by tests/test_exact_second_opinion.py)."""
from fractions import Fraction as Fr


def _fr(x):
    return Fr(float(x))  # exact: every double is a dyadic rational


def corner_values(q, is_vf, box):
    """q: 8 x 3 doubles (four vertices at t = 0, then at t = 1, in the order the pair is assembled: vertex-face
    (p, a, b, c), edge-edge (p, q, a, b)); box = ((t0, t1), (u0, u1), (v0, v1)) of Fractions.
    Returns per axis the exact (min, max) over the eight corners."""
    X0 = [[_fr(q[i][k]) for k in range(3)] for i in range(4)]
    X1 = [[_fr(q[i + 4][k]) for k in range(3)] for i in range(4)]
    out = []
    for k in range(3):
        vals = []
        for t in box[0]:
            x = [(1 - t) * X0[i][k] + t * X1[i][k] for i in range(4)]
            for u in box[1]:
                for v in box[2]:
                    if is_vf:
                        vals.append(x[0] - ((1 - u - v) * x[1] + u * x[2] + v * x[3]))
                    else:
                        vals.append(((1 - u) * x[0] + u * x[1]) - ((1 - v) * x[2] + v * x[3]))
        out.append((min(vals), max(vals)))
    return out


def may_contain_root(q, is_vf, box, ms=Fr(0)):
    """False only if the origin is PROVABLY farther than ms (L-infinity) from F(box)."""
    for lo, hi in corner_values(q, is_vf, box):
        if lo > ms or hi < -ms:
            return False
    return True


def earliest_unexcluded(q, is_vf, levels, t_limit=Fr(1)):
    """Exact bisection, earliest time first: the lower time bound of the first box of side 2^-levels (in every
    dimension it was split in) that cannot be excluded, or None if [0, t_limit] x domain is root free.  The true first
    time of impact t* (if any) satisfies  result <= t*  -- rigorously: every box before it was excluded exactly."""
    one = Fr(1)
    stack = [((Fr(0), one), (Fr(0), one), (Fr(0), one))]
    best = None
    while stack:
        box = stack.pop()
        (t0, t1), (u0, u1), (v0, v1) = box
        if t0 >= t_limit or (best is not None and t0 >= best):
            continue
        if is_vf and u0 + v0 > 1:
            continue  # outside the triangle
        if not may_contain_root(q, is_vf, box):
            continue
        w = (t1 - t0, u1 - u0, v1 - v0)
        small = Fr(1, 2 ** levels)
        if all(x <= small for x in w):
            best = t0 if best is None else min(best, t0)
            continue
        k = max(range(3), key=lambda i: w[i])  # the widest side (ties: time first)
        lo, hi = box[k]
        mid = (lo + hi) / 2
        first = list(box)
        second = list(box)
        first[k] = (lo, mid)
        second[k] = (mid, hi)
        # depth first, the earlier half on top of the stack
        stack.append(tuple(second))
        stack.append(tuple(first))
    return best
