"""A second opinion on the oracle that does not come from the same reading of root_finder.cu: tests/exact_ti.py
evaluates the collision function exactly (rational arithmetic, from the paper's definition).  Checked here:

 1. every verdict of the oracle's inclusion function is SOUND against the exact corner values -- a rejected box is
    exactly separated from the origin by more than the minimum separation, an accepted "box inside" box lies exactly
    within (minimum separation + twice the floating-point error bound) -- and COMPLETE: a box exactly separated by
    more than twice the error bound is rejected;
 2. the oracle's time of impact brackets the exact one: no root exists before it (rigorous, by exact exclusion of
    [0, toi) down to a fixed resolution), and an un-excludable box exists right behind it.

(The oracle stays parity-UNPINNED with respect to the reference's binaries -- DESIGN.md section 3 -- this pins it to
the mathematics.)"""
from fractions import Fraction as Fr

import numpy as np
import pytest

from exact_ti import corner_values, earliest_unexcluded, may_contain_root


def _rng_query(rng, is_vf, hit):
    """a query assembled like narrow_phase.cu:41-67: 4 vertices at t = 0 then at t = 1"""
    q = np.zeros((8, 3))
    if is_vf:
        tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], float) + rng.uniform(-0.05, 0.05, (3, 3))
        x, y = rng.uniform(0.1, 0.4, 2)
        p0 = np.array([x, y, rng.uniform(0.2, 0.6)])
        p1 = np.array([x + rng.uniform(-0.05, 0.05), y + rng.uniform(-0.05, 0.05), -rng.uniform(0.2, 0.6) if hit else rng.uniform(0.1, 0.3)])
        q[0], q[1:4] = p0, tri
        q[4], q[5:8] = p1, tri + rng.uniform(-0.03, 0.03, (3, 3))
    else:
        a0 = np.array([rng.uniform(0.2, 0.8), 0.0, rng.uniform(0.2, 0.6)])
        a1 = np.array([rng.uniform(0.2, 0.8), 1.0, rng.uniform(0.2, 0.6)])
        b0 = np.array([0.0, rng.uniform(0.2, 0.8), 0.0]) + rng.uniform(-0.02, 0.02, 3)
        b1 = np.array([1.0, rng.uniform(0.2, 0.8), 0.0]) + rng.uniform(-0.02, 0.02, 3)
        dz = -rng.uniform(0.3, 0.9) if hit else rng.uniform(0.0, 0.2)
        q[0], q[1], q[2], q[3] = a0, a1, b0, b1
        q[4], q[5] = a0 + [0, 0, dz], a1 + [0, 0, dz]
        q[6], q[7] = b0 + rng.uniform(-0.02, 0.02, 3), b1 + rng.uniform(-0.02, 0.02, 3)
    return q


def _dyadic_box(rng):
    box = []
    coarse = rng.random() < 0.6  # large boxes mostly pass the inclusion test, small ones mostly fail it
    for _ in range(3):
        d = int(rng.integers(0, 3)) if coarse else int(rng.integers(0, 9))
        k = int(rng.integers(0, 2 ** d))
        box.append((Fr(k, 2 ** d), Fr(k + 1, 2 ** d)))
    return tuple(box)


@pytest.mark.parametrize("is_vf", [True, False])
@pytest.mark.parametrize("ms", [0.0, 1e-3])
def test_inclusion_verdicts_are_sound_and_complete_in_exact_arithmetic(orc, is_vf, ms):
    rng = np.random.default_rng(77 + int(is_vf) + int(ms > 0) * 2)
    n_rej = n_in = n_pass = 0
    for it in range(250):
        q = _rng_query(rng, is_vf, hit=bool(it % 2))
        if it % 5 == 0:
            q *= 37.5  # larger coordinates, larger error bound
        tol3, err3 = orc.query_constants(q.reshape(-1), is_vf, ms > 0, 1e-6)
        one = (Fr(0), Fr(1))
        for b in range(7):
            box = (one, one, one) if b == 0 else _dyadic_box(rng)  # (the whole domain: passes for every colliding query)
            dom6 = [float(x) for lohi in box for x in lohi]
            passed, true_tol, box_in = orc.inclusion(q.reshape(-1), dom6, err3, ms, is_vf)
            exact = corner_values(q, is_vf, box)
            msf = Fr(float(ms))
            e = [Fr(float(x)) for x in err3]
            separated = any(lo > msf or hi < -msf for lo, hi in exact)
            far = any(lo - msf > 2 * e[k] or hi + msf < -2 * e[k] for k, (lo, hi) in enumerate(exact))
            if not passed:  # sound: really no root within ms of the box
                assert separated, (it, box)
                n_rej += 1
            else:
                assert not far, (it, box)  # complete: nothing clearly separated slips through
                n_pass += 1
                if box_in:  # the whole image lies inside the epsilon box, up to the error bound
                    assert all(lo >= -(msf + 2 * e[k]) and hi <= msf + 2 * e[k] for k, (lo, hi) in enumerate(exact))
                    n_in += 1
                # the reported width is the exact one up to the error bound
                width = max(hi - lo for lo, hi in exact)
                assert abs(Fr(float(true_tol)) - width) <= 2 * max(e) + Fr(1, 10**12)
    assert n_rej > 100 and n_pass > 100, (n_rej, n_pass, n_in)


@pytest.mark.parametrize("is_vf", [True, False])
def test_time_of_impact_brackets_the_exact_one(orc, is_vf):
    """no root before the oracle's TOI (rigorous), and something un-excludable right behind it"""
    rng = np.random.default_rng(5 + int(is_vf))
    E = np.array([[0, 1], [2, 3]], np.int32)
    F = np.array([[1, 2, 3]], np.int32)
    pair = np.array([[0, 0]] if is_vf else [[0, 1]], np.int32)
    levels = 10
    res = Fr(1, 2 ** levels)
    hits = 0
    for it in range(24):
        q = _rng_query(rng, is_vf, hit=(it % 4 != 3))
        V0, V1 = q[:4].copy(), q[4:].copy()
        toi, _, _ = orc.narrow_phase(V0, V1, E, F, pair, is_vf, 0.0, -1, 1e-6, True)
        exact = earliest_unexcluded(q, is_vf, levels)
        if toi >= 1.0:
            # the oracle found nothing: the exact walk may still hold a box it cannot exclude at this coarse
            # resolution, but never a PROVEN miss turned into a hit
            continue
        hits += 1
        t = Fr(float(toi))
        # (a) rigorous: [0, toi) is root free -- the exact walk excludes everything before toi down to its resolution
        # (a box of side 2^-levels in u and v has an image a few times 2^-levels wide, which the moving element
        # crosses in a few times 2^-levels of time: hence the factor)
        assert exact is not None and exact >= t - 6 * res, (it, float(exact) if exact is not None else None, toi)
        # (b) the oracle is not early by more than the exact resolution plus its own tolerance
        assert exact <= t + 2 * res, (it, float(exact), toi)
        # (c) a box that starts at the oracle's TOI cannot be excluded exactly: the impact is really there
        assert earliest_unexcluded(q, is_vf, levels, t_limit=t + 2 * res) is not None
    assert hits >= 12
