"""Finds a query on which the two arithmetic contracts of the narrow phase give DIFFERENT times of impact, and writes it into
tests/golden/golden.json as case "contract_split" (python tests/golden/make_contract_case.py).

Why: min-TOI values of whole scenes are dyadic rationals decided far from any rounding boundary, so `toi_fma == toi_strict`
in every other golden case -- a GPU build whose fused path were silently disabled would pass them all.  Here the minimum
separation `ms` is placed (by bisection on its bits) exactly where the strict evaluation of root_finder.cu:137-198 still
keeps a domain that the fused evaluation (one rounding per a*b+c, what nvcc's -fmad=true does to the reference) already
rejects, or the other way round.  Inputs and both expected values are committed; the oracle made them (oracle/np_core.inc).
"""
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "scalable-ccd_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def bits(x):
    return struct.unpack("<q", struct.pack("<d", x))[0]


def from_bits(b):
    return struct.unpack("<d", struct.pack("<q", b))[0]


def find(orc, seed):
    """a vertex flying past a moving triangle; returns (V0, V1, E, F, ms, toi_strict, toi_fma) or None"""
    rng = np.random.default_rng(seed)
    tri0 = rng.uniform(-1, 1, (3, 3)) * np.array([1, 1, 0.05])
    tri1 = tri0 + rng.uniform(-0.2, 0.2, (3, 3))
    c = tri0.mean(axis=0)
    p0 = c + np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(0.3, 0.6)])
    p1 = c + np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(0.02, 0.2)])  # stops short of the face
    V0 = np.vstack([p0, tri0])
    V1 = np.vstack([p1, tri1])
    F = np.array([[1, 2, 3]], dtype=np.int32)
    E = np.array([[1, 2], [2, 3], [1, 3]], dtype=np.int32)
    pairs = np.array([[0, 0]], dtype=np.int32)

    def toi(ms, arith):
        t, _ = orc.narrow_phase(V0, V1, E, F, pairs, True, ms=ms, arith=arith)[:2]
        return t

    lo, hi = 1e-4, 1.0  # no contact at lo, contact at hi?
    if toi(lo, 0) < 1 or toi(hi, 0) >= 1:
        return None
    # the smallest ms (by bits) at which each contract reports a contact
    def threshold(arith):
        a, b = bits(lo), bits(hi)
        while b - a > 1:
            mid = (a + b) // 2
            if toi(from_bits(mid), arith) < 1:
                b = mid
            else:
                a = mid
        return b

    ts, tf = threshold(0), threshold(1)
    if ts == tf:
        return None
    ms = from_bits(min(ts, tf))  # one contract sees the contact at this ms, the other not yet
    a, b = toi(ms, 0), toi(ms, 1)
    if a == b:
        return None
    return V0, V1, E, F, ms, a, b


if __name__ == "__main__":
    import orc

    for seed in range(1, 2000):
        r = find(orc, seed)
        if r is None:
            continue
        V0, V1, E, F, ms, a, b = r
        # the full driver must see the same (the pair survives the broad phase: boxes are inflated by ms)
        ca = orc.ccd(V0, V1, E, F, ms, -1, 1e-6, True, arith=0)[0]
        cb = orc.ccd(V0, V1, E, F, ms, -1, 1e-6, True, arith=1)[0]
        if ca == cb:
            continue
        case = {
            "seed": seed, "ms": float(ms).hex(), "V0": [[float(x).hex() for x in row] for row in V0],
            "V1": [[float(x).hex() for x in row] for row in V1], "E": E.tolist(), "F": F.tolist(),
            "toi_strict": float(ca).hex(), "toi_fma": float(cb).hex(),
        }
        path = os.path.join(HERE, "golden.json")
        G = json.load(open(path))
        G["contract_split"] = case
        json.dump(G, open(path, "w"), indent=1, sort_keys=True)
        print("seed", seed, "ms", ms, "toi strict", ca, "fma", cb)
        break
    else:
        raise SystemExit("no case found")
