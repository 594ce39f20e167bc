"""Generates tests/golden/golden.json from the CPU oracle (python tests/golden/make_golden.py).

These vectors pin the oracle against ITSELF across rebuilds/compilers and give the GPU tests a
fixture that does not need the oracle at the full sizes; they do NOT pin the oracle to the
reference (which cannot be built or run in this image: see oracle/sccd_oracle.h).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "scalable-ccd_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

CASES = ["cloth_ball_10k", "cloth_ball_10k_ms", "soup_400", "random_100k", "cloth_ball_small", "folded_cloth_708", "random_1m"]
# BASELINE configs[3]/[4] and configs[2]; ~10 s of CPU each, skipped by the CPU test suite
SLOW_CASES = {"folded_cloth_708", "random_1m"}


def _sha(pairs):
    return hashlib.sha256(np.ascontiguousarray(pairs, dtype=np.int32).tobytes()).hexdigest()


def scene_of(name):
    from sccd import scenes

    if name.startswith("cloth_ball_10k"):
        return scenes.cloth_ball()  # C1/C2 of BASELINE.json
    if name == "cloth_ball_small":
        return scenes.cloth_ball(20, 1, seed=3)
    if name == "soup_400":
        return scenes.triangle_soup(400, seed=11)
    if name == "folded_cloth_708":
        return scenes.folded_cloth(708)
    raise KeyError(name)


def compute_case(orc, name):
    from sccd import scenes

    if name in ("random_100k", "random_1m"):
        b = scenes.random_boxes(100_000 if name == "random_100k" else 1_000_000, seed=42, max_extent=0.027)
        pairs, ax, tests = orc.sort_and_sweep(b, nthreads=8)
        return {"n": int(len(pairs)), "sha256": _sha(pairs), "next_axis": int(ax), "candidate_tests": int(tests)}
    V0, V1, E, F = scene_of(name)
    ms = 1e-3 if name.endswith("_ms") else 0.0
    vb, eb, fb = orc.build_boxes(V0, V1, E, F, ms)
    vf, _, _ = orc.sort_and_sweep(vb, fb, nthreads=8)
    ee, _, _ = orc.sort_and_sweep(eb, nthreads=8)
    out = {
        "nV": int(len(V0)), "nE": int(len(E)), "nF": int(len(F)),
        "n_vf": int(len(vf)), "sha_vf": _sha(vf), "n_ee": int(len(ee)), "sha_ee": _sha(ee),
        "sha_vertex_boxes": hashlib.sha256(vb.tobytes()).hexdigest(),
    }
    for arith, tag in ((0, "strict"), (1, "fma")):
        t_vf, _ = orc.narrow_phase_mt(V0, V1, E, F, vf, True, ms=ms, arith=arith, nthreads=8)
        t_ee, _ = orc.narrow_phase_mt(V0, V1, E, F, ee, False, ms=ms, arith=arith, toi=t_vf, nthreads=8)
        out[f"toi_vf_{tag}"] = float(t_vf).hex()
        out[f"toi_{tag}"] = float(t_ee).hex()
    return out


if __name__ == "__main__":
    import orc

    G = {name: compute_case(orc, name) for name in CASES}
    try:  # (the case that tells the arithmetic contracts apart is made by make_contract_case.py: keep it)
        G["contract_split"] = json.load(open(os.path.join(HERE, "golden.json")))["contract_split"]
    except Exception:
        pass
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(G, f, indent=1, sort_keys=True)
    print(json.dumps(G, indent=1, sort_keys=True))
