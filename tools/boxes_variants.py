#!/usr/bin/env python3
"""Broad phase on variants of the 1M-box workload (SURVEY 8d asks for a cloth-like one too):
isotropic, flat in z (extent x 0.01), and clustered.  Prints pairs, candidates and time per build+detect."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-ccd_amd"))
import numpy as np
import sccd
from sccd import scenes

ctx = sccd.default_context()
n = 1_000_000
base = scenes.random_boxes(n, seed=42, max_extent=0.027)
variants = {"isotropic": base.copy()}
flat = base.copy()
c = (flat["min"][:, 2] + flat["max"][:, 2]) / 2
h = (flat["max"][:, 2] - flat["min"][:, 2]) / 2 * 0.01
flat["min"][:, 2], flat["max"][:, 2] = c - h, c + h  # cloth-like: boxes 100x thinner in z, centres unchanged
variants["thin in z (extent x 0.01)"] = flat
slab = base.copy()
slab["min"][:, 2] *= 0.01
slab["max"][:, 2] *= 0.01  # everything squeezed into one z-layer (same overlaps as isotropic: a check of the grid)
variants["squeezed slab (z x 0.01)"] = slab
cl = base.copy()
rng = np.random.default_rng(1)
centres = rng.random((64, 3))
which = rng.integers(0, 64, n)
ext = cl["max"] - cl["min"]
ctr = centres[which] + rng.normal(0, 0.02, (n, 3))
cl["min"], cl["max"] = ctr - ext / 2, ctr + ext / 2
variants["64 clusters (sigma 0.02)"] = cl
for name, b in variants.items():
    d = sccd.DeviceAABBs(b, ctx)
    bp = sccd.BroadPhase(ctx)
    best = 1e9
    for rep in range(4):
        ctx.synchronize()
        t0 = time.perf_counter()
        bp.build(d)
        pairs = bp.detect_overlaps_partial()[1]
        ctx.synchronize()
        if rep:
            best = min(best, time.perf_counter() - t0)
    print(f"{name:28s} pairs {pairs:>11d}  candidates {bp.candidates():>12d}  {best*1e3:7.3f} ms", flush=True)
