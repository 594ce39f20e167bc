#!/usr/bin/env python3
"""PCIe-inclusive rate: ccd() from HOST matrices (upload + pack + everything), C4 workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-ccd_amd"))
import numpy as np
import sccd
from sccd import scenes

V0, V1, E, F = scenes.folded_cloth(708, seed=7)
V0f, V1f = np.asfortranarray(V0), np.asfortranarray(V1)
Ef, Ff = np.asfortranarray(E.astype(np.int32)), np.asfortranarray(F.astype(np.int32))
ctx = sccd.default_context()
best = 1e9
for rep in range(6):
    t0 = time.perf_counter()
    toi = sccd.ccd(V0f, V1f, Ef, Ff, 0.0, -1, 1e-6, True, ctx=ctx)
    dt = time.perf_counter() - t0
    if rep:
        best = min(best, dt)
print(f"host-matrix ccd(): {best*1e3:.3f} ms per call (toi {toi!r}); bytes uploaded {V0f.nbytes*2 + Ef.nbytes + Ff.nbytes}")
