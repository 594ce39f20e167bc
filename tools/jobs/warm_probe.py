"""How long does a fresh box need before the step time settles?  ms per step in windows of 0.25 s from process start."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "scalable-ccd_amd"))
import numpy as np
import torch
import sccd
from sccd import scenes
t_start = time.perf_counter()
V0, V1, E, F = scenes.folded_cloth(708)
ctx = sccd.Context(0)
mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
print("setup %.2f s" % (time.perf_counter() - t_start), flush=True)
t0 = time.perf_counter()
out = []
while time.perf_counter() - t0 < 12.0:
    w0 = time.perf_counter(); n = 0
    while time.perf_counter() - w0 < 0.25:
        sccd.ccd_mesh(mesh); n += 1
    out.append((time.perf_counter() - w0) / n * 1e3)
print(" ".join("%.3f" % x for x in out))
