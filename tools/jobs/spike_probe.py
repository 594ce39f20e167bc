"""Which steps of a frozen-mesh loop are slow, and did they allocate?  python3 tools/jobs/spike_probe.py [steps] [toi_guess 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "scalable-ccd_amd"))
import numpy as np
import sccd
from sccd import scenes

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
guess = int(sys.argv[2]) if len(sys.argv) > 2 else 0
V0, V1, E, F = scenes.folded_cloth(708)
ctx = sccd.Context(0)
mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
ctx.set_option(sccd.OPT_TOI_GUESS, guess)
for _ in range(10):
    sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True)
ts, al = [], []
for k in range(steps):
    a0 = ctx.get_option(sccd.OPT_ALLOC_COUNT)
    t0 = time.perf_counter()
    t, st = sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True, want_stats=True)
    ts.append((time.perf_counter() - t0) * 1e3)
    al.append((ctx.get_option(sccd.OPT_ALLOC_COUNT) - a0, st["n_vf_pairs"], st["n_ee_pairs"], st["n_vf_checks"], st["n_ee_checks"]))
ts = np.array(ts)
print("steps", steps, "median %.4f p99 %.4f max %.4f" % (np.median(ts), np.percentile(ts, 99), ts.max()), " steps over 1.5 ms:", int((ts > 1.5).sum()), " over 3 ms:", int((ts > 3).sum()))
for k in np.argsort(ts)[-6:]:
    print("  step", int(k), "%.3f ms" % ts[k], "allocs, pairs, checks:", al[k])
