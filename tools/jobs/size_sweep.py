"""ms per ccd() step over mesh sizes with the projection cull / the two halves of time on and off (frozen folded cloth, from toi = 1):
python3 tools/jobs/size_sweep.py [sides...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "scalable-ccd_amd"))
import numpy as np
import sccd
from sccd import scenes

sides = [int(a) for a in sys.argv[1:]] or [50, 71, 100, 158, 224, 316, 500]
ctx = sccd.Context(0)
for n in sides:
    V0, V1, E, F = scenes.folded_cloth(n)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    row = []
    # (2 = forced, 0 = off: where each starts to pay -- the size rules of drivers.hip apply under 1; SIZE_SWEEP_DEFAULTS=1: 1 / 0 as before)
    forced = 1 if os.environ.get("SIZE_SWEEP_DEFAULTS") == "1" else 2
    for cull, halves, hist in ((forced, forced, 0), (forced, 0, 0), (forced, forced, 1), (0, 0, 0)):
        ctx.set_option(sccd.OPT_CULL, cull)
        ctx.set_option(sccd.OPT_TWO_HALVES, halves)
        ctx.set_option(sccd.OPT_TOI_GUESS, hist)
        for _ in range(8):
            t = sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True)
        ts = []
        for _ in range(60):
            t0 = time.perf_counter()
            t = sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True)
            ts.append((time.perf_counter() - t0) * 1e3)
        row.append("cull %d halves %d history %d: %.4f" % (cull, halves, hist, float(np.median(ts))))
    print("side %4d  tris %8d  toi %.4f | " % (n, len(F), t) + " | ".join(row), flush=True)
    mesh.close()
ctx.set_option(sccd.OPT_CULL, 1)
ctx.set_option(sccd.OPT_TWO_HALVES, 1)
ctx.set_option(sccd.OPT_TOI_GUESS, 1)
