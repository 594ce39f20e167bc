#!/usr/bin/env python3
"""Per-check cost of the two narrow-phase schemes on the cloth workload (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-ccd_amd"))
import sccd
from sccd import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
V0, V1, E, F = scenes.folded_cloth(n, seed=7)
ctx = sccd.default_context()
mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
ctx.set_option(sccd.OPT_PROFILE, 1)
for algo in (0, 1):
    ctx.set_option(sccd.OPT_NARROW_ALGO, algo)
    for rep in range(3):
        ctx.reset_profile()
        sccd.ccd_mesh_prepare(mesh, 0.0)
        t = 1.0
        checks = 0
        for is_vf in (True, False):
            t, st = sccd.ccd_mesh_pass(mesh, is_vf, t)
            checks += st["n_vf_checks"] + st["n_ee_checks"]
        prof = ctx.profile()
    ms = prof["narrow_vf"][0] + prof["narrow_ee"][0]
    print(f"algo {algo}: toi {t} checks {checks} narrow ms {ms:.3f}  ns/check (whole chip) {ms*1e6/checks:.3f}  "
          f"vf {prof['narrow_vf'][0]:.3f} ee {prof['narrow_ee'][0]:.3f}")
