"""one soak seed through the double and the float build on the GPU, timed (no oracle): is the float build's depth-first path slow on it?"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "scalable-ccd_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch  # noqa: F401
import sccd, soak
seed = int(sys.argv[1])
V0, V1, E, F, kind, scale, shift, ms, allow_zero, arith, world, sweep_algo, scan_build, narrow_algo = soak.scene_of(seed)
ctx = sccd.default_context()
ctx.set_option(sccd.OPT_ARITH, arith)
mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
for scalar in (0, 1):
    ctx.set_option(sccd.OPT_SCALAR, scalar)
    for k in range(2):
        t0 = time.perf_counter()
        t, st = sccd.ccd_mesh(mesh, ms, -1, 1e-6, allow_zero, want_stats=True)
        print("scalar %d: %.3f ms toi %r checks vf %d ee %d" % (scalar, (time.perf_counter() - t0) * 1e3, t, st["n_vf_checks"], st["n_ee_checks"]), flush=True)
