"""How many inclusion checks does an a-priori TOI bound save?  Runs C4 passes with toi seeded at
1, at the final value and at values in between (the queue kernel prunes by the running minimum)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "scalable-ccd_amd"))
import torch  # noqa: F401  (loads the ROCm runtime first)
import sccd
from sccd import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
V0, V1, E, F = scenes.folded_cloth(n)
ctx = sccd.default_context()
mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
final = sccd.ccd_mesh(mesh)
print("final toi", final)
for seed in (1.0, 0.9, 0.7, 0.5, final * 1.05, final):
    sccd.ccd_mesh_prepare(mesh)
    out = []
    t = seed
    for is_vf in (True, False):
        for rep in range(3):
            ctx.synchronize()
            t0 = time.perf_counter()
            t1, st = sccd.ccd_mesh_pass(mesh, is_vf, t)
            ctx.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
        out.append((st["n_vf_checks"] + st["n_ee_checks"], st["ms_narrow"], dt, t1))
        t = t1
    print("seed %.6f  vf checks %d narrow %.3f ms (pass %.3f)  ee checks %d narrow %.3f ms (pass %.3f)  toi %r" % (
        seed, out[0][0], out[0][1], out[0][2], out[1][0], out[1][1], out[1][2], out[1][3]))
