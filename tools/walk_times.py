"""When the waves of the last narrow-phase launch (edge-edge pass of ccd()) started, ran dry and ended -- a library built with
-DNW_ENDTIMES (bash tools/variants.sh times=-DNW_ENDTIMES):  SCCD_LIB=.../libsccd_times.so python3 tools/walk_times.py [cloth side]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scalable-ccd_amd"))
import numpy as np
import sccd
from sccd import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 708
V0, V1, E, F = scenes.folded_cloth(n)
ctx = sccd.Context(0)
mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
for rep in range(3):
    t = sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True)
ctx.synchronize()
out = np.zeros((8192, 4), np.uint64)
assert sccd.lib().sccd_debug_walk_times(out.ctypes.data_as(C.c_void_p)) == 0
out = out[out[:, 2] > 0].astype(np.int64)
t0 = out[:, 0].min()
start, dry, end, steps = (out[:, 0] - t0) / 100.0, (out[:, 1] - t0) / 100.0, (out[:, 2] - t0) / 100.0, out[:, 3]
dry[out[:, 1] == 0] = np.nan
print("toi", t, "waves", len(out), "kernel %.1f us" % end.max())
print("start  (us): max %.1f" % start.max())
qs = [0, 10, 25, 50, 75, 90, 95, 99, 100]
print("ran dry (us) percentiles", qs, ":", " ".join("%.0f" % np.nanpercentile(dry, q) for q in qs))
print("ended   (us) percentiles", qs, ":", " ".join("%.0f" % np.percentile(end, q) for q in qs))
print("tail = ended - ran dry (us):", " ".join("%.0f" % np.nanpercentile(end - dry, q) for q in qs))
last = np.argsort(end)[-12:]
print("the last waves: (ran dry, ended, steps)", [(round(float(dry[i])), round(float(end[i])), int(steps[i])) for i in last])
alive = [(x, int(((start <= x) & (end > x)).sum())) for x in np.linspace(0, end.max(), 25)]
print("waves alive over time (us, waves):", " ".join("%.0f:%d" % a for a in alive))
