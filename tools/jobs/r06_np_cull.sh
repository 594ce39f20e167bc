# sccd_narrow_phase on the 1M-triangle cloth's own pair lists, with and without the cull in front (round 6):  bash tools/jobs/r06_np_cull.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python3 - <<PY 2>&1 | tee gpurun_out/r06/narrow_phase_on_a_list.log
import sys, os, time
sys.path.insert(0, "scalable-ccd_amd")
import numpy as np, torch, sccd
from sccd import scenes
V0, V1, E, F = scenes.folded_cloth(708)
ctx = sccd.default_context()
mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
vb, eb, fb = sccd.DeviceAABBs.from_mesh(mesh, 0.0)
bp = sccd.BroadPhase(ctx); bp.build(vb, fb); vf = bp.detect_overlaps()
bp2 = sccd.BroadPhase(ctx); bp2.build(eb); ee = bp2.detect_overlaps()
print("pairs", len(vf), len(ee))
for cull in (0, 1):
    ctx.set_option(sccd.OPT_CULL, cull)
    for rep in range(3):
        t0 = time.perf_counter(); a = sccd.narrow_phase(mesh, vf, True); t1 = time.perf_counter(); b = sccd.narrow_phase(mesh, ee, False, toi=a); t2 = time.perf_counter()
        print("cull %d: vf %.3f ms, ee %.3f ms (host lists: upload included), toi %r" % (cull, (t1 - t0) * 1e3, (t2 - t1) * 1e3, b))
# the lists on the device (what a caller of BroadPhase::detect_overlaps_partial has)
dv = torch.from_numpy(vf).cuda(); de = torch.from_numpy(ee).cuda(); torch.cuda.synchronize()
for cull in (0, 1):
    ctx.set_option(sccd.OPT_CULL, cull)
    for rep in range(4):
        t0 = time.perf_counter(); a = sccd.narrow_phase(mesh, dv.data_ptr(), True, n=len(vf)); t1 = time.perf_counter(); b = sccd.narrow_phase(mesh, de.data_ptr(), False, toi=a, n=len(ee)); t2 = time.perf_counter()
        print("cull %d, lists on the device: vf %.3f ms, ee %.3f ms, toi %r" % (cull, (t1 - t0) * 1e3, (t2 - t1) * 1e3, b))
PY
