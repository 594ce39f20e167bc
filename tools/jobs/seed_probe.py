"""Probe: does a strided sample of the vertex-face queries, checked first, find a TOI that prunes the rest?"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "scalable-ccd_amd"))
import numpy as np
import torch
import sccd
from sccd import scenes

V0, V1, E, F = scenes.folded_cloth(708)
ctx = sccd.default_context()
mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
for is_vf in (True, False):
    vb = sccd.build_vertex_boxes(V0, V1, 0.0, ctx)
    if is_vf:
        a = sccd.DeviceAABBs(vb, ctx); b = sccd.DeviceAABBs(sccd.build_face_boxes(vb, F, ctx), ctx)
    else:
        a = sccd.DeviceAABBs(sccd.build_edge_boxes(vb, E, ctx), ctx); b = None
    bp = sccd.BroadPhase(ctx)
    bp.build(a, b)
    pairs = bp.detect_overlaps().reshape(-1, 2)
    n = len(pairs)
    d_all = torch.from_numpy(pairs).cuda()
    def run(t_dev, toi):
        best = 1e9
        for rep in range(3):
            ctx.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            t = sccd.narrow_phase(mesh, t_dev.data_ptr(), is_vf, toi=toi, n=t_dev.shape[0])
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        return t, best
    t_full, ms_full = run(d_all, 1.0)
    print("VF" if is_vf else "EE", "n", n, "full from 1.0:", t_full, "%.3f ms" % ms_full)
    t2, ms2 = run(d_all, t_full)
    print("   full from final: %.3f ms" % ms2)
    for s in (8, 16, 32, 64, 128, 256):
        d_s = d_all[::s].contiguous()
        ts, ms_s = run(d_s, 1.0)
        t3, ms3 = run(d_all, ts)
        d_h = d_all[: n // s].contiguous()
        th, ms_h = run(d_h, 1.0)
        print("   stride %3d: sample toi %.6f in %.3f ms, rest %.3f ms, sum %.3f | head sample toi %.6f" % (s, ts, ms_s, ms3, ms_s + ms3, th))
