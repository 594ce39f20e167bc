import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scalable-ccd_amd"))
import numpy as np
import sccd
from sccd import scenes
V0, V1, E, F = scenes.folded_cloth(708, seed=7)
V0f, V1f = np.asfortranarray(V0), np.asfortranarray(V1)
Ef, Ff = np.asfortranarray(E.astype(np.int32)), np.asfortranarray(F.astype(np.int32))
ctx = sccd.default_context()
for rep in range(5):
    t0 = time.perf_counter()
    m = sccd.Mesh(V0f, V1f, Ef, Ff, ctx=ctx)
    t1 = time.perf_counter()
    toi = sccd.ccd_mesh(m)
    t2 = time.perf_counter()
    m.close()
    t3 = time.perf_counter()
    toi2 = sccd.ccd(V0f, V1f, Ef, Ff, 0.0, -1, 1e-6, True, ctx=ctx)
    t4 = time.perf_counter()
    print("mesh_create %.3f  ccd_mesh %.3f  destroy %.3f | sccd_ccd %.3f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3), toi, toi2)
# host scan alone
t0 = time.perf_counter(); ok = (Ef.min() >= 0) and (Ef.max() < len(V0f)) and (Ff.min() >= 0) and (Ff.max() < len(V0f)); t1 = time.perf_counter()
print("numpy index scan %.3f ms" % ((t1-t0)*1e3), ok)
import torch
a = torch.from_numpy(V0f.T.copy())
t0 = time.perf_counter(); b = a.cuda(); torch.cuda.synchronize(); t1 = time.perf_counter()
print("torch pageable H2D 12 MB: %.3f ms" % ((t1-t0)*1e3))
p = a.pin_memory()
t0 = time.perf_counter(); b = p.cuda(non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
print("torch pinned H2D 12 MB: %.3f ms" % ((t1-t0)*1e3))
t0 = time.perf_counter(); p.copy_(a); t1 = time.perf_counter()
print("host memcpy 12 MB into pinned: %.3f ms" % ((t1-t0)*1e3))
