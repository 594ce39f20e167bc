"""ccd() with the collision list on the 1M-triangle cloth, a few calls (for a kernel trace: where does the call go?)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "scalable-ccd_amd"))
import numpy as np
import torch  # noqa: F401
import sccd
from sccd import scenes

V0, V1, E, F = scenes.folded_cloth(int(sys.argv[1]) if len(sys.argv) > 1 else 708)
mi = int(sys.argv[2]) if len(sys.argv) > 2 else -1
ctx = sccd.default_context()
hV0, hV1 = np.asfortranarray(V0), np.asfortranarray(V1)
hE, hF = np.asfortranarray(E, dtype=np.int32), np.asfortranarray(F, dtype=np.int32)
for k in range(4):
    t0 = time.perf_counter()
    toi, col = sccd.ccd(hV0, hV1, hE, hF, 0.0, mi, 1e-6, True, ctx=ctx, want_collisions=True)
    print("call %d: %.3f ms, toi %r, %d collisions" % (k, (time.perf_counter() - t0) * 1e3, toi, len(col)))
