"""Stage stamps of os_pass_k (a library built with -DOS_STAMPS: VARIANT_SRC=sort bash tools/variants.sh stamps=-DOS_STAMPS):
SCCD_LIB=.../libsccd_stamps.so python3 tools/sort_stamps.py [n] [bits]  -- per stage, the mean / max over the blocks of the
time since the FIRST block's start (100 MHz clock: 10 ns ticks)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scalable-ccd_amd"))
import numpy as np
import torch
import sccd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_700_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ctx = sccd.Context(0)
g = torch.Generator(device="cuda").manual_seed(1)
keys0 = torch.randint(0, 2**bits - 1, (n,), generator=g, dtype=torch.int32, device="cuda")
vals0 = torch.arange(n, dtype=torch.int32, device="cuda")
keys, vals = keys0.clone(), vals0.clone()
names = ["start", "tickets", "loaded+transposed", "counted+published", "ranked", "looked back", "staged", "written"]
for rep in range(4):
    keys.copy_(keys0)
    vals.copy_(vals0)
    torch.cuda.synchronize()
    ctx.sort_pairs_u32(keys.data_ptr(), vals.data_ptr(), n)
    ctx.synchronize()
    out = np.zeros((2048, 12), np.uint64)
    rc = sccd.lib().sccd_debug_sort_stamps(out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    nb = min(2048, (n + 4095) // 4096)
    t0 = out[:nb, 0].astype(np.int64).min()
    ran = out[:nb, 10] > 0
    print("blocks with a tile:", int(ran.sum()), "of", nb, " tiles per working block: max", int(out[:nb, 10].max()),
          " end of the last tile: %.2f us" % ((out[:nb, 11].astype(np.int64)[out[:nb, 10] > 1].max() - t0) / 100.0 if (out[:nb, 10] > 1).any() else 0.0))
    out = out[:nb][ran]
    nb = int(ran.sum())
    st = out[:nb, :8].astype(np.int64)
    rel = (st - t0) * 10.0 / 1000.0  # us
    if rep >= 2:
        print("pass stamps (last pass of the sort), blocks", nb)
        for k in range(8):
            print("  %-20s mean %7.2f us  min %7.2f  max %7.2f   (stage: mean %6.2f us)" % (names[k], rel[:, k].mean(), rel[:, k].min(), rel[:, k].max(), (rel[:, k] - rel[:, k - 1]).mean() if k else 0.0))
        order = np.argsort(out[:nb, 8])
        lb = (rel[:, 5] - rel[:, 4])[order]
        print("  look-back by tile number (us):", " ".join("%.1f" % lb[i] for i in range(0, nb, max(1, nb // 16))))
