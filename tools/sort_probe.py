"""Radix sort of n (u32 key, u32 index) pairs with `bits`-bit random keys, `reps` times -- for kernel traces of the
sort at the broad phase's own sizes:  rocprofv3 --kernel-trace --stats -- python3 tools/sort_probe.py 1700000 24 20"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scalable-ccd_amd"))
import torch
import sccd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_700_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 24
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = sccd.Context(0)
g = torch.Generator(device="cuda").manual_seed(1)
keys0 = torch.randint(0, 2**bits - 1, (n,), generator=g, dtype=torch.int32, device="cuda")
vals0 = torch.arange(n, dtype=torch.int32, device="cuda")
keys, vals = keys0.clone(), vals0.clone()
for _ in range(reps):
    keys.copy_(keys0)
    vals.copy_(vals0)
    torch.cuda.synchronize()
    ctx.sort_pairs_u32(keys.data_ptr(), vals.data_ptr(), n)
ctx.synchronize()
ok = bool((keys[1:] >= keys[:-1]).all().item())
print("n", n, "bits", bits, "sorted", ok)
