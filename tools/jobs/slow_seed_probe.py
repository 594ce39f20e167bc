"""One soak seed whose oracle is too slow to run on the GPU box inside a batch: the library's results per rank, timed, against
the oracle's value computed beforehand (python tools/jobs/slow_seed_probe.py SEED WANT_HEX).  Seed 900004 (a 802-triangle scene
with a minimum separation of 2.2: every query touches) takes the oracle 150 s."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["SCCD_SOAK_CHILD"] = "1"
import importlib.util

spec = importlib.util.spec_from_file_location("soak", os.path.join(ROOT, "tools", "soak.py"))
soak = importlib.util.module_from_spec(spec)
spec.loader.exec_module(soak)
import sccd

seed = int(sys.argv[1])
want = float.fromhex(sys.argv[2])
V0, V1, E, F, kind, scale, shift, ms, allow_zero, arith, world, sweep_algo, scan_build, narrow_algo = soak.scene_of(seed)
ctx = sccd.default_context()
ctx.set_option(sccd.OPT_BUILD_SCAN, 1 if scan_build else 0)
ctx.set_option(sccd.OPT_ARITH, arith)
ctx.set_option(sccd.OPT_SWEEP_ALGO, sweep_algo)
ctx.set_option(sccd.OPT_NARROW_ALGO, narrow_algo)
mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
tois = []
for r in range(world):
    ctx.set_option(sccd.OPT_SHARD_COUNT, world)
    ctx.set_option(sccd.OPT_SHARD_RANK, r)
    t0 = time.time()
    toi, st = sccd.ccd_mesh(mesh, ms, -1, 1e-6, allow_zero, want_stats=True)
    tois.append(toi)
    print(f"rank {r}/{world}: toi {toi.hex()} {time.time() - t0:.2f} s  vf {st['n_vf_pairs']} ee {st['n_ee_pairs']} checks {st.get('n_vf_checks')} {st.get('n_ee_checks')}", flush=True)
print("seed", seed, "min", min(tois).hex(), "want", want.hex(), "MATCH" if min(tois) == want else "MISMATCH", flush=True)
