#!/usr/bin/env python3
"""Per-rank cost of the multi-GPU shards, measured on ONE GPU by running every rank of an
N-rank job in turn (one ccd() on the rank's shard, as bench.py does; --split: prepare + VF pass + EE pass
with the TOI threaded through as dist.ccd_sharded does).  Prints, per N, the slowest rank's time -- what an N-GPU step would take without the two
scalar all-reduces -- next to the single-GPU step.

    python tools/shard_balance.py [--n 708] [--reps 5] [--worlds 1,2,4,8]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-ccd_amd"))

import sccd  # noqa: E402
from sccd import scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=708)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--profile", action="store_true", help="also print rank 0's per-class device time")
    ap.add_argument("--prior", action="store_true", help="leave the speculative TOI bound on (SCCD_OPT_TOI_GUESS = 1: every repetition on this frozen mesh then starts "
                    "from its own previous answer); default: every call from toi = 1, like bench.py's headline")
    ap.add_argument("--global-prior", type=float, default=0.0, help="every rank starts from this bound (sccd_ccd_mesh_from): what a job's ranks do with "
                    "sccd.dist.GlobalPrior -- 1.125 x the job's last reduced TOI, e.g. 0.4592 for the folded cloth")
    ap.add_argument("--two-halves", type=int, default=None, help="SCCD_OPT_TWO_HALVES (0 never, 1 the default rules, 2 always)")
    ap.add_argument("--split", action="store_true", help="the pass-by-pass protocol (prepare + VF pass + EE pass) instead of one ccd() per rank")
    args = ap.parse_args()

    V0, V1, E, F = scenes.folded_cloth(args.n, seed=7)
    ctx = sccd.default_context()
    if not args.prior:
        ctx.set_option(sccd.OPT_TOI_GUESS, 0)
    if args.two_halves is not None:
        ctx.set_option(sccd.OPT_TWO_HALVES, args.two_halves)
    mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
    out = []
    for world in [int(x) for x in args.worlds.split(",")]:
        per_rank = []
        toi_ranks = []
        for r in range(world):
            ctx.set_option(sccd.OPT_SHARD_COUNT, world)
            ctx.set_option(sccd.OPT_SHARD_RANK, r)
            best = 1e30
            pairs = 0
            for rep in range(args.reps + 1):
                ctx.synchronize()
                t0 = time.perf_counter()
                if args.split:
                    sccd.ccd_mesh_prepare(mesh, 0.0)
                    toi = 1.0
                    pairs = 0
                    for is_vf in (True, False):
                        toi, st = sccd.ccd_mesh_pass(mesh, is_vf, toi)
                        pairs += st["n_vf_pairs"] + st["n_ee_pairs"]
                elif args.global_prior > 0:
                    toi, st = sccd.ccd_mesh_from(mesh, args.global_prior, want_stats=True)
                    pairs = st["n_vf_pairs"] + st["n_ee_pairs"]
                else:  # what bench.py runs per rank: one ccd() on the rank's shard
                    toi, st = sccd.ccd_mesh(mesh, want_stats=True)
                    pairs = st["n_vf_pairs"] + st["n_ee_pairs"]
                ctx.synchronize()
                dt = (time.perf_counter() - t0) * 1e3
                if rep > 0:
                    best = min(best, dt)
            per_rank.append((best, pairs))
            if args.profile and r == 0:
                ctx.set_option(sccd.OPT_PROFILE, 1)
                ctx.reset_profile()
                for rep in range(args.reps):
                    if args.split:
                        sccd.ccd_mesh_prepare(mesh, 0.0)
                        t = 1.0
                        for is_vf in (True, False):
                            t, _ = sccd.ccd_mesh_pass(mesh, is_vf, t)
                    else:
                        sccd.ccd_mesh(mesh)
                prof = ctx.profile()
                ctx.set_option(sccd.OPT_PROFILE, 0)
                print("  rank 0 device ms/step:", {k: round(v[0] / args.reps, 3) for k, v in prof.items()},
                      "sum", round(sum(v[0] for v in prof.values()) / args.reps, 3))
            toi_ranks.append(toi)
        ms = [p[0] for p in per_rank]
        row = dict(world=world, max_ms=round(max(ms), 3), mean_ms=round(sum(ms) / len(ms), 3),
                   ms=[round(x, 3) for x in ms], pairs=[p[1] for p in per_rank], toi=min(toi_ranks))
        out.append(row)
        print(json.dumps(row), flush=True)
    ctx.set_option(sccd.OPT_SHARD_COUNT, 1)
    ctx.set_option(sccd.OPT_SHARD_RANK, 0)
    base = out[0]["max_ms"]
    for row in out:
        print(f"N={row['world']}: slowest rank {row['max_ms']:.3f} ms  -> speed-up {base / row['max_ms']:.2f}x "
              f"(pairs min/max {min(row['pairs'])}/{max(row['pairs'])})")


if __name__ == "__main__":
    main()
