#!/usr/bin/env python3
"""Randomised soak of REPEATED steps on one mesh -- what a simulation does, and what the speculative build of the broad phase
(csrc/api.hip bp_build: sort, records and sweep enqueued on the previous step's entry counts) lives on: seeded sequences of
steps whose vertices move a little (the guess holds), sometimes a lot (it breaks: more entries than the margin, another key
width), on one GPU or as the ranks of a multi-GPU job, every step's TOI and pair counts against the CPU oracle.
    python tools/soak_steps.py [sequences] [first seed]        (children of 10 sequences each, like tools/soak.py)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-ccd_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import orc
import sccd
from sccd import scenes

BATCH = 10
CHILD_TIMEOUT = 420


def supervise(cases, first):
    import subprocess

    bad = 0
    for b0 in range(first, first + cases, BATCH):
        n = min(BATCH, first + cases - b0)
        env = dict(os.environ, SCCD_SOAK_CHILD="1")
        try:
            rc = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), str(b0)], env=env, timeout=CHILD_TIMEOUT).returncode
        except subprocess.TimeoutExpired:
            rc = -1
            print(f"batch {b0}..{b0 + n - 1}: TIMEOUT after {CHILD_TIMEOUT} s", flush=True)
        if rc != 0:
            bad += 1
            print(f"batch {b0}..{b0 + n - 1}: exit code {rc}", flush=True)
    print(f"step-soak supervisor: {cases} sequences from seed {first}, {bad} bad batch(es)")
    return 1 if bad else 0


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if os.environ.get("SCCD_SOAK_CHILD") != "1":
        sys.exit(supervise(cases, first))
    shared = sccd.default_context()
    bad = steps_run = 0
    t0 = time.time()
    for seed in range(first, first + cases):
        rng = np.random.default_rng(55_000 + seed)
        kind = seed % 3
        if kind == 0:
            V0, V1, E, F = scenes.folded_cloth(int(rng.integers(12, 90)), seed=int(rng.integers(1, 10**6)))
        elif kind == 1:
            V0, V1, E, F = scenes.cloth_ball(int(rng.integers(8, 50)), int(rng.integers(1, 3)), seed=int(rng.integers(1, 10**6)))
        else:
            V0, V1, E, F = scenes.triangle_soup(int(rng.integers(50, 1200)), seed=int(rng.integers(1, 10**6)),
                                                size=float(rng.uniform(0.03, 0.25)), motion=float(rng.uniform(0.0, 0.4)))
        world = int(rng.choice([1, 1, 2, 3, 8]))
        extent = float(np.ptp(V0, axis=0).max())
        # Every third sequence runs on a context of its OWN with a small initial pair buffer: buffers of the shared context only
        # ever grow, so the overflow paths -- the first sweep of a sequence, and a sweep BEHIND A SPECULATIVE BUILD when a jump
        # multiplies the pairs (round 4: that one swept padded rows) -- would otherwise be met once per child process
        own = sccd.Context(0) if seed % 3 == 1 else None
        ctx = own if own is not None else shared
        if own is not None:
            own.set_option(sccd.OPT_OVERLAP_CAPACITY, int(rng.choice([1024, 4096, 65536])))
        # (small scenes: the defaults (1) would leave the projection cull and the two halves of time out -- forced (2) for three
        # sequences in four; with history on, the two halves then follow what the previous step returned)
        forced = 1 if seed % 4 == 3 else 2
        ctx.set_option(sccd.OPT_CULL, forced)
        ctx.set_option(sccd.OPT_TWO_HALVES, forced)
        mesh = sccd.Mesh(V0, V1, E, F, ctx=ctx)
        n_steps = int(rng.integers(4, 9))
        for step in range(n_steps):
            # mostly a simulation's step (per mille of the scene), now and then a jump (the guess must break and recover)
            amp = extent * float(rng.choice([0.0, 1e-5, 1e-4, 1e-3, 1e-3, 3e-2, 0.2]))
            W0 = V0 + rng.uniform(-amp, amp, V0.shape)
            W1 = V1 + rng.uniform(-amp, amp, V1.shape)
            mesh.update_vertices(W0, W1)
            want, nvf, nee = orc.ccd(W0, W1, E, F, nthreads=8)
            tois, pairs = [], 0
            try:
                for r in range(world):
                    ctx.set_option(sccd.OPT_SHARD_COUNT, world)
                    ctx.set_option(sccd.OPT_SHARD_RANK, r)
                    t, st = sccd.ccd_mesh(mesh, want_stats=True)
                    tois.append(t)
                    pairs += st["n_vf_pairs"] + st["n_ee_pairs"]
            finally:
                ctx.set_option(sccd.OPT_SHARD_COUNT, 1)
                ctx.set_option(sccd.OPT_SHARD_RANK, 0)
            steps_run += 1
            if min(tois) != want or pairs != nvf + nee:
                bad += 1
                print(f"MISMATCH seed {seed} kind {kind} nF {len(F)} world {world} step {step} amp {amp:.3g}: toi {min(tois)!r} want {want!r}, "
                      f"pairs {pairs} want {nvf + nee}", flush=True)
        mesh.close()
        if own is not None:
            own.close()
    print(f"step-soak: {cases} sequences, {steps_run} steps, {bad} mismatches, {time.time() - t0:.1f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
