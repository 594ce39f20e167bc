#!/usr/bin/env python3
"""bench.py -- CCD queries/s (broad + narrow phase) on MI355X, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W [--workload cloth1m|clothball10k|boxes1m|sort16m]

A step = one full pass of the hot path over one batch of synthetic input already resident in
HBM: box build -> radix sort -> STQ sweep -> projection cull -> Tight-Inclusion bisection,
for the VF list pair then the EE list (scalable_ccd::cuda::ccd(), ccd.cu:80-146).  With N > 1
(one process per GPU, launched by torch.distributed.run) every rank builds, sorts, sweeps and
narrows its window of grid cells; the only exchange is one all-reduce(min) of the time of impact
per step (RCCL) -- total work is fixed: strong scaling.

Prints ONE JSON line (rank 0) with the contract keys plus `roofline` (dominant kernel class,
algorithmic bytes / measured device time) and `cpu_baseline` (the CPU oracle on this box's
host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "scalable-ccd_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

# algorithmic bytes per unit (SURVEY.md 8d / DESIGN.md "Roofline")
BYTES_PER_QUERY = 220.0       # narrow phase: pair 8 + element indices 12/16 + 24 coords + toi
BYTES_SWEEP_PER_BOX = 64.0    # sweep: sorted box record, + 8 B per emitted pair
BYTES_SORT_PER_KEY_PASS = 16  # one radix pass over (u32 key, u32 index): read 8 + write 8
FLOP_PER_CHECK = 520.0        # one inclusion-function check = 8 corners x 57 FLOP + min/max/tests (the REFERENCE's form)
FLOP_PER_CHECK_EXECUTED = 156.0  # what np_walk_k executes per check: the min/max form of ti_inclusion_mm
FP64_VALU_PEAK_TFLOPS = 78.6  # AMD's MI355X specification: 78.6 TFLOP/s FP64 vector (= 256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz)
N_SETTLE = 3                  # untimed steps behind every change of an option, before anything is timed
BYTES_BROAD_PER_BOX = 548.0   # box build 124 + radix sort 204 + payload gather 128 + counts/scan 28 + sweep 64, + 8 B per pair


def _pmc_file(name):
    """a PMC profile (tools/pmc_traffic.sh, tools/pmc_sq.sh) if it was taken on THIS build's kernels, else None.  The files carry
    the SHA-256 of the library they profiled and of its device code (the .hip_fatbin section: every kernel's code object); a
    library whose HOST code changed and whose kernels did not -- the last fix of round 4 was one such line -- still runs the
    kernels the counters were read from, so either hash admits the file."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None
    if tj.get("lib_sha256") == lib_sha256():
        return tj
    if tj.get("device_code_sha256") and tj.get("device_code_sha256") == device_code_sha256():
        return tj
    return None


def _pmc_latest(kind, workload):
    """(file name, contents) of the newest profiles/rNN_pmc_<kind>_<workload>.json that was taken on THIS build's kernels, else (None, None)"""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_%s_%s.json" % (kind, workload))), reverse=True):
        tj = _pmc_file(os.path.basename(path))
        if tj:
            return os.path.basename(path), tj
    return None, None


def pmc_kernels(workload):
    """per-kernel HBM bytes of the newest profiles/rNN_pmc_traffic_<workload>.json that is of THIS build's kernels, else None"""
    tj = _pmc_latest("traffic", workload)[1]
    return tj["kernels"] if tj else None


def pmc_sq(workload):
    """SQ counters per kernel of the newest profiles/rNN_pmc_sq_<workload>.json (tools/pmc_sq.sh) taken on THIS build's kernels, else None"""
    tj = _pmc_latest("sq", workload)[1]
    return tj["kernels"] if tj else None


def lib_sha256():
    """hash of the libsccd_hip.so this process runs (stamps PMC profiles: bench.py only quotes a traffic figure that was
    measured on the same build)"""
    import hashlib

    import sccd

    with open(sccd._LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def device_code_sha256(path=None):
    """SHA-256 of the .hip_fatbin section -- the code objects of every kernel -- of the libsccd_hip.so this process runs (or of
    `path`): an ELF64 section table walked by hand, no tool needed.  None if there is no such section."""
    import hashlib
    import struct

    if path is None:
        import sccd

        path = sccd._LIB_PATH
    with open(path, "rb") as f:
        b = f.read()
    if b[:4] != b"\x7fELF" or b[4] != 2:
        return None
    shoff = struct.unpack_from("<Q", b, 0x28)[0]
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)

    def section(i):
        name, _typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", b, shoff + i * shentsize)
        return name, off, size

    _, stroff, _ = section(shstrndx)
    for i in range(shnum):
        name, off, size = section(i)
        if b[stroff + name: b.index(b"\0", stroff + name)] == b".hip_fatbin":
            return hashlib.sha256(b[off:off + size]).hexdigest()
    return None


def clock_warmup(step, seconds, agree=None):
    """Untimed: `seconds` of the workload's own steps BEFORE the W warm-up steps.  A fresh box starts a process with the GPU idle
    (clocks down, nothing paged in): the first run of the day on the round-3 library measured 1.41 ms per step and the second, a
    minute later on the same box, 1.30 -- W = 5 warm-up steps are 6 ms, far less than the clocks need.  Returns the steps run
    (reported as config.clock_warmup_steps); --clock-warmup 0 switches it off.
    `agree` (several ranks: a step holds a collective, so every rank must run the SAME number of steps -- a loop that each rank
    ends by its own clock leaves the ranks out of step and the job hangs): maps a rank's own count to the count all ranks run."""
    if seconds <= 0:
        return 0
    if agree is None:
        n = 0
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            step()
            n += 1
        return n
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    per = max(1e-5, (time.perf_counter() - t0) / 3)
    n = agree(min(20000, max(0, int(seconds / per) - 3)))
    for _ in range(n):
        step()
    return n + 3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cloth1m", choices=["cloth1m", "clothball10k", "boxes1m", "sort16m"])
    ap.add_argument("--cloth-n", type=int, default=708, help="cloth grid side for cloth1m (708 -> 999,698 tris)")
    ap.add_argument("--arith", type=int, default=1, help="1 (library default): a*b+c fused as the reference's nvcc --use_fast_math build fuses them; 0 strict (every product and sum rounded separately)")
    ap.add_argument("--sweep-algo", type=int, default=0, help="0 auto, 1 plain SAP, 2 filter/queue/confirm, 3 direct")
    ap.add_argument("--max-iter", type=int, default=-1, help="Tight-Inclusion check limit per query (the IPC Toolkit passes 10000000); -1: none")
    ap.add_argument("--limit-level-order", action="store_true", help="SCCD_OPT_LIMIT_LEVEL_ORDER: check limits on the level-synchronous kernels (round 2's default)")
    ap.add_argument("--boxes-n", type=int, default=1_000_000, help="boxes1m: number of boxes (SURVEY 8d also asks for 16,000,000; extents shrink so that ~10 overlaps per box remain)")
    ap.add_argument("--boxes-variant", default="iso", choices=["iso", "thin"], help="boxes1m: isotropic, or cloth-like (z extents x 0.01: SURVEY 8d)")
    ap.add_argument("--jitter", type=float, default=0.0, help="cloth workloads: the mesh MOVES between steps (end positions + A x uniform noise, eight "
                    "device-resident variants in turn, sccd_mesh_update_vertices inside the timed region): speculative builds can miss; "
                    "reports spec_hit_rate, p50 / p99 ms per step and the cost of a missed guess.  0 (default): the frozen mesh of the headline")
    ap.add_argument("--jitter-alternate", type=float, default=1.0, help="--jitter: every second variant's amplitude is scaled by this factor "
                    "(e.g. 0.05: large and small motions alternate, the entry counts jump from step to step and the speculative guesses break)")
    ap.add_argument("--jitter-fraction", type=float, default=1.0, help="--jitter: only this fraction of the vertices moves (a few large boxes "
                    "among small ones change how many cells a box spans, i.e. the entry counts the speculative build guesses)")
    ap.add_argument("--cliffs", action="store_true", help="cloth workloads: also time the paths that do not run on the plain fast kernel -- ccd() with the "
                    "per-query collision list (with and without a check limit), a check limit on the level-synchronous kernels, the float build "
                    "-- each as a multiple of the default step (block `cliffs` of the line)")
    ap.add_argument("--cull", type=int, default=None, help="SCCD_OPT_CULL: 1 (library default) the projection cull in front of the bisection, 0: every pair is bisected")
    ap.add_argument("--two-halves", type=int, default=None, help="SCCD_OPT_TWO_HALVES: 1 (library default) plain narrow launches from a TOI above 0.5 run as two launches over the halves of time, 0: one launch")
    ap.add_argument("--clock-warmup", type=float, default=1.0, help="seconds of untimed steps before the W warm-up steps (GPU clocks, first touches); 0: none")
    ap.add_argument("--passes-apart", action="store_true", help="SCCD_OPT_PASSES_APART for the whole run: one stream, the passes one after the other (kernel profiles: one kernel at a time on the chip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-n", type=int, default=0, help="cloth side of the CPU sample (0 = auto)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("SCCD_FORCE_DIST") == "1"  # the latter: exercise RCCL with one rank
    # SCCD_BENCH_BACKEND=gloo lets several ranks share ONE GPU (a functional check of the N > 1 path on a
    # 1-GPU box; the timings of such a run mean nothing).  The driver's runs use nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("SCCD_BENCH_BACKEND", "nccl")
    dev_index = (local_rank % max(1, torch.cuda.device_count())) if world > 1 else 0
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    # --gpus N means N ranks, one GPU each (python -m torch.distributed.run --nproc-per-node N ...)
    rccl_ranks = dist.get_world_size() if use_dist else 1
    if args.gpus != rccl_ranks and os.environ.get("SCCD_FORCE_DIST") != "1":
        raise SystemExit("bench.py --gpus %d needs %d ranks (launch with python -m torch.distributed.run --nproc-per-node %d), "
                         "found %d" % (args.gpus, args.gpus, args.gpus, rccl_ranks))
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the scalars of the all-reduces live

    import sccd
    from sccd import dist as sdist
    from sccd import scenes

    ctx = sccd.Context(dev.index)
    ctx.set_option(sccd.OPT_ARITH, args.arith)
    ctx.set_option(sccd.OPT_SWEEP_ALGO, args.sweep_algo)
    ctx.set_option(sccd.OPT_SHARD_RANK, rank)
    ctx.set_option(sccd.OPT_SHARD_COUNT, world)
    if args.cull is not None:
        ctx.set_option(sccd.OPT_CULL, args.cull)
    if args.two_halves is not None:
        ctx.set_option(sccd.OPT_TWO_HALVES, args.two_halves)
    apart_run = 1 if args.passes_apart else 0
    ctx.set_option(sccd.OPT_PASSES_APART, apart_run)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    result = None
    if args.workload in ("cloth1m", "clothball10k"):
        if args.workload == "cloth1m":
            V0, V1, E, F = scenes.folded_cloth(args.cloth_n)
            wl = {"workload": f"folded cloth self-collision {args.cloth_n}x{args.cloth_n} ({len(F)} tris), full broad+narrow (BASELINE configs[3]/[4])"}
        else:
            V0, V1, E, F = scenes.cloth_ball()
            wl = {"workload": f"cloth-ball 10k-tri two-frame pair ({len(F)} tris), full broad+narrow (BASELINE configs[1])"}
        # inputs resident in HBM before the timed region: upload through torch tensors (column-major)
        tV0 = torch.from_numpy(np.asfortranarray(V0).T.copy()).to(dev)  # [3][nV] contiguous == column-major nV x 3
        tV1 = torch.from_numpy(np.asfortranarray(V1).T.copy()).to(dev)
        tE = torch.from_numpy(np.ascontiguousarray(E.T)).to(dev)
        tF = torch.from_numpy(np.ascontiguousarray(F.T)).to(dev)
        torch.cuda.synchronize()
        mesh = sccd.Mesh(tV0.data_ptr(), tV1.data_ptr(), tE.data_ptr(), tF.data_ptr(), ctx=ctx, on_device=True,
                         nV=len(V0), nE=len(E), nF=len(F))
        params = dict(min_distance=0.0, max_iterations=args.max_iter, tolerance=1e-6, allow_zero_toi=True)
        if args.limit_level_order:
            ctx.set_option(sccd.OPT_LIMIT_LEVEL_ORDER, 1)

        if args.jitter > 0:
            result = bench_jitter(args, ctx, sccd, torch, mesh, tV0, tV1, params, wl, world)
            if use_dist:
                dist.destroy_process_group()
            if rank == 0:
                sys.stdout.flush()
                print(json.dumps(result), flush=True)
            return

        def step():
            # one ccd() per rank on its shard of the cell grid (SHARD_RANK / SHARD_COUNT options of the context),
            # then ONE all-reduce(min) of the 8-byte TOI; SCCD_BENCH_SPLIT=1 runs the pass-by-pass protocol instead
            if os.environ.get("SCCD_BENCH_SPLIT") == "1":
                return sdist.ccd_sharded(
                    lambda is_vf, toi: sccd.ccd_mesh_pass(mesh, is_vf, toi, **params),
                    rank, world, device=red_dev, prepare=lambda: sccd.ccd_mesh_prepare(mesh, 0.0))
            if dev_min is not None:
                # RCCL: the TOI stays in a persistent device word that is all-reduced in place, stream-ordered behind the step;
                # the host does not wait for the collective (the value is read once, after the timed region)
                t, st = sccd.ccd_mesh_dev(mesh, dev_min.ptr(), want_stats=True, **params)
                dev_min.reduce()
                return t, st
            t, st = sccd.ccd_mesh(mesh, want_stats=True, **params)
            return sdist.allreduce_min(t, device=red_dev), st

        dev_min = sdist.DeviceMin(ctx, dev) if (use_dist and backend == "nccl" and os.environ.get("SCCD_BENCH_SPLIT") != "1") else None

        # Warm-up, then two more untimed steps with hipEvents around EVERY kernel class: the per-class breakdown,
        # and which class dominates.  The timed region keeps events on that one class only -- two event records
        # per class scope cost 0.15 ms of a 2.3 ms step with all classes on.
        n_prof = 2
        def agree(n):  # the smallest of the ranks' counts
            t = torch.tensor([n], dtype=torch.int64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item())

        # THE HEADLINE STARTS EVERY CALL FROM toi = 1 (ccd.cu:125): SCCD_OPT_TOI_GUESS = 0 for warm-up, profile and timed steps.  The
        # speculative bound (a step starts from 1.125 x the previous step's TOI on the same mesh) is timed afterwards, in its own
        # block -- on this frozen mesh every step would hand itself its own answer (VERDICT r04 / ADVICE r04).
        ctx.set_option(sccd.OPT_TOI_GUESS, 0)
        n_clock = clock_warmup(step, args.clock_warmup, agree if use_dist else None)
        for _ in range(args.warmup):
            step()
        # ---- the passes apart (untimed steps, SCCD_OPT_PASSES_APART): each class's own duration with nothing beside it -- the
        # figures that need one kernel at a time on the chip (broad_phase, rank_max, roofline.alone), and the choice of the DOMINANT
        # kernel: the class with the longest launch when it has the chip to itself.  (In the default step the launches overlap on two
        # streams and a launch's own span says little: the vertex-face walk kernel is enqueued while the edge-edge sweep holds every CU
        # and "runs" for 250 us of which it works for 100.)
        ctx.set_option(sccd.OPT_PASSES_APART, 1)
        for _ in range(2):
            step()
        ctx.set_option(sccd.OPT_PROFILE, 1)
        ctx.reset_profile()
        st_apart = {}
        for _ in range(n_prof):
            _, st_apart = step()
        prof_apart_raw = ctx.profile()
        prof_apart = {k: v[0] / n_prof for k, v in prof_apart_raw.items()}
        ctx.set_option(sccd.OPT_PROFILE, 0)
        ctx.set_option(sccd.OPT_PASSES_APART, apart_run)
        for _ in range(N_SETTLE):
            step()
        ctx.set_option(sccd.OPT_PROFILE, 1)
        ctx.reset_profile()
        for _ in range(n_prof):
            step()
        prof_all = ctx.profile()
        # the dominant KERNEL: the class with the longest launch, each alone on the chip (above)
        dom = max(prof_apart_raw, key=lambda k: prof_apart_raw[k][0] / max(1, prof_apart_raw[k][1]))
        class_id = {"boxes": 0, "sort": 1, "cull": 2, "sweep": 3, "narrow_vf": 4, "narrow_ee": 5, "sweep_ee": 6}  # SCCD_PROF_*
        ctx.set_option(sccd.OPT_PROFILE, (1 << class_id[dom]) << 1)
        # settling steps AFTER the last option change and BEFORE t0 (round 4 timed the steps right behind three option writes and a
        # profile reset: the driver's 20-step window came out 10 % above the 100-step lines of the same library); the profile of
        # the timed region is reset behind them
        for _ in range(N_SETTLE):
            step()
        ctx.reset_profile()
        barrier()
        t0 = time.perf_counter()
        q_local = 0
        stats = {}
        toi = 1.0
        step_ms = []
        dev_ms = []  # the DEVICE's own clock around each step (SCCD_OPT_DEVICE_SPAN_NS: first kernel's start .. the kernel behind the last read-back)
        waits0 = ctx.get_option(sccd.OPT_HOST_WAITS)
        for _ in range(args.steps):
            ts = time.perf_counter()
            toi, stats = step()
            step_ms.append((time.perf_counter() - ts) * 1e3)  # (a step ends with its TOI on the host: the host clock sees all of it)
            dev_ms.append(ctx.get_option(sccd.OPT_DEVICE_SPAN_NS) * 1e-6)
            q_local += stats["n_vf_pairs"] + stats["n_ee_pairs"]
        barrier()
        dt = time.perf_counter() - t0
        host_waits_per_step = (ctx.get_option(sccd.OPT_HOST_WAITS) - waits0) / max(1, args.steps)
        if dev_min is not None:
            toi = dev_min.value()  # (the reduced word of the last step; every step of this frozen mesh has the same)
        prof = ctx.profile()
        ctx.set_option(sccd.OPT_PROFILE, 0)
        # the speculative TOI bound (a step of a mesh whose previous step found an impact at T starts from 1.125 T, verified):
        # the same steps WITH it, 20 of them behind settling steps -- hits, misses, ms per step
        ctx.set_option(sccd.OPT_TOI_GUESS, 1)
        for _ in range(N_SETTLE):
            step()
        ctx.set_option(sccd.OPT_TOI_GUESS_HITS, 0)
        barrier()
        tg = time.perf_counter()
        for _ in range(20):
            step()
        barrier()
        guess = {"ms_per_step_with": round((time.perf_counter() - tg) / 20 * 1e3, 4),
                 "hits": ctx.get_option(sccd.OPT_TOI_GUESS_HITS), "misses": ctx.get_option(sccd.OPT_TOI_GUESS_MISSES)}
        ctx.set_option(sccd.OPT_TOI_GUESS, 0)
        if use_dist:
            # a JOB's history: every rank starts from 1.125 x the REDUCED result of the last step (sccd.dist.GlobalPrior; a rank's own last
            # result is a worse bound, and a rank must not redo a step because ITS shard has nothing below the bound)
            gp = sdist.GlobalPrior(device=red_dev)
            run_from = lambda b: sccd.ccd_mesh_from(mesh, b, want_stats=True, **params)  # noqa: E731
            for _ in range(N_SETTLE):
                gp.step(run_from)
            gp.hits = gp.misses = 0
            barrier()
            tgp = time.perf_counter()
            for _ in range(20):
                gp.step(run_from)
            barrier()
            guess["ms_per_step_global_prior"] = round((time.perf_counter() - tgp) / 20 * 1e3, 4)
            guess["global_prior_hits"], guess["global_prior_misses"] = gp.hits, gp.misses
        for _ in range(N_SETTLE):
            step()
        # max over ranks of the elapsed time, sum over ranks of the queries
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        qq = torch.tensor([float(q_local), float(stats["n_vf_checks"] + stats["n_ee_checks"]),
                           float(stats["n_vf_candidates"] + stats["n_ee_candidates"])], dtype=torch.float64, device=red_dev)
        if use_dist:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(qq, op=dist.ReduceOp.SUM)
        dt = float(tt.item())
        queries_per_step = float(qq[0].item()) / args.steps
        value = float(qq[0].item()) / dt

        # roofline of the dominant kernel class of this rank, measured live with hipEvents on the context's stream
        # (SCCD_OPT_PROFILE).  Broad-phase classes are HBM-bound: algorithmic bytes / device time against
        # 8 TB/s.  The narrow phase is bound by FP64 vector issue: inclusion checks x 520 FLOP
        # (the survey's per-check figure: 8 corners x 57 + min/max/tests) / device time against the 78.6 TFLOP/s FP64
        # vector peak; its HBM figure (220 B per query) is kept beside it as `hbm`.
        n_boxes = len(V0) + len(F) + len(E)
        q_vf, q_ee = stats["n_vf_pairs"], stats["n_ee_pairs"]  # this rank's queries per step
        c_vf, c_ee = stats["n_vf_checks"], stats["n_ee_checks"]  # ... and inclusion checks (last step)
        # (the walk kernels see what the projection cull left of a pass's queries; the cull sees them all)
        k_vf, k_ee = q_vf - stats.get("n_vf_culled", 0), q_ee - stats.get("n_ee_culled", 0)
        units = {  # class -> (algorithmic bytes per step, kernel name, name in the rocprofv3 summaries)
            "narrow_ee": (BYTES_PER_QUERY * k_ee, "np_walk_k<false> (edge-edge Tight-Inclusion)", "np_walk_k<false, %d, 0>" % args.arith),
            "narrow_vf": (BYTES_PER_QUERY * k_vf, "np_walk_k<true> (vertex-face Tight-Inclusion)", "np_walk_k<true, %d, 0>" % args.arith),
            "sweep": (BYTES_SWEEP_PER_BOX * (len(V0) + len(F)) + 8.0 * q_vf, "sweep_band_k<false, 3> (vertices x faces: the faces' rows)", "sweep_band_k<false, 3>"),
            "sweep_ee": (BYTES_SWEEP_PER_BOX * len(E) + 8.0 * q_ee, "sweep_band_k<true, 1> (the edge-edge sweep)", "sweep_band_k<true, 1>"),
            "sort": (BYTES_SORT_PER_KEY_PASS * 4 * n_boxes, "onesweep radix sort + scans", "os_pass_k"),
            "boxes": (124.0 * n_boxes, "box build, cell fill, sorted records", "entry_record_k"),
            "cull": (BYTES_PER_QUERY * (q_vf + q_ee), "np_cull_k (the projection cull: 2 launches per step)", "np_cull_k"),
        }
        ms_dom, launches = prof[dom]  # live, over the timed region
        per_launch_ms = ms_dom / max(1, launches)
        launches_per_step = max(1, launches) / args.steps
        hbm_achieved = units[dom][0] / launches_per_step / (per_launch_ms * 1e-3) / 1e9 if ms_dom > 0 else 0.0
        # HBM bytes per launch from the PMC passes (tools/pmc_traffic.sh, same workload) -- only if they were taken on
        # THIS build of the library: the JSON carries the hash of the libsccd_hip.so it profiled
        traffic, traffic_note = None, "no PMC profile of this workload"
        try:
            defaults = args.cull is None and args.two_halves is None and args.max_iter < 0 and args.jitter == 0.0  # (the profile is of the default step)
            if defaults and args.workload == "cloth1m" and args.cloth_n == 708 and world == 1:
                fn, tj = _pmc_latest("traffic", "cloth1m")
                if tj:
                    traffic = tj["kernels"][units[dom][2]]["hbm_bytes_per_launch_corrected"]
                    traffic_note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on this build's kernels (profiles/%s: " % fn
                                    + ("same library" if tj.get("lib_sha256") == lib_sha256() else "same device code, host code changed since") + ")")
                else:
                    traffic_note = "no profiles/rNN_pmc_traffic_cloth1m.json taken on this library's kernels: dropped"
        except Exception:
            pass
        checks = float(c_vf + c_ee)
        narrow_ms = (prof_all["narrow_vf"][0] + prof_all["narrow_ee"][0] + prof_all["cull"][0]) / n_prof
        if dom.startswith("narrow"):
            dom_checks = float(c_ee if dom == "narrow_ee" else c_vf)
            achieved = dom_checks * FLOP_PER_CHECK / launches_per_step / (per_launch_ms * 1e-3) / 1e12 if ms_dom > 0 else 0.0
            sq = pmc_sq("cloth1m") if (args.workload == "cloth1m" and args.cloth_n == 708 and world == 1) else None
            valu_per_check = None
            if sq and units[dom][2] in sq and sq[units[dom][2]].get("SQ_INSTS_VALU"):
                valu_per_check = round(sq[units[dom][2]]["SQ_INSTS_VALU"] * launches_per_step / max(1.0, dom_checks), 3)
            roofline = {
                "bound": "fp64_valu", "kernel": units[dom][1], "achieved": round(achieved, 3), "peak": FP64_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / FP64_VALU_PEAK_TFLOPS, 5), "traffic": traffic,
                "flop_per_check": FLOP_PER_CHECK, "checks_per_launch": dom_checks / launches_per_step,
                # `achieved` prices a check at the REFERENCE's 520 FLOP (the algorithmic unit).  The kernel executes
                # the min/max form of the inclusion function, ~156 FP64 operations per check: what the vector ALU really does
                "executed_flop_per_check": FLOP_PER_CHECK_EXECUTED,
                "executed_tflops": round(achieved * FLOP_PER_CHECK_EXECUTED / FLOP_PER_CHECK, 3),
                "executed_frac": round(achieved * FLOP_PER_CHECK_EXECUTED / FLOP_PER_CHECK / FP64_VALU_PEAK_TFLOPS, 5),
                "queries_per_launch": float(k_ee if dom == "narrow_ee" else k_vf),
                "note": ("since round 5 the pass's projection cull drops ~95 %% of the overlap pairs in front of this kernel and its launch asks about the first half of the "
                         "step only: it bisects %.2f M of the pass's %.2f M queries, and its duration is the longest walk's dependent chain (50-65 checks of ~1.7 us on a lone "
                         "wave), not vector issue -- `frac` fell from 0.34 (round 4: 27 M checks in 0.52 ms) because the work went away, the step from 1.18 to 0.86 ms" % (
                             (k_ee if dom == "narrow_ee" else k_vf) / 1e6, (q_ee if dom == "narrow_ee" else q_vf) / 1e6)),
                "valu_per_check": valu_per_check,  # wave-level VALU instructions per check (SQ_INSTS_VALU of this build's PMC profile), else null
                "hbm": {"achieved": round(hbm_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(hbm_achieved / HBM_PEAK_GBS, 5), "bytes_per_query": BYTES_PER_QUERY},
            }
        else:
            roofline = {"bound": "hbm", "kernel": units[dom][1], "achieved": round(hbm_achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(hbm_achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "algorithmic_bytes_per_launch": units[dom][0] / launches_per_step}
            if dom.startswith("sweep"):
                roofline["note"] = ("since round 5 the longest kernel of the step is a sweep, not a narrow launch (the projection cull took 95 % of the bisection's work: "
                                    "DESIGN 5.5).  SURVEY 8d prices a sweep in bytes -- 64 B per sorted box + 8 B per emitted pair -- against HBM; the kernel itself is bound "
                                    "by instruction issue (filter, queue and confirm stages per candidate column: DESIGN 5.4), which is why `frac` is low and has been since round 3")
            if prof_apart.get(dom, 0) > 0:
                a_alone = units[dom][0] / launches_per_step / (prof_apart[dom] / launches_per_step * 1e-3) / 1e9
                roofline["alone"] = {"launch_ms": round(prof_apart[dom] / launches_per_step, 4), "achieved": round(a_alone, 2), "frac": round(a_alone / HBM_PEAK_GBS, 5),
                                     "note": "the same launch with nothing beside it (SCCD_OPT_PASSES_APART, untimed steps in this process)"}
        roofline.update({
            "traffic_note": traffic_note, "avg_launch_ms": round(per_launch_ms, 4), "launches": launches,
            "class_ms_per_step": {k: round(v[0] / n_prof, 4) for k, v in prof_all.items()},
            "class_ms_source": "%d untimed steps after the warm-up with events on every class (the timed region times %s only)" % (n_prof, dom),
            "class_ms_note": "each class's own device time; the classes of a step OVERLAP (two streams: the edge-edge build beside the vertex-face broad phase, the edge-edge sweep and narrow kernel beside the vertex-face narrow kernel): they sum to more than ms_per_step",
            "narrow_phase": {"checks_per_step": checks, "launch_ms_sum": round(narrow_ms, 4),
                             "note": "the two narrow launches overlap in the default configuration: launch_ms_sum is not their wall span (broad_phase.passes_apart has each launch alone)"},
        })
        # ---- the broad phase against SURVEY 8d's formula (548 B per box + 8 B per pair), passes apart (measured before the timed region)
        broad_ms = prof_apart["boxes"] + prof_apart["sort"] + prof_apart["sweep"] + prof_apart.get("sweep_ee", 0.0)
        if dom.startswith("narrow") and prof_apart.get(dom, 0) > 0:
            # `achieved` above is the contract's figure: the launch's own duration in the timed region.  Since round 4's read-back
            # mailbox (profiles/HISTORY.md 5.6) the edge-edge launch STARTS ~120 us earlier -- in the SIMD slots its sweep left, beside the
            # vertex-face kernel, taking over as that kernel's waves retire -- so its duration now holds ~150 us in which it has
            # a fraction of the chip: the step got shorter, the launch longer, `frac` lower.  The same launch with the chip to
            # itself (passes apart, same process, untimed) is the kernel's own figure:
            c_alone = float(st_apart["n_ee_checks" if dom == "narrow_ee" else "n_vf_checks"])
            a_alone = c_alone * FLOP_PER_CHECK / (prof_apart[dom] * 1e-3) / 1e12
            roofline["alone"] = {
                "launch_ms": round(prof_apart[dom], 4), "checks_per_launch": c_alone, "achieved": round(a_alone, 3),
                "frac": round(a_alone / FP64_VALU_PEAK_TFLOPS, 5),
                "executed_frac": round(a_alone * FLOP_PER_CHECK_EXECUTED / FLOP_PER_CHECK / FP64_VALU_PEAK_TFLOPS, 5),
                "note": "the dominant launch with nothing beside it (SCCD_OPT_PASSES_APART, %d untimed steps in this process); `frac` above "
                        "is the same launch as it runs in the timed steps, where it shares the SIMDs with the vertex-face kernel for "
                        "its first ~150 us (DESIGN 5.7)" % n_prof,
            }
        # the slowest rank's device time per phase (passes apart): narrow-phase scaling is readable on its own
        mx = torch.tensor([broad_ms, prof_apart["narrow_vf"] + prof_apart["narrow_ee"] + prof_apart["cull"]], dtype=torch.float64, device=red_dev)
        if use_dist:
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        rank_max = {"broad_ms": round(float(mx[0].item()), 4), "narrow_ms": round(float(mx[1].item()), 4),
                    "note": "max over ranks of each rank's device time per phase, passes apart (two untimed steps)"}
        broad_bytes = BYTES_BROAD_PER_BOX * n_boxes + 8.0 * (q_vf + q_ee)
        broad = {"bytes_per_step": broad_bytes, "formula": "548 B x boxes + 8 B x pairs (SURVEY 8d)", "boxes": n_boxes,
                 "ms_passes_apart": round(broad_ms, 4), "achieved": round(broad_bytes / max(1e-9, broad_ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                 "peak": HBM_PEAK_GBS, "frac": round(broad_bytes / max(1e-9, broad_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                 "passes_apart": {k: round(v, 4) for k, v in prof_apart.items()},
                 # the sort + scan class by SURVEY 8d's fixed credit (232 B per box: eight 8-bit passes of a 12-byte pair + histogram + scan), whatever the passes really run
                 "sort_class": {"bytes_per_step": 232.0 * n_boxes, "ms": round(prof_apart["sort"], 4),
                                "frac": round(232.0 * n_boxes / max(1e-9, prof_apart["sort"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}}
        # ---- the reference ccd()'s real entry cost (ccd.cu:103-106 uploads the mesh inside the call): sccd_ccd from pageable host matrices
        host_ms = None
        if world == 1:
            ctx.synchronize()
            # (column-major already, like the Eigen matrices of a C++ caller: the binding passes the pointers through)
            hV0, hV1 = np.asfortranarray(V0, dtype=np.float64), np.asfortranarray(V1, dtype=np.float64)
            hE, hF = np.asfortranarray(E, dtype=np.int32), np.asfortranarray(F, dtype=np.int32)
            sccd.ccd(hV0, hV1, hE, hF, 0.0, args.max_iter, 1e-6, True, ctx=ctx)
            t_h = []
            for _ in range(5):
                th0 = time.perf_counter()
                sccd.ccd(hV0, hV1, hE, hF, 0.0, args.max_iter, 1e-6, True, ctx=ctx)
                t_h.append((time.perf_counter() - th0) * 1e3)
            host_ms = round(min(t_h), 4)
        # ---- THE BET OF THE TWO HALVES, PRICED IN THE DEFAULT LINE: the same mesh with the step cut to 0.6 -- the earliest
        # impact at 0.68, behind the first half of time -- under the headline's terms (no history: every call takes the bet, and loses it here)
        late_headline = None
        if world == 1 and args.max_iter < 0 and args.jitter == 0.0:
            m_late = sccd.Mesh(V0, V0 + 0.6 * (V1 - V0), E, F, ctx=ctx)
            for _ in range(N_SETTLE + 2):
                toi_late = sccd.ccd_mesh(m_late, **params)
            lt = []
            for _ in range(20):
                tl0 = time.perf_counter()
                sccd.ccd_mesh(m_late, **params)
                lt.append((time.perf_counter() - tl0) * 1e3)
            ctx.set_option(sccd.OPT_TWO_HALVES, 0)
            for _ in range(N_SETTLE):
                sccd.ccd_mesh(m_late, **params)
            lo = []
            for _ in range(20):
                tl0 = time.perf_counter()
                sccd.ccd_mesh(m_late, **params)
                lo.append((time.perf_counter() - tl0) * 1e3)
            ctx.set_option(sccd.OPT_TWO_HALVES, args.two_halves if args.two_halves is not None else 1)
            late_headline = {"step_scale": 0.6, "toi": toi_late, "ms_per_step": round(sum(lt) / len(lt), 4), "ms_per_step_p50": round(sorted(lt)[len(lt) // 2], 4),
                             "one_launch_ms_per_step": round(sum(lo) / len(lo), 4), "one_launch_ms_per_step_p50": round(sorted(lo)[len(lo) // 2], 4),
                             "note": "the headline's mesh with its motion cut to 0.6 (earliest impact behind 0.5), history off like the headline: the default settings "
                                     "(two launches per pass, a bet on an early impact, lost here) against SCCD_OPT_TWO_HALVES = 0; 20 steps each"}
            m_late.close()
            for _ in range(N_SETTLE):
                step()
        cliffs = None
        if args.cliffs and world == 1:
            def best(fn, reps=3):
                fn()
                ts = []
                for _ in range(reps):
                    ctx.synchronize()
                    tc = time.perf_counter()
                    fn()
                    ctx.synchronize()
                    ts.append((time.perf_counter() - tc) * 1e3)
                return round(min(ts), 4)

            hV0, hV1 = np.asfortranarray(V0, dtype=np.float64), np.asfortranarray(V1, dtype=np.float64)
            hE, hF = np.asfortranarray(E, dtype=np.int32), np.asfortranarray(F, dtype=np.int32)
            base_host = best(lambda: sccd.ccd(hV0, hV1, hE, hF, 0.0, -1, 1e-6, True, ctx=ctx))
            base_step = best(lambda: sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True), 5)
            cliffs = {"default_step_ms": base_step, "default_host_path_ms": base_host}
            cliffs["collisions_host_path_ms"] = best(lambda: sccd.ccd(hV0, hV1, hE, hF, 0.0, -1, 1e-6, True, ctx=ctx, want_collisions=True))
            cliffs["collisions_max_iter_1e7_host_path_ms"] = best(lambda: sccd.ccd(hV0, hV1, hE, hF, 0.0, 10_000_000, 1e-6, True, ctx=ctx, want_collisions=True))
            cliffs["collisions_max_iter_1000_host_path_ms"] = best(lambda: sccd.ccd(hV0, hV1, hE, hF, 0.0, 1000, 1e-6, True, ctx=ctx, want_collisions=True))
            cliffs["max_iter_1e7_step_ms"] = best(lambda: sccd.ccd_mesh(mesh, 0.0, 10_000_000, 1e-6, True), 5)
            cliffs["ipc_ccd_strategy_max_iter_1e7_host_path_ms"] = best(lambda: sccd.ipc_ccd_strategy(hV0, hV1, hE, hF, 0.0, 10_000_000, 1e-6, ctx=ctx))
            ctx.set_option(sccd.OPT_LIMIT_LEVEL_ORDER, 1)
            cliffs["max_iter_1e7_level_order_step_ms"] = best(lambda: sccd.ccd_mesh(mesh, 0.0, 10_000_000, 1e-6, True), 2)
            ctx.set_option(sccd.OPT_LIMIT_LEVEL_ORDER, 0)
            cliffs["max_iter_100_step_ms"] = best(lambda: sccd.ccd_mesh(mesh, 0.0, 100, 1e-6, True), 2)  # below 4096: level order
            ctx.set_option(sccd.OPT_SCALAR, 1)
            cliffs["float_build_step_ms"] = best(lambda: sccd.ccd_mesh(mesh, 0.0, -1, 1e-6, True), 2)
            ctx.set_option(sccd.OPT_SCALAR, 0)
            # THE BET OF THE TWO HALVES OF TIME (csrc/narrow_walk.inc): the same mesh, the step cut short so that the earliest impact comes
            # late (x 0.6: at 0.68) or not at all (x 0.3) -- without history (every call bets on an early impact), with one launch per
            # pass (SCCD_OPT_TWO_HALVES = 0), and with history (the library's default: the last call on the mesh settles the bet)
            g_was, h_was = ctx.get_option(sccd.OPT_TOI_GUESS), ctx.get_option(sccd.OPT_TWO_HALVES)
            late = {}
            for tag, scale in (("impact_at_0.68", 0.6), ("no_impact", 0.3)):
                m2 = sccd.Mesh(V0, V0 + scale * (V1 - V0), E, F, ctx=ctx)
                row = {}
                for name, gh, hv in (("two_halves_no_history_ms", 0, 1), ("one_launch_no_history_ms", 0, 0), ("with_history_ms", 1, 1)):  # (1: the defaults)
                    ctx.set_option(sccd.OPT_TOI_GUESS, gh)
                    ctx.set_option(sccd.OPT_TWO_HALVES, hv)
                    row[name] = best(lambda: sccd.ccd_mesh(m2, 0.0, -1, 1e-6, True), 5)
                row["toi"] = sccd.ccd_mesh(m2, 0.0, -1, 1e-6, True)
                late[tag] = row
                m2.close()
            ctx.set_option(sccd.OPT_TOI_GUESS, g_was)
            ctx.set_option(sccd.OPT_TWO_HALVES, h_was)
            cliffs["late_impact"] = late
            cliffs["note"] = ("host_path: sccd_ccd() from pageable host matrices (upload inside the call), collisions = the SCALABLE_CCD_TOI_PER_QUERY signature; "
                              "step: sccd_ccd_mesh() on the resident mesh; best of 3 / 5 / 2")
        result = {
            "metric": "CCD queries/sec (broad+narrow)", "value": value, "unit": "queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": dict(wl, queries_per_step=queries_per_step, toi=toi, arith=["strict", "fma"][args.arith], clock_warmup_steps=n_clock,
                           checks_per_step=float(qq[1].item()), candidates_per_step=float(qq[2].item()),
                           parallelism=f"cell windows sharded over {world} GPU(s), one RCCL all-reduce(min) of the TOI per step",
                           rccl_ranks=rccl_ranks, backend=(backend if use_dist else "none"),
                           projection_cull=int(ctx.get_option(sccd.OPT_CULL)), two_halves_of_time=int(ctx.get_option(sccd.OPT_TWO_HALVES)), culled_per_step=int(stats.get("n_vf_culled", 0) + stats.get("n_ee_culled", 0)),
                           value_counts_culled_pairs=True,
                           history=0,
                           history_note="value / ms_per_step: SCCD_OPT_TOI_GUESS = 0 -- no call uses anything of the call before it (neither the TOI bound nor the choice "
                                        "between one and two narrow launches per pass: with history off every call runs the two halves of time, a bet on an impact before 0.5 -- "
                                        "`--cliffs` late_impact prices the bet where it is lost)",
                           culled_note="overlap pairs the projection cull (csrc/narrow_cull.inc) dropped before the bisection: provably no impact; they count as answered queries"),
            # schema 2 (round 4): min_toi_latency_ms is ONE thing again -- the step's latency on a device-resident mesh, = ms_per_step,
            # as in rounds 1-2 and for any number of ranks; the reference-shaped call from host matrices is host_path_ms only
            # schema 3 (round 5): value / ms_per_step are steps that start from toi = 1 (SCCD_OPT_TOI_GUESS = 0, as ccd.cu:125); every
            # timed step is also clocked on its own (ms_per_step_p50 / p99 / min / max: host clock around a step, which ends with its TOI
            # on the host); the speculative TOI bound has its own block (toi_guess.ms_per_step_with)
            # schema 4 (round 6): what is credited stands at the top level -- ms_per_step (mean, the driver's figure), its median, the
            # DEVICE's own span per step beside the host's clock, inclusion checks per second (the work that is left: `value` counts the
            # culled pairs as answered queries), and the lost bet of the two halves (late_impact)
            "schema": 4,
            "checks_per_s": float(qq[1].item()) / (dt / args.steps),
            "device_span_ms": {"p50": round(sorted(dev_ms)[len(dev_ms) // 2], 4), "p99": round(sorted(dev_ms)[min(len(dev_ms) - 1, int(len(dev_ms) * 0.99))], 4),
                               "max": round(max(dev_ms), 4), "mean": round(sum(dev_ms) / len(dev_ms), 4),
                               "note": "the device's real-time clock from the start of a step's first kernel to the end of the kernel behind the last read-back the host "
                                       "waited for (SCCD_OPT_DEVICE_SPAN_NS); host clock minus this = what the host added in front of the first launch and behind the last"},
            "slowest_steps": [{"step": i, "host_ms": round(step_ms[i], 4), "device_ms": round(dev_ms[i], 4)}
                              for i in sorted(range(len(step_ms)), key=lambda k: -step_ms[k])[:3]],
            "host_waits_per_step": host_waits_per_step,
            "late_impact": late_headline,
            "ms_per_step_p50": round(sorted(step_ms)[len(step_ms) // 2], 4),
            "ms_per_step_p99": round(sorted(step_ms)[min(len(step_ms) - 1, int(len(step_ms) * 0.99))], 4),
            "ms_per_step_min": round(min(step_ms), 4), "ms_per_step_max": round(max(step_ms), 4),
            "ms_first_steps": [round(x, 4) for x in step_ms[:5]],
            "min_toi_latency_ms": dt / args.steps * 1e3,
            "host_path_ms": host_ms,
            "host_path_note": "sccd_ccd() from pageable host matrices, upload and packing inside the call (best of 5; the reference's ccd() uploads inside the call too, ccd.cu:103-106); one rank only",
            "max_iter": args.max_iter,
            "broad_phase": broad,
            "toi_guess": dict(guess, ms_per_step_without=round(dt / args.steps * 1e3, 4),
                              note="history (SCCD_OPT_TOI_GUESS = 1, the library's default; OFF in value / ms_per_step): a step starts from "
                                   "1.125 x the previous step's TOI on the same mesh and is redone from 1 if nothing is found below the bound, and it runs one narrow "
                                   "launch per pass instead of two when the previous step found nothing before 0.5 (exact "
                                   "either way).  On this FROZEN mesh every bound is the step's own previous answer: ms_per_step_with is an upper bound "
                                   "on what the prior can buy, not a headline; `--jitter` lines show it on a mesh that moves"),
            "cliffs": cliffs,
            "rank_max": rank_max,
            "roofline": roofline,
        }
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, scenes)
    elif args.workload == "boxes1m":
        result = bench_boxes(args, ctx, sccd, scenes, torch)
    else:
        result = bench_sort(args, ctx, sccd, torch)

    if use_dist:
        dist.destroy_process_group()
    if rank == 0:  # the ONE JSON line is the last thing on stdout (RCCL prints a banner through C stdio: flush that first)
        sys.stdout.flush()
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(result), flush=True)


def bench_jitter(args, ctx, sccd, torch, mesh, tV0, tV1, params, wl, world):
    """A step that looks like a simulation (the reference's caller moves the mesh between ccd() calls,
    cuda/ipc_ccd_strategy.cu:97-152): before every step the end positions are replaced by one of eight device-resident
    variants V1 + A x noise (sccd_mesh_update_vertices, INSIDE the timed region: a caller pays it), so the entry counts, the key
    width and the pair counts change from step to step and the speculative build's guesses (api.hip bp_build) can miss."""
    n_var = 8
    g = torch.Generator(device=tV1.device).manual_seed(12345)
    amps = [args.jitter * (args.jitter_alternate if (k & 1) else 1.0) for k in range(n_var)]
    def noise(k):
        r = torch.rand(tV1.shape, generator=g, dtype=tV1.dtype, device=tV1.device) * 2.0 - 1.0
        if args.jitter_fraction < 1.0:  # tV1 is [3][nV]: one mask per vertex
            r = r * (torch.rand((1, tV1.shape[1]), generator=g, dtype=tV1.dtype, device=tV1.device) < args.jitter_fraction)
        return amps[k] * r

    variants = [tV1 + noise(k) for k in range(n_var)]
    torch.cuda.synchronize()

    def step(k):
        mesh.update_vertices(tV0.data_ptr(), variants[k % n_var].data_ptr(), on_device=True)
        return sccd.ccd_mesh(mesh, want_stats=True, **params)

    kk = [0]

    def one():
        kk[0] += 1
        step(kk[0])

    clock_warmup(one, args.clock_warmup)
    for k in range(max(args.warmup, n_var)):
        step(k)
    # the update alone (pack kernel + the synchronisation that releases the caller's buffers)
    ctx.synchronize()
    tu = time.perf_counter()
    for k in range(20):
        mesh.update_vertices(tV0.data_ptr(), variants[k % n_var].data_ptr(), on_device=True)
    ctx.synchronize()
    update_ms = (time.perf_counter() - tu) / 20 * 1e3
    step(0)
    ctx.set_option(sccd.OPT_SPEC_HITS, 0)
    ctx.set_option(sccd.OPT_TOI_GUESS_HITS, 0)
    steps = max(args.steps, 200)
    times, missed, tois, queries, allocs, toi_missed, qs = [], [], [], 0, [], [], []
    ctx.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        m0 = ctx.get_option(sccd.OPT_SPEC_MISSES)
        a0 = ctx.get_option(sccd.OPT_ALLOC_COUNT)
        g0 = ctx.get_option(sccd.OPT_TOI_GUESS_MISSES)
        ts = time.perf_counter()
        toi, st = step(k + 1)
        times.append((time.perf_counter() - ts) * 1e3)
        missed.append(ctx.get_option(sccd.OPT_SPEC_MISSES) - m0)
        allocs.append(ctx.get_option(sccd.OPT_ALLOC_COUNT) - a0)
        toi_missed.append(ctx.get_option(sccd.OPT_TOI_GUESS_MISSES) - g0)
        tois.append(toi)
        qs.append(st["n_vf_pairs"] + st["n_ee_pairs"])
        queries += st["n_vf_pairs"] + st["n_ee_pairs"]
    ctx.synchronize()
    dt = time.perf_counter() - t0
    hits, misses = ctx.get_option(sccd.OPT_SPEC_HITS), ctx.get_option(sccd.OPT_SPEC_MISSES)
    ts_sorted = sorted(times)
    hit_t = [t for t, m in zip(times, missed) if m == 0]
    miss_t = [t for t, m in zip(times, missed) if m > 0]
    med = lambda v: (sorted(v)[len(v) // 2] if v else None)  # noqa: E731
    return {
        "metric": "CCD queries/sec (broad+narrow), moving mesh", "value": queries / dt, "unit": "queries/s", "n_gpus": world,
        "steps": steps, "warmup": max(args.warmup, n_var), "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": dict(wl, jitter=args.jitter, jitter_alternate=args.jitter_alternate, jitter_fraction=args.jitter_fraction, variants=n_var, arith=["strict", "fma"][args.arith],
                       note="end positions + jitter x U(-1, 1)^3 per vertex, eight variants in turn; the vertex update is inside the timed region"),
        "p50_ms": round(ts_sorted[len(ts_sorted) // 2], 4), "p99_ms": round(ts_sorted[min(len(ts_sorted) - 1, int(len(ts_sorted) * 0.99))], 4),
        "min_ms": round(ts_sorted[0], 4), "max_ms": round(ts_sorted[-1], 4),
        "update_vertices_ms": round(update_ms, 4),
        "spec_builds": hits + misses, "spec_hits": hits, "spec_misses": misses,
        "spec_hit_rate": round(hits / max(1, hits + misses), 4),
        "toi_guess_hits": ctx.get_option(sccd.OPT_TOI_GUESS_HITS), "toi_guess_misses": ctx.get_option(sccd.OPT_TOI_GUESS_MISSES),
        "steps_with_a_miss": len(miss_t), "median_ms_hit": med(hit_t), "median_ms_miss": med(miss_t),
        "miss_over_hit": (round(med(miss_t) / med(hit_t), 3) if miss_t and hit_t else None),
        "toi_min": min(tois), "toi_max": max(tois), "queries_per_step": queries / steps,
        # what a miss of the speculative TOI bound costs (the narrow phases redone from 1), and the slow steps attributed
        "median_ms_toi_hit": med([t for t, g in zip(times, toi_missed) if g == 0]),
        "median_ms_toi_miss": med([t for t, g in zip(times, toi_missed) if g > 0]),
        "steps_with_a_toi_miss": sum(1 for g in toi_missed if g > 0),
        "steps_that_allocated": sum(1 for a in allocs if a > 0),
        "median_ms_alloc_step": med([t for t, a in zip(times, allocs) if a > 0]),
        "slowest_steps": [{"step": i, "ms": round(times[i], 3), "allocs": allocs[i], "spec_miss": missed[i], "toi_miss": toi_missed[i], "queries": qs[i]}
                          for i in sorted(range(len(times)), key=lambda i: -times[i])[:8]],
    }


def bench_boxes(args, ctx, sccd, scenes, torch):
    """BASELINE configs[2]: 1M random AABBs, broad phase only."""
    n = args.boxes_n
    boxes = scenes.random_boxes(n, seed=42, max_extent=0.027 * (1_000_000 / n) ** (1.0 / 3.0))
    if args.boxes_variant == "thin":  # cloth-like: boxes 100x thinner in z, centres unchanged
        cz = (boxes["min"][:, 2] + boxes["max"][:, 2]) / 2
        hz = (boxes["max"][:, 2] - boxes["min"][:, 2]) / 2 * 0.01
        boxes["min"][:, 2], boxes["max"][:, 2] = cz - hz, cz + hz
    dboxes = sccd.DeviceAABBs(boxes, ctx)
    bp = sccd.BroadPhase(ctx)

    def step():
        bp.build(dboxes)
        return bp.detect_overlaps_partial()[1]

    clock_warmup(step, args.clock_warmup)
    for _ in range(args.warmup):
        step()
    ctx.set_option(sccd.OPT_PROFILE, 1)
    ctx.reset_profile()
    ctx.synchronize()
    t0 = time.perf_counter()
    pairs = 0
    for _ in range(args.steps):
        pairs = step()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    prof = ctx.profile()
    ms_sweep, launches = prof["sweep"]
    achieved = (BYTES_SWEEP_PER_BOX * n + 8.0 * pairs) * args.steps / (ms_sweep * 1e-3) / 1e9
    # HBM bytes per sweep launch from the PMC passes of this build (tools/pmc_traffic.sh boxes1m): taken on the default
    # 1M isotropic boxes, quoted for those only
    tk = pmc_kernels("boxes1m") if (n == 1_000_000 and args.boxes_variant == "iso") else None
    kname = next((k for k in (tk or {}) if k.startswith("sweep_band_k")), None)
    traffic = tk[kname]["hbm_bytes_per_launch_corrected"] if kname else None
    cls = {k: round(v[0] / args.steps, 4) for k, v in prof.items() if v[0] > 0}
    return {
        "metric": "broad-phase boxes/sec", "value": n * args.steps / dt, "unit": "boxes/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%d random AABBs (%s), one list, broad phase only (BASELINE configs[2])" % (n, args.boxes_variant), "pairs": pairs,
                   "candidates": bp.candidates()},
        "broad_phase": {"bytes_per_step": BYTES_BROAD_PER_BOX * n + 8.0 * pairs, "formula": "548 B x boxes + 8 B x pairs (SURVEY 8d)",
                        "achieved": round((BYTES_BROAD_PER_BOX * n + 8.0 * pairs) * args.steps / dt / 1e9, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
                        "frac": round((BYTES_BROAD_PER_BOX * n + 8.0 * pairs) * args.steps / dt / 1e9 / HBM_PEAK_GBS, 5)},
        "roofline": {"bound": "hbm", "kernel": "sweep_band_k", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "avg_launch_ms": round(ms_sweep / max(1, launches), 4), "class_ms_per_step": cls,
                     "candidate_tests_per_s": bp.candidates() * args.steps / (ms_sweep * 1e-3)},
    }


def bench_sort(args, ctx, sccd, torch):
    """radix sort of 16M (u32 key, u32 index) pairs: the HBM-bound kernel of the broad phase."""
    n = 16_000_000
    g = torch.Generator(device="cuda").manual_seed(1)
    keys0 = torch.randint(0, 2**31 - 1, (n,), generator=g, dtype=torch.int32, device="cuda")
    vals0 = torch.arange(n, dtype=torch.int32, device="cuda")
    keys, vals = keys0.clone(), vals0.clone()

    def step():
        keys.copy_(keys0)
        vals.copy_(vals0)
        torch.cuda.synchronize()
        ctx.sort_pairs_u32(keys.data_ptr(), vals.data_ptr(), n)

    clock_warmup(step, args.clock_warmup)
    for _ in range(args.warmup):
        step()
    ctx.set_option(sccd.OPT_PROFILE, 1)
    ctx.reset_profile()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    prof = ctx.profile()
    ms, launches = prof["sort"]
    ok = bool((keys[1:] >= keys[:-1]).all().item())
    n_passes = 4
    achieved = BYTES_SORT_PER_KEY_PASS * 4 * n * args.steps / (ms * 1e-3) / 1e9
    tk = pmc_kernels("sort16m")  # HBM bytes of one sort (passes + histogram) from the PMC passes of this build
    traffic = None
    if tk and "os_pass_k" in tk and "os_hist_k" in tk:
        traffic = n_passes * tk["os_pass_k"]["hbm_bytes_per_launch_corrected"] + tk["os_hist_k"]["hbm_bytes_per_launch_corrected"]
    return {
        "metric": "radix sort keys/sec", "value": n * args.steps / (ms * 1e-3), "unit": "keys/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "16M (u32 key, u32 index) pairs, 4 x 8-bit LSD passes", "sorted": ok},
        "roofline": {"bound": "hbm", "kernel": "onesweep: os_hist_k + os_bases_k + os_pass_k x 4", "achieved": round(achieved, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "avg_launch_ms": round(ms / max(1, launches), 4), "launches": launches},
    }


def cpu_baseline(args, scenes):
    """The CPU oracle (kind 'port': the reference's CPU broad phase restated + the build's own
    CPU Tight-Inclusion; the reference has no CPU narrow phase) on all host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc

    cores = os.cpu_count() or 1
    if args.workload == "clothball10k":
        V0, V1, E, F = scenes.cloth_ball()
        sample = "the full 10k-tri cloth-ball scene"
    else:
        n = args.cpu_sample_n or args.cloth_n
        V0, V1, E, F = scenes.folded_cloth(n)
        sample = f"folded cloth {n}x{n} ({len(F)} tris): same generator as the GPU workload"
    orc.ccd(V0[:64], V1[:64], E[:1], F[:1], nthreads=cores)  # warm the thread pool
    t0 = time.perf_counter()
    toi, n_vf, n_ee = orc.ccd(V0, V1, E, F, 0.0, -1, 1e-6, True, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": (n_vf + n_ee) / dt, "unit": "queries/s", "cores": cores, "kind": "port", "sample": sample,
            "seconds": round(dt, 3), "toi": toi}


if __name__ == "__main__":
    main()
