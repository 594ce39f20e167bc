"""The hot kernels' register budgets, checked on the cross-compiled code (no GPU needed): np_walk_k runs three waves per SIMD
only below 171 VGPRs and has no scalar registers to spare -- a change that tips it into scratch spills costs more than it can
gain (and one such build hung the GPU in round 2), so the build is checked, not trusted."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only"]


def _kernels(src, tmp_path):
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, *FLAGS, os.path.join(ROOT, "scalable-ccd_amd", "csrc", src + ".hip"), "-o", str(out)],
                   check=True, stderr=subprocess.DEVNULL, timeout=600)
    text = out.read_text()
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        body = m.group(2)
        res[m.group(1)] = {k: int(re.search(r"\.amdhsa_" + k + r" (\d+)", body).group(1))
                           for k in ("private_segment_fixed_size", "next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size")}
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_walk_kernels_keep_three_waves_per_simd_without_scratch(tmp_path):
    ks = _kernels("narrow", tmp_path)
    plain = {k: v for k, v in ks.items() if re.match(r"_Z9np_walk_kILb[01]ELi[01]ELi0EE", k)}
    assert len(plain) == 4  # VF / EE x strict / fused arithmetic
    for name, r in plain.items():
        assert r["private_segment_fixed_size"] == 0, (name, r)
        assert r["next_free_vgpr"] <= 168, (name, r)  # 512 / 3 waves, allocation granule 8
        assert r["group_segment_fixed_size"] <= 13 * 1024, (name, r)  # twelve one-wave blocks per CU
    # the float build's depth-first kernel: three waves per SIMD or more, no scratch
    f32 = {k: v for k, v in ks.items() if "np_walk_f32_k" in k}
    assert len(f32) == 8
    for name, r in f32.items():
        assert r["private_segment_fixed_size"] == 0 and r["next_free_vgpr"] <= 168, (name, r)
    # the projection cull (narrow_cull.inc): three waves per SIMD, no scratch (spilling builds at four and five waves were 1.5 x and 5 x slower)
    cull = {k: v for k, v in ks.items() if "np_cull_k" in k}
    assert len(cull) == 4  # VF / EE x double / float build
    for name, r in cull.items():
        assert r["private_segment_fixed_size"] == 0 and r["next_free_vgpr"] <= 168, (name, r)
        assert r["group_segment_fixed_size"] <= 48 * 1024, (name, r)  # three four-wave blocks per CU
    # the level-order kernel: blocks of 1,024 threads (one atomic per block on the next level's length: narrow.hip) -- 128 registers at
    # most, no scratch, and its dynamically indexed private arrays (kept in LDS by the compiler, 56 bytes per thread) inside a block's 64 KB
    level = {k: v for k, v in ks.items() if "np_level_kIL" in k}
    assert len(level) == 8  # VF / EE x strict / fused x double / float build
    for name, r in level.items():
        assert r["private_segment_fixed_size"] == 0 and r["next_free_vgpr"] <= 128 and r["group_segment_fixed_size"] <= 64 * 1024, (name, r)
    # the bookkeeping kernels (two waves per SIMD) must not spill either
    book = {k: v for k, v in ks.items() if re.match(r"_Z9np_walk_kILb[01]ELi[01]ELi1EE", k)}
    for name, r in book.items():
        assert r["private_segment_fixed_size"] == 0 and r["next_free_vgpr"] <= 256, (name, r)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_kernel_uses_scratch_memory(tmp_path):
    # every kernel of the library: a spill, or an array indexed by a run-time value (`b.lo[g.aa]` put 56 bytes per thread of
    # the two-list fill into scratch until round 3: 19 us of a 1.3 ms step), shows up here
    for src in ("sort", "sweep", "boxes", "scan", "api", "build", "drivers"):
        for name, r in _kernels(src, tmp_path).items():
            assert r["private_segment_fixed_size"] == 0, (name, r)
            if "sweep_band_k" in name:
                # two four-wave blocks per CU by LDS (78 KB each): two waves per SIMD, so 256 VGPRs are the budget
                assert r["next_free_vgpr"] <= 256, (name, r)
                assert r["group_segment_fixed_size"] <= 80 * 1024, (name, r)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_readback_gather_kernel_fits_beside_narrow_phase_waves(tmp_path):
    # three np_walk_k waves of 168 VGPRs leave 8 of a SIMD lane's 512: the read-back's gather kernel (api.hip) must fit into
    # those, or a read-back issued beside the narrow phase waits for a wave to retire (profiles/HISTORY.md 5.6)
    ks = {k: v for k, v in _kernels("api", tmp_path).items() if "readback_gather_k" in k}
    # ... and so must the kernel that starts a narrow launch's counters (the helper issues it beside the vertex-face kernel)
    ks.update({k: v for k, v in _kernels("narrow", tmp_path).items() if "np_counters_init_k" in k})
    assert len(ks) == 2
    for name, r in ks.items():
        assert r["next_free_vgpr"] <= 8 and r["group_segment_fixed_size"] == 0, (name, r)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_edge_records_kernel_fits_beside_the_vertex_face_sweep(tmp_path):
    # ccd() runs the edge list's records kernel beside the vertex-face sweep (the records gate): a block of it must fit
    # on a CU that holds two sweep waves per SIMD -- 512 registers per SIMD lane minus the sweep's two waves, shared by the block's
    # waves on that SIMD (ER_THREADS / 256); with blocks that do not fit the gate costs 15 us instead of gaining 20
    boxes = _kernels("boxes", tmp_path)
    sweep = _kernels("sweep", tmp_path)
    rec = [v for k, v in boxes.items() if "entry_record_kILi0E" in k]
    vf = [v for k, v in sweep.items() if re.search(r"sweep_band_kILb0ELi3E", k)]
    assert len(rec) == 1 and len(vf) == 1
    granule = lambda v: (v + 7) // 8 * 8
    left = 512 - 2 * granule(vf[0]["next_free_vgpr"])
    src = open(os.path.join(ROOT, "scalable-ccd_amd", "csrc", "boxes.hip")).read()
    threads = int(re.search(r"#define ER_THREADS_ (\d+)", src).group(1))
    waves_per_simd = threads // 256
    assert waves_per_simd * granule(rec[0]["next_free_vgpr"]) <= left, (rec[0], vf[0], threads)
    assert rec[0]["group_segment_fixed_size"] <= 160 * 1024 - 2 * vf[0]["group_segment_fixed_size"]  # (the sweep's two blocks take 153 of a CU's 160 KB of LDS: 6.9 KB are left)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_sort_pass_keeps_two_blocks_per_cu(tmp_path):
    # os_pass_k's tile is ranked by eight waves; radix_sort_pairs_u32 launches a block per tile, tiles by block index, while
    # num_tiles <= 2 blocks per CU (`pass_blocks`) -- which assumes two 512-thread blocks ARE resident per CU: four waves per SIMD
    # (<= 128 registers), <= 80 KB of LDS each
    ks = {k: v for k, v in _kernels("sort", tmp_path).items() if "os_pass_k" in k}
    assert len(ks) == 1
    src = open(os.path.join(ROOT, "scalable-ccd_amd", "csrc", "sort.hip")).read()
    threads = int(re.search(r"#define RS_THREADS_ (\d+)", src).group(1))
    blocks = int(re.search(r"constexpr int pass_blocks = (\d+);", src).group(1))
    for name, r in ks.items():
        waves_per_simd = blocks * threads // 256
        assert waves_per_simd * ((r["next_free_vgpr"] + 7) // 8 * 8) <= 512, (name, r, threads, blocks)
        assert blocks * r["group_segment_fixed_size"] <= 160 * 1024, (name, r)
        assert r["private_segment_fixed_size"] == 0, (name, r)
