"""bench.py's host-side helpers (no GPU): which PMC profile a run may quote."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "scalable-ccd_amd", "sccd", "libsccd_hip.so")


@pytest.mark.skipif(not os.path.exists(LIB), reason="libsccd_hip.so not built")
def test_device_code_hash_reads_the_fatbin_section(tmp_path):
    import bench

    h = bench.device_code_sha256(LIB)
    assert isinstance(h, str) and len(h) == 64 and h != bench.lib_sha256()
    junk = tmp_path / "not_elf.so"
    junk.write_bytes(b"hello")
    assert bench.device_code_sha256(str(junk)) is None


@pytest.mark.skipif(not os.path.exists(LIB), reason="libsccd_hip.so not built")
def test_a_profile_is_quoted_only_for_the_kernels_it_was_taken_on(tmp_path, monkeypatch):
    """A PMC file is admitted by the library's hash or by the hash of its device code (a host-only fix keeps the kernels the
    counters were read from), and by nothing else."""
    import bench

    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    dev, lib = bench.device_code_sha256(LIB), bench.lib_sha256()
    cases = {
        "same_lib.json": ({"lib_sha256": lib, "kernels": {"k": 1}}, True),
        "same_kernels.json": ({"lib_sha256": "0" * 64, "device_code_sha256": dev, "kernels": {"k": 2}}, True),
        "other_build.json": ({"lib_sha256": "0" * 64, "device_code_sha256": "1" * 64, "kernels": {"k": 3}}, False),
        "no_device_hash.json": ({"lib_sha256": "0" * 64, "kernels": {"k": 4}}, False),
    }
    for name, (body, admitted) in cases.items():
        (prof / name).write_text(json.dumps(body))
        got = bench._pmc_file(name)
        assert (got is not None) == admitted, name
    assert bench._pmc_file("missing.json") is None
