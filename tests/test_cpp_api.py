"""The C++ mirror of the reference host API (include/scalable_ccd/hip/ccd.hpp): it compiles
against the C ABI on any machine, and on a GPU it passes its reference-style checks."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_ccd_api")


def test_cpp_header_compiles_and_links():
    subprocess.check_call(["make", "-s", "-C", ROOT, "cpptest"])
    assert os.path.exists(EXE)


def test_cpp_host_api_refuses_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    subprocess.check_call(["make", "-s", "-C", ROOT, "cpptest"])
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_cpp_host_api_parity():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", ROOT, "cpptest"])
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failure(s)" in r.stdout
