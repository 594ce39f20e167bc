"""The float arithmetic of the library's SCCD_OPT_SCALAR = 1 path (csrc/ti_math_f32.hpp) is plain C++: compile
it with the HOST compiler and compare it with the CPU oracle's float twin (orc_*_f32) -- per-query constants,
single inclusion-function evaluations and whole queries, bit for bit.  No GPU needed."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_float_arithmetic_header_matches_the_oracle_twin(orc, tmp_path):
    orc.lib()  # builds oracle/libsccd_oracle.so if needed
    exe = str(tmp_path / "test_ti_f32_host")
    subprocess.check_call([
        "g++", "-std=c++17", "-O1", "-ffp-contract=off", "-mfma", "-Wall", "-Wextra",
        "-I" + os.path.join(ROOT, "scalable-ccd_amd", "csrc"), "-I" + os.path.join(ROOT, "oracle"),
        os.path.join(ROOT, "tests", "cpp", "test_ti_f32_host.cpp"), "-o", exe,
        "-L" + os.path.join(ROOT, "oracle"), "-lsccd_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
    ])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "0 failure(s)" in out.stdout
