"""The C-ABI boundary without a GPU: the library loads, exports every symbol include/sccd.h
declares, and refuses to run (no CPU fallback) when there is no device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "sccd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sccd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(sccd):
    L = sccd.lib()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), f"libsccd_hip.so does not export {name}"
    assert sorted(sccd.ABI_SYMBOLS) == declared


def test_version_string(sccd):
    assert b"gfx950" in sccd.lib().sccd_version()


def test_box_layout_matches_reference_struct(sccd):
    # scalable_ccd::cuda::AABB: Scalar3 min, Scalar3 max, int3 vertex_ids, int element_id (aabb.cuh:82-92)
    d = sccd.AABB_DTYPE
    assert d.itemsize == 64
    assert [d.fields[k][1] for k in ("min", "max", "vertex_ids", "element_id")] == [0, 24, 48, 60]


def test_no_cpu_fallback(sccd):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = sccd.lib().sccd_create(0, C.byref(h))
    assert rc == -2 and not h  # SCCD_E_NO_DEVICE
    assert b"no HIP device" in sccd.lib().sccd_last_error(None)
    with pytest.raises(RuntimeError):
        sccd.Context(0)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "scalable-ccd_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".inc", ".cpp", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "sccd_oracle" not in text and "import orc" not in text and "oracle/" not in text, fn
