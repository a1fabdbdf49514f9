"""The C-ABI library loads and exports every symbol include/vo_hip.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "vo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from visual_odometry_amd import capi
    names = _declared_functions()
    assert len(names) >= 20 and "vo_create" in names and "vo_batch_submit" in names
    lib = ctypes.CDLL(capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in vo_hip.h but not exported"
    # the Python binding covers the same set
    assert sorted(capi.EXPORTS) == names
    assert capi.load().vo_abi_version() == 1
    assert capi.load().vo_batch_stage_count() == 12


def test_struct_layouts_match_header():
    from visual_odometry_amd import capi
    assert capi.KP_DTYPE.itemsize == 24 and capi.KNN_DTYPE.itemsize == 16 and capi.MATCH_DTYPE.itemsize == 28
    assert capi.RESULT_DTYPE.itemsize == 12 * 4 + 6 * 8
    assert ctypes.sizeof(capi.VoCfg) == 32


def test_no_gpu_means_loud_failure():
    """Without a usable GPU the product refuses to run; it never falls back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from visual_odometry_amd import capi
    with pytest.raises(capi.VoError):
        capi.Context(device=0, max_width=64, max_height=64, max_features=10, max_batch=1)


def test_bad_configuration_rejected():
    from visual_odometry_amd import capi
    lib = capi.load()
    cfg = capi.VoCfg(8192, 8192, 100, 1, 0, 0, 0, 0)  # beyond 2^24 pixels
    assert not lib.vo_create(0, ctypes.byref(cfg))
    assert b"bad configuration" in lib.vo_last_error(None)


def test_product_does_not_import_oracle():
    """Only tests/, smoke() and bench's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "visual_odometry_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"
    for dirpath, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "oracle" not in open(os.path.join(dirpath, f), errors="ignore").read().lower()
