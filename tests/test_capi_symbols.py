"""The C-ABI library loads and exports every function include/msm377.h declares. CPU only
(no compute call is made: without a GPU only host-only entry points may run)."""
import ctypes
import os
import re

import pytest

import webgpu_msm_bls12_377_amd as msm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    with open(os.path.join(ROOT, "include", "msm377.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msm377_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ("msm377_g1_msm", "msm377_g1_msm_device", "msm377_ctx_create", "msm377_ctx_destroy", "msm377_g1_combine_partials"):
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(msm.library_path())
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_version_and_strerror():
    lib = msm.load_library()
    assert lib.msm377_version().decode().startswith("msm377")
    assert lib.msm377_strerror(0).decode() == "ok"
    assert "HIP" in lib.msm377_strerror(-2).decode()


def test_no_cpu_fallback_without_a_device():
    """Product path must fail loudly, never compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(msm.MsmError) as e:
        msm.MsmEngine(1 << 10)
    assert e.value.code == -2


def test_product_does_not_touch_the_oracle():
    """Nothing under the package may import, link or name the oracle."""
    pkg = os.path.join(ROOT, "webgpu-msm-bls12-377_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cc", ".cpp", ".js", ".ts", "Makefile")):
                with open(os.path.join(dirpath, fn), errors="ignore") as f:
                    text = f.read()
                assert "libmsm_oracle" not in text and "oracle/" not in text.replace("the oracle/", ""), os.path.join(dirpath, fn)
