"""The C-ABI library builds for gfx950, loads, exports every symbol include/mtts.h declares, and the
product refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    return ge.build_device()


def test_header_symbols_are_exported(lib_path):
    hdr = open(os.path.join(ROOT, "include", "mtts.h")).read()
    declared = sorted(set(re.findall(r"\b(mtts_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    lib = ctypes.CDLL(lib_path)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    from meta_tts_amd import _lib
    assert sorted(_lib.EXPORTS) == declared  # the Python binding types exactly the header's surface


def test_device_code_object_is_gfx950(lib_path):
    data = open(lib_path, "rb").read()
    assert b"gfx950" in data
    assert b"v_mfma" not in data  # sanity: a binary, not text


def test_product_fails_loudly_without_library(tmp_path):
    from meta_tts_amd import _lib
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load(str(tmp_path / "libmtts.so"))


def test_create_fails_loudly_without_gpu(lib_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from meta_tts_amd.config import ModelDims
    from meta_tts_amd.engine import Engine, MttsError
    with pytest.raises(MttsError):
        Engine(ModelDims(), max_tasks=1, max_B=1, max_S=8, max_T=16)
