"""CPU: the C-ABI library builds, loads, exports every symbol include/ispk.h declares, and validates arguments
without touching a GPU (argument checks run before any launch)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

from isp_tts_amd import runtime


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "ispk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ispk_[a-z0-9_]+)\s*\(", src)))


def test_build_entry_point_runs():
    import __graft_entry__
    __graft_entry__.build()
    assert os.path.exists(runtime.LIB_PATH)
    assert os.path.exists(os.path.join(ROOT, "oracle", "_build", "libmas_oracle.so"))


def test_every_header_symbol_is_exported_and_bound():
    syms = _header_symbols()
    assert len(syms) >= 12
    assert syms == sorted(runtime.SIGNATURES), "runtime.SIGNATURES must list exactly the header's entry points"
    handle = ctypes.CDLL(runtime.LIB_PATH)
    for s in syms:
        assert hasattr(handle, s), f"{s} is declared in include/ispk.h but missing from libispk.so"
    assert runtime.lib().ispk_abi_version() == 2


def test_argument_errors_without_gpu():
    lib = runtime.lib()
    E_NULL, E_SHAPE, E_ALIGN, E_UNSUP = -1, -2, -3, -4
    assert lib.ispk_mas_f32(None, None, None, None, None, None, 1, 8, 8, 64, 8, None) == E_NULL
    assert b"null" in lib.ispk_last_error_string()
    one = ctypes.c_void_p(16)  # never dereferenced: shape checks fail first
    assert lib.ispk_mas_f32(one, one, one, one, None, None, 1, 8, 513, 8 * 513, 513, None) == E_SHAPE
    assert b"512" in lib.ispk_last_error_string()
    assert lib.ispk_mas_f32(one, one, one, one, None, None, 1, 4096, 512, 4096 * 512, 512, None) == E_SHAPE
    assert b"LDS" in lib.ispk_last_error_string()
    assert lib.ispk_gemm_f32(one, 12, one, 12, one, 8, None, None, 0, None, 4, 8, 12, 0, 0, 0, None) == E_SHAPE
    assert lib.ispk_gemm_f32(one, 18, one, 16, one, 8, None, None, 0, None, 4, 8, 16, 0, 0, 0, None) == E_ALIGN
    assert lib.ispk_gemm_f32(one, 16, one, 16, one, 8, None, None, 0, None, 4, 8, 16, 4, 0, 0, None) == E_NULL  # mask flag
    assert lib.ispk_gemm_f32(one, 16, one, 16, one, 8, None, None, 0, None, 4, 8, 16, 64, 0, 0, None) == E_UNSUP
    assert lib.ispk_layernorm_f32(one, 100, None, None, None, None, 0, 1, None, one, 100, 4, 100, 1e-5, None) == E_SHAPE
    assert lib.ispk_alibi_mqa_attn_f32(one, 384, one, one, 128, one, None, one, 384, 2, 16, 9, None) == E_SHAPE
    assert lib.ispk_alibi_mqa_attn_f32(one, 384, one, one, 126, one, None, one, 384, 2, 16, 6, None) == E_ALIGN
    assert lib.ispk_linear_small_f32(None, 1, None, 1, None, None, 0, None, 1, 1, 1, 1, 0, None) == E_NULL
    assert lib.ispk_gemm_f32_tile(32768, 384, 384) == 22 and lib.ispk_gemm_f32_tile(6400, 384, 384) == 12


def test_zero_sized_batches_are_noops():
    lib = runtime.lib()
    one = ctypes.c_void_p(16)
    assert lib.ispk_mas_f32(one, one, one, one, None, None, 0, 8, 8, 64, 8, None) == 0
    assert lib.ispk_gemm_f32(one, 16, one, 16, one, 8, None, None, 0, None, 0, 8, 16, 0, 0, 0, None) == 0
    assert lib.ispk_layernorm_f32(one, 64, None, None, None, None, 0, 1, None, one, 64, 0, 64, 1e-5, None) == 0


def test_product_fails_loudly_without_gpu_or_library(monkeypatch, tmp_path):
    import torch
    with pytest.raises(runtime.IspkError, match="GPU tensors"):
        runtime.gemm(torch.zeros(8, 64), torch.zeros(16, 64))
    with pytest.raises(runtime.IspkError, match="GPU tensors"):
        runtime.mas(torch.zeros(1, 4, 4), torch.tensor([4]), torch.tensor([4]))
    monkeypatch.setattr(runtime, "_lib", None)
    monkeypatch.setattr(runtime, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(runtime.IspkError, match="not built"):
        runtime.lib()
