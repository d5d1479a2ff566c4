"""ctypes binding of libispk.so (the C ABI declared in include/ispk.h) and thin tensor-level wrappers.

PyTorch here is plumbing only: it owns device memory and the stream.  Every wrapper passes raw
`data_ptr()`s and the CURRENT torch stream to the library, so launches are ordered with torch ops and are
capturable in a HIP graph.  There is no CPU fallback anywhere: a missing library or a non-GPU tensor raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch
from torch import Tensor

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libispk.so")

EP_GELU, EP_SILU, EP_MASK_ACC, EP_MASK_OUT, EP_BIAS_ROW, EP_MASK_COL, EP_OUT_BF16, EP_RESID_BF16, EP_ROWS_T, EP_OUT_SPLIT = (
    1, 2, 4, 8, 16, 32, 64, 128, 256, 512)

_P, _I32, _I64, _U32, _F32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_float
_U64 = ctypes.c_uint64

# name -> argtypes; must list EVERY symbol of include/ispk.h (tests/test_abi.py checks header == this table == .so)
SIGNATURES = {
    "ispk_abi_version": [],
    "ispk_last_error_string": [],
    "ispk_device_info": [ctypes.c_char_p, _I32],
    "ispk_mas_f32": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I64, _I64, _P],
    "ispk_layernorm_f32": [_P, _I64, _P, _P, _P, _P, _I64, _I32, _P, _P, _I64, _I32, _I32, _F32, _P],
    "ispk_layernorm_f32_bf16": [_P, _I64, _P, _P, _P, _P, _I64, _I32, _P, _P, _I64, _I32, _I32, _F32, _P],
    "ispk_gemm_f32_tile": [_I32, _I32, _I32],
    "ispk_gemm_f32": [_P, _I64, _P, _I64, _P, _I64, _P, _P, _I64, _P, _I32, _I32, _I32, _U32, _I32, _I64, _P],
    "ispk_gemm_f32_batched": [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I32, _I32, _I32, _I32, _P],
    "ispk_gemm_bf16_gelu_train": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I32, _I32, _I32, _F32, _U64, _P],
    "ispk_gemm_bf16_gelu_bwd": [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _I32, _I32, _I32, _F32, _U64, _P],
    "ispk_segments_f32": [_P, _I32, _P],
    "ispk_stage_weights": [_P, _I32, _P],
    "ispk_fill_zero": [_P, _I64, _P],
    "ispk_scale_f32": [_P, _I64, _P, _F32, _P],
    "ispk_sum_scalars_f32": [_P, _P, _I32, _P, _P],
    "ispk_exp_pad_f32": [_P, _P, _I32, _I32, _P],
    "ispk_sqrt_scale_f32": [_P, _P, _I32, _F32, _P],
    "ispk_copy2d_f32": [_P, _I64, _P, _I64, _I32, _I32, _P],
    "ispk_permute021_f32": [_P, _P, _I32, _I32, _I32, _P],
    "ispk_conv_weight_flip_f32": [_P, _P, _I32, _I32, _I32, _P],
    "ispk_gemm_bf16_last_variant": [],
    "ispk_gemm_bf16": [_P, _I64, _P, _I64, _P, _I64, _P, _P, _I64, _P, _I32, _I32, _I32, _U32, _I32, _I64, _P],
    "ispk_gemm_bf16_splitk_plan": [_I32, _I32, _I32, _U32],
    "ispk_gemm_bf16_splitk": [_P, _I64, _P, _I64, _P, _I64, _P, _P, _I64, _P, _I32, _I32, _I32, _U32, _P, _I32, _P],
    "ispk_ffn_bf16": [_P, _I64, _P, _I64, _P, _P, _I64, _P, _P, _I64, _P, _P, _I64, _I32, _I32, _I32, _U32, _P],
    "ispk_ffn_pack_w2_bf16": [_P, _I64, _I32, _I32, _P, _P],
    "ispk_ffn_bf16_prenorm": [_P, _I64, _P, _P, _F32, _P, _I64, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _U32, _P, _F32, _P],
    "ispk_ffn_chunk_w2_bf16": [_P, _I64, _I32, _I32, _P, _P],
    "ispk_ffn_bf16_prenorm2_split": [_P, _I64, _P, _P, _F32, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P],
    "ispk_ffn_combine_ln_f32": [_P, _I64, _P, _I64, _I32, _P, _P, _I64, _P, _P, _F32, _I32, _P, _I64, _I32, _I32, _I32, _P],
    "ispk_ffn_bf16_prenorm2": [_P, _I64, _P, _P, _F32, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _U32, _P, _F32, _P],
    "ispk_attn_out_ffn_bf16": [_P, _I64, _P, _I64, _P, _P, _P, _F32, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _U32, _P, _F32, _P],
    "ispk_attn_out_ffn_qkv_bf16": [_P, _I64, _P, _I64, _P, _P, _P, _F32, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _U32, _P, _P, _F32,
                                   _P, _P, _I64, _P],
    "ispk_chunk_k16_bf16": [_P, _I64, _I32, _I32, _P, _P],
    "ispk_attn_out_ffn_norm_bf16": [_P, _I64, _P, _I64, _P, _P, _P, _F32, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _U32, _P, _P, _F32,
                                    _I32, _P, _I64, _I32, _P],
    "ispk_attn_out_ffn_split_bf16": [_P, _I64, _P, _I64, _P, _P, _P, _F32, _P, _P, _P, _U32, _P, _I64, _I32, _I32, _I32, _I32, _P],
    "ispk_gemm_bf16_lnin": [_P, _I64, _P, _P, _P, _F32, _P, _I64, _P, _I64, _P, _P, _I64, _P, _I32, _I32, _I32, _U32, _P],
    "ispk_linear_small_f32": [_P, _I64, _P, _I64, _P, _P, _I64, _P, _I64, _I32, _I32, _I32, _U32, _P],
    "ispk_alibi_mqa_attn_f32": [_P, _I64, _P, _P, _I64, _P, _P, _P, _I64, _I32, _I32, _I32, _P],
    "ispk_alibi_mqa_attn_bf16": [_P, _I64, _P, _P, _I64, _P, _P, _P, _I64, _I32, _I32, _I32, _P],
    "ispk_alibi_mqa_attn_bf16_tiles": [_P, _I64, _P, _P, _I64, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P],
    "ispk_cast_f32_bf16": [_P, _I64, _P, _I64, _I32, _I32, _P],
    "ispk_split_f16": [_P, _I64, _P, _P, _I64, _I32, _I32, _P],
    "ispk_gemm_split_f16_tile": [_I32, _I32, _I32],
    "ispk_gemm_split_f16": [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _P, _P, _I64, _P, _I32, _I32, _I32, _U32, _I32,
                            _I64, _P],
    "ispk_layernorm_f32_split": [_P, _I64, _P, _P, _P, _P, _I64, _I32, _P, _P, _I64, _I64, _I32, _I32, _F32, _P],
    "ispk_alibi_mqa_attn_split_f16": [_P, _I64, _P, _P, _I64, _P, _P, _P, _I64, _I64, _I32, _I32, _I32, _P],
    "ispk_flow_mix_f32": [_P, _P, _P, _F32, _P, _P, _I32, _I32, _I32, _P],
    "ispk_flow_finish_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P],
    "ispk_flow_head_f32": [_P, _I64, _P, _P, _F32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P],
    "ispk_flow_euler_f32": [_P, _P, _F32, _P, _P, _I32, _I32, _I32, _P],
    "ispk_infer_features_f32": [_P, _P, _P, _P, _P, _F32, _F32, _F32, _F32, _F32, _P, _P, _I32, _I32, _P],
    "ispk_embed_tokens_f32": [_P, _P, _I64, _I32, _P, _P, _P, _I32, _I32, _I32, _P],
    "ispk_add_speaker_f32": [_P, _P, _I64, _I32, _P, _I32, _I32, _I32, _I32, _P],
    "ispk_time_embedding_f32": [_P, _I32, _P, _P, _I32, _P, _P, _P, _P, _I32, _P, _P],
    "ispk_length_regulate_f32": [_P, _P, _P, _P, _P, _I64, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "ispk_length_regulate_split_bf16": [_P, _P, _P, _P, _P, _I64, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "ispk_length_regulate_split_f16": [_P, _P, _P, _P, _P, _I64, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P],
    "ispk_pad_rows_f32": [_P, _I64, _I64, _I64, _P, _P, _I32, _I32, _I32, _I32, _P],
    "ispk_masked_instnorm_f32": [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _F32, _P],
    "ispk_aligner_scores_f32": [_P, _I64, _P, _I64, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P],
    "ispk_aligner_scores_fast_f32": [_P, _I64, _P, _I64, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P],
    "ispk_soft_average_f32": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P],
    "ispk_transpose_f32": [_P, _I64, _P, _I64, _I32, _I32, _P],
    "ispk_gemm_tn_f32": [_P, _I64, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _I32, _P, _I64, _P],
    "ispk_gemm_tn_bf16": [_P, _I64, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _I32, _P, _I64, _P],
    "ispk_gemm_tn_b16": [_P, _I64, _P, _I64, _P, _I64, _I32, _I32, _I32, _P, _I32, _P, _I64, _P],
    "ispk_gelu_f32_bf16": [_P, _P, _I64, _F32, _U64, _P],
    "ispk_gelu_bf16": [_P, _P, _I64, _F32, _U64, _P],
    "ispk_gelu_bwd_b16": [_P, _P, _P, _I64, _F32, _U64, _P],
    "ispk_gelu_bwd_bf16": [_P, _P, _P, _I64, _F32, _U64, _P],
    "ispk_gemm_tn_batched_f32": [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I32, _I32, _I32, _I32, _P, _I32, _P, _I64, _P],
    "ispk_layernorm_bwd_f32": [_P, _I64, _P, _I64, _P, _P, _P, _I64, _I32, _P, _P, _P, _I64, _I64, _I32, _F32, _P],
    "ispk_layernorm_bwd_dual_f32": [_P, _I64, _P, _I64, _P, _P, _P, _I64, _I32, _P, _P, _P, _I64, _I64, _I32, _F32, _P, _I64, _P],
    "ispk_gelu_f32": [_P, _P, _I64, _F32, _U64, _P],
    "ispk_gelu_bwd_f32": [_P, _P, _P, _I64, _F32, _U64, _P],
    "ispk_dropout_mask_u8": [_P, _I64, _F32, _U64, _P],
    "ispk_alibi_mqa_attn_train_f32": [_P, _I64, _P, _P, _P, _I64, _P, _I32, _I32, _I32, _F32, _U64, _P],
    "ispk_alibi_mqa_attn_bwd_f32": [_P, _I64, _P, _P, _I64, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _P, _F32, _U64, _P],
    "ispk_alibi_mqa_attn_train_bf16": [_P, _I64, _P, _P, _P, _I64, _P, _I32, _I32, _I32, _F32, _U64, _P],
    "ispk_alibi_mqa_attn_bwd_bf16": [_P, _I64, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _F32, _U64, _P],
    "ispk_mel_loss_f32": [_P, _P, _P, _P, _P, _P, _F32, _I32, _I32, _I32, _P],
    "ispk_aligner_scores_bwd_f32": [_P, _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _I32, _I32, _I32, _F32, _P],
    "ispk_masked_instnorm_bwd_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _F32, _P],
    "ispk_soft_average_bwd_f32": [_P, _P, _P, _P, _P, _P, _I64, _P, _I32, _I32, _I32, _I32, _P],
    "ispk_flow_loss_bwd_f32": [_P, _P, _P, _F32, _P, _I32, _I32, _I32, _P],
    "ispk_adaln_bwd_f32": [_P, _I64, _P, _I64, _P, _I64, _P, _P, _I64, _I32, _P, _P, _I64, _I32, _I32, _I32, _F32, _P],
    "ispk_time_embedding_bwd_f32": [_P, _I32, _P, _P, _I32, _P, _P, _P, _I32, _P, _P, _P, _P, _P, _P],
    "ispk_attn_ctc_loss_f32": [_P, _P, _P, _F32, _P, _I64, _P, _P, _F32, _I32, _I32, _I32, _P],
    "ispk_attn_bin_loss_f32": [_P, _P, _F32, _P, _P, _P, _F32, _I32, _I32, _I32, _P],
    "ispk_mel_grad_rows_f32": [_P, _P, _P, _I32, _I32, _I32, _P],
    "ispk_colsum_f32": [_P, _I64, _I64, _I32, _P, _P, _I64, _P, _P],
    "ispk_smallk_wgrad_f32": [_P, _I64, _P, _I64, _I64, _I32, _I32, _P, _I64, _P, _P],
    "ispk_embedding_bwd_f32": [_P, _P, _I64, _I32, _I32, _I32, _P, _I64, _P],
    "ispk_grad_sqnorm_f32": [_P, _I64, _P, _P, _P],
    "ispk_adamw_f32": [_P, _P, _P, _P, _I64, _I64, _F32, _F32, _F32, _F32, _F32, _I32, _P, _F32, _F32, _P],
    "ispk_adamw_f32_dev": [_P, _P, _P, _P, _I64, _I64, _P, _P, _P],
    "ispk_adam_args_f32": [_F32, _F32, _F32, _F32, _F32, _I32, _F32, _F32, _P],
    "ispk_set_dropout_seed_source": [_P],
}

_lib = None


class IspkError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Loads libispk.so.  Fails loudly when it has not been built (`python -m isp_tts_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IspkError(f"{LIB_PATH} is missing: the HIP extension was not built "
                            f"(run `python -m isp_tts_amd.build` or `__graft_entry__.build()`); there is no CPU fallback")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_char_p if name == "ispk_last_error_string" else ctypes.c_int32
        if handle.ispk_abi_version() != 2:
            raise IspkError(f"libispk.so ABI version {handle.ispk_abi_version()} != 2 (rebuild: python -m isp_tts_amd.build)")
        _lib = handle
    return _lib


class LaunchProfiler:
    """Optional per-launch timing with HIP events on the launch stream (used by bench.py for the roofline object).
    Each record: (kernel label, algorithmic FLOPs, algorithmic HBM bytes, start event, end event)."""

    def __init__(self):
        self.records = []

    def summary(self) -> dict:
        """label -> {launches, total_ms, avg_us, flops, bytes} (call after a device synchronise)."""
        out: dict = {}
        for label, flops, nbytes, e0, e1 in self.records:
            d = out.setdefault(label, {"launches": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["total_ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        for d in out.values():
            d["avg_us"] = 1e3 * d["total_ms"] / d["launches"]
        return out


_profiler: Optional[LaunchProfiler] = None


def set_profiler(p: Optional[LaunchProfiler]) -> None:
    global _profiler
    _profiler = p


# epilogue instances launch_panel() in csrc/gemm.hip compiles: qkv (bf16 out), FFN1 (bf16 out + GELU), out-projection
# (mask-acc + residual), to_mel (ROWS_T + mask-out + bias)
_PANEL_EPS = (64, 65, 4 | (1 << 17), 256 | 8 | (1 << 16))


def _launch(label: str, flops: float, nbytes: float, fn, *args) -> None:
    if _profiler is None:
        _check(fn(*args), label)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    _check(rc, label)
    if label == "gemm_bf16_kernel":   # resolve to the instance the library actually dispatched
        v = lib().ispk_gemm_bf16_last_variant()
        if v // 1000 == 1:   # panel kernel <KC, EP>: EP = the compile-time epilogue instance (-1: the generic one)
            ep = (args[13] & 0xffff) | ((1 << 16) if args[6] else 0) | ((1 << 17) if args[7] else 0)
            label = f"gemm_bf16_panel_kernel<{v % 1000},{ep if ep in _PANEL_EPS else -1}>"
        else:
            label = {2: f"gemm_bf16_wide_kernel<{(v % 1000) // 10},{v % 10}>",
                     3: f"gemm_bf16_kernel<{(v % 1000) // 10},{v % 10}>"}.get(v // 1000, label)
    _profiler.records.append((label, flops, nbytes, e0, e1))


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ispk_last_error_string()
        raise IspkError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def _dev(*tensors: Optional[Tensor]) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise IspkError("isp_tts_amd kernels need GPU tensors (got a CPU tensor); there is no CPU fallback")


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def device_info() -> tuple[str, int]:
    buf = ctypes.create_string_buffer(256)
    n = lib().ispk_device_info(buf, 256)
    if n <= 0:
        _check(n if n != 0 else -1, "ispk_device_info")
    return buf.value.decode(), n


def _rows2d(t: Tensor) -> Tensor:
    """[..., D] -> [rows, D] view with unit inner stride (copies only when the layout requires it)."""
    if t.stride(-1) != 1:
        t = t.contiguous()
    return t.reshape(-1, t.shape[-1])


# ------------------------------------------------------------------------------------------------- MAS
def mas(logits: Tensor, text_len: Tensor, mel_len: Tensor, want_dur: bool = True, want_path: bool = False):
    """ispk_mas_f32.  logits fp32 [B,M,L] (unit stride on L), lengths int64 [B] on the same device.
    Returns (attn_hard int16 [B,M,L], dur int64 [B,L] | None, path int16 [B,M] | None)."""
    _dev(logits, text_len, mel_len)
    assert logits.dtype == torch.float32 and logits.ndim == 3
    if logits.stride(2) != 1:
        logits = logits.contiguous()
    B, M, L = logits.shape
    text_len = text_len.to(torch.int64).contiguous()
    mel_len = mel_len.to(torch.int64).contiguous()
    hard = torch.empty((B, M, L), dtype=torch.int16, device=logits.device)
    dur = torch.empty((B, L), dtype=torch.int64, device=logits.device) if want_dur else None
    path = torch.empty((B, M), dtype=torch.int16, device=logits.device) if want_path else None
    _launch(f"mas_kernel<{(L + 63) // 64}>", 0.0, 6.0 * B * M * L, lib().ispk_mas_f32, logits.data_ptr(),
            text_len.data_ptr(), mel_len.data_ptr(), hard.data_ptr(), _ptr(dur), _ptr(path), B, M, L, logits.stride(0),
            logits.stride(1), _stream())
    return hard, dur, path


# ------------------------------------------------------------------------------------------------- LayerNorm
def layernorm(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], ada_scale: Optional[Tensor] = None,
              ada_shift: Optional[Tensor] = None, rows_per_batch: int = 1, row_mask: Optional[Tensor] = None,
              eps: float = 1e-5, out_dtype: torch.dtype = torch.float32) -> Tensor:
    """ispk_layernorm_f32[_bf16].  x fp32 [..., D]; ada_* [Bc, D] with Bc == batch or 1 (broadcast)."""
    _dev(x, gamma, beta, ada_scale, ada_shift, row_mask)
    assert x.dtype == torch.float32
    x2 = _rows2d(x)
    rows, D = x2.shape
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    ada_stride = 0
    if ada_scale is not None:   # [Bc, D] rows, possibly column slices of one wide projection (row stride kept, no copy)
        ada_scale = ada_scale.reshape(-1, D) if ada_scale.ndim != 2 else ada_scale
        if ada_scale.stride(1) != 1:
            ada_scale = ada_scale.contiguous()
        if ada_shift is not None:
            ada_shift = ada_shift.reshape(-1, D) if ada_shift.ndim != 2 else ada_shift
            if ada_shift.stride(1) != 1 or ada_shift.stride(0) != ada_scale.stride(0):
                ada_scale, ada_shift = ada_scale.contiguous(), ada_shift.contiguous()
        ada_stride = ada_scale.stride(0) if ada_scale.shape[0] > 1 else 0
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).contiguous()
        assert row_mask.dtype == torch.bool and row_mask.numel() == rows
    fn = lib().ispk_layernorm_f32 if out_dtype == torch.float32 else lib().ispk_layernorm_f32_bf16
    label = f"layernorm_vec_kernel<{D // 128}>" if D % 128 == 0 and D <= 512 else f"layernorm_kernel<{D // 64}>"
    _launch(label, 0.0, float(rows) * D * (4 + y.element_size()), fn, x2.data_ptr(),
            x2.stride(0), _ptr(gamma), _ptr(beta), _ptr(ada_scale), _ptr(ada_shift), ada_stride, rows_per_batch,
            _ptr(row_mask), y.data_ptr(), D, rows, D, eps, _stream())
    return y


# ------------------------------------------------------------------------------------------------- GEMM
def gemm(a: Tensor, w: Tensor, bias: Optional[Tensor] = None, resid: Optional[Tensor] = None,
         mask: Optional[Tensor] = None, flags: int = 0, out: Optional[Tensor] = None,
         out_dtype: Optional[torch.dtype] = None) -> Tensor:
    """C[..., N] = epilogue(a[..., K] @ w[N, K]^T)  (ispk_gemm_f32 / ispk_gemm_bf16 by a.dtype)."""
    _dev(a, w, bias, resid, mask, out)
    a2 = _rows2d(a)
    M, K = a2.shape
    N = w.shape[0]
    assert w.shape[1] == K and w.stride(1) == 1 and a.dtype == w.dtype
    bf16 = a.dtype == torch.bfloat16
    if out_dtype is None:
        out_dtype = torch.float32 if not bf16 else torch.bfloat16
    if out is None:
        out = torch.empty((*a.shape[:-1], N), dtype=out_dtype, device=a.device)
    c2 = out.view(-1, N)
    r2 = None
    if resid is not None:
        r2 = _rows2d(resid)
        assert r2.shape == (M, N)
    if mask is not None:
        mask = mask.reshape(-1).contiguous()
        assert mask.dtype == torch.bool
    if bf16:
        if out.dtype == torch.bfloat16:
            flags |= EP_OUT_BF16
        if r2 is not None and r2.dtype == torch.bfloat16:
            flags |= EP_RESID_BF16
        fn = lib().ispk_gemm_bf16
        ks = lib().ispk_gemm_bf16_splitk_plan(M, N, K, flags) if M < 2048 and K >= 512 else 1
        if ks > 1:      # few rows, long K: K slices on separate workgroups + one combine pass (ispk_gemm_bf16_splitk)
            ws = torch.empty((ks * M * N,), dtype=torch.float32, device=a.device)
            _launch(f"gemm_bf16_splitk<{ks}>", 2.0 * M * N * K, _gemm_bytes(a2, w, out, r2) + 8.0 * ks * M * N, lib().ispk_gemm_bf16_splitk,
                    a2.data_ptr(), a2.stride(0), w.data_ptr(), w.stride(0), c2.data_ptr(), c2.stride(0), _ptr(bias), _ptr(r2),
                    r2.stride(0) if r2 is not None else 0, _ptr(mask), M, N, K, flags, ws.data_ptr(), ks, _stream())
            return out
    else:
        assert out.dtype == torch.float32 and (r2 is None or r2.dtype == torch.float32)
        fn = lib().ispk_gemm_f32
    _launch(_gemm_label(bf16, M, N, K), 2.0 * M * N * K, _gemm_bytes(a2, w, out, r2), fn, a2.data_ptr(), a2.stride(0),
            w.data_ptr(), w.stride(0), c2.data_ptr(), c2.stride(0), _ptr(bias), _ptr(r2),
            r2.stride(0) if r2 is not None else 0, _ptr(mask), M, N, K, flags, 0, 0, _stream())
    return out


def ffn_pack_w2(w2: Tensor) -> Tensor:
    """ispk_ffn_pack_w2_bf16: W2 bf16 [D, inner] -> packed [inner/32, D, 32] (one-time weight staging for ffn_fused)."""
    _dev(w2)
    assert w2.dtype == torch.bfloat16 and w2.dim() == 2 and w2.stride(1) == 1
    D, Fi = w2.shape
    out = torch.empty((Fi // 32, D, 32), dtype=torch.bfloat16, device=w2.device)
    _launch("ffn_pack_w2_kernel", 0.0, 4.0 * D * Fi, lib().ispk_ffn_pack_w2_bf16, w2.data_ptr(), w2.stride(0), D, Fi,
            out.data_ptr(), _stream())
    return out


def ffn_fused(x: Tensor, w1: Tensor, w2: Tensor, resid: Optional[Tensor] = None, mask: Optional[Tensor] = None,
              bias1: Optional[Tensor] = None, bias2: Optional[Tensor] = None, flags: int = 0) -> Tensor:
    """ispk_ffn_bf16: out fp32 [..., D] = [mask] * (resid + gelu(x @ w1^T + bias1) @ w2^T + bias2), x / w1 / w2 bf16.
    w2 is either [D, inner] (nn.Linear layout) or the 3-D packed image from `ffn_pack_w2` (faster)."""
    _dev(x, w1, w2, resid, mask, bias1, bias2)
    assert x.dtype == torch.bfloat16 and w1.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16
    x2 = _rows2d(x)
    R, D = x2.shape
    Fi = w1.shape[0]
    assert w1.shape == (Fi, D) and w1.stride(1) == 1
    packed = w2.dim() == 3
    if packed:
        assert w2.shape == (Fi // 32, D, 32) and w2.is_contiguous()
    else:
        assert w2.shape == (D, Fi) and w2.stride(1) == 1
    out = torch.empty((*x.shape[:-1], D), dtype=torch.float32, device=x.device)
    r2 = _rows2d(resid) if resid is not None else None
    if mask is not None:
        mask = mask.reshape(-1).contiguous()
    nb = x2.numel() * 2 + (w1.numel() + w2.numel()) * 2 + out.numel() * 4 + (r2.numel() * 4 if r2 is not None else 0)
    _launch(f"ffn_bf16_kernel<{D // 64}>", 4.0 * R * D * Fi, float(nb), lib().ispk_ffn_bf16, x2.data_ptr(), x2.stride(0),
            w1.data_ptr(), w1.stride(0), _ptr(bias1), w2.data_ptr(), 0 if packed else w2.stride(0), _ptr(bias2), _ptr(r2),
            r2.stride(0) if r2 is not None else 0, _ptr(mask), out.data_ptr(), D, R, D, Fi, flags, _stream())
    return out


def ffn_prenorm(x: Tensor, norm_weight: Tensor, norm_bias: Tensor, w1: Tensor, w2p: Tensor, mask: Optional[Tensor] = None,
                bias2: Optional[Tensor] = None, flags: int = 0, norm_eps: float = 1e-5, want_stats: bool = False,
                stats_eps: float = 1e-5):
    """ispk_ffn_bf16_prenorm: out fp32 [..., D] = [mask] * (x + gelu(LN(x) @ w1^T) @ w2^T + bias2) from the fp32 rows x
    (LayerNorm input AND residual); with `want_stats` also the (mean, rstd) of the output rows, fp32 [rows, 2]."""
    _dev(x, norm_weight, norm_bias, w1, w2p, mask, bias2)
    assert x.dtype == torch.float32 and w1.dtype == torch.bfloat16 and w2p.dtype == torch.bfloat16
    x2 = _rows2d(x)
    R, D = x2.shape
    Fi = w1.shape[0]
    assert w1.shape == (Fi, D) and w1.stride(1) == 1 and w2p.shape == (Fi // 32, D, 32) and w2p.is_contiguous()
    out = torch.empty((*x.shape[:-1], D), dtype=torch.float32, device=x.device)
    stats = torch.empty((R, 2), dtype=torch.float32, device=x.device) if want_stats else None
    if mask is not None:
        mask = mask.reshape(-1).contiguous()
    nb = x2.numel() * 8 + (w1.numel() + w2p.numel()) * 2 + out.numel() * 4 + (R * 8 if want_stats else 0)
    _launch(f"ffn_bf16_kernel<{D // 64}>", 4.0 * R * D * Fi, float(nb), lib().ispk_ffn_bf16_prenorm, x2.data_ptr(),
            x2.stride(0), norm_weight.data_ptr(), norm_bias.data_ptr(), norm_eps, w1.data_ptr(), w1.stride(0),
            w2p.data_ptr(), _ptr(bias2), _ptr(mask), out.data_ptr(), D, R, D, Fi, flags, _ptr(stats), stats_eps, _stream())
    return (out, stats) if want_stats else out


def ffn_chunk_w2(w2: Tensor) -> Tensor:
    """ispk_ffn_chunk_w2_bf16: W2 bf16 [D, inner] -> chunk-contiguous [inner/32, D, 32] (weight staging for ffn_prenorm2)."""
    _dev(w2)
    assert w2.dtype == torch.bfloat16 and w2.dim() == 2 and w2.stride(1) == 1
    D, Fi = w2.shape
    out = torch.empty((Fi // 32, D, 32), dtype=torch.bfloat16, device=w2.device)
    _launch("ffn_chunk_w2_kernel", 0.0, 4.0 * D * Fi, lib().ispk_ffn_chunk_w2_bf16, w2.data_ptr(), w2.stride(0), D, Fi,
            out.data_ptr(), _stream())
    return out


def ffn_prenorm2(x: Tensor, norm_weight: Tensor, norm_bias: Tensor, w1: Tensor, w2c: Tensor, mask: Optional[Tensor] = None,
                 flags: int = 0, norm_eps: float = 1e-5, want_stats: bool = False, stats_eps: float = 1e-5):
    """ispk_ffn_bf16_prenorm2 (dim 384, eight-wave kernel): out fp32 [..., D] = [mask] * (x + gelu(LN(x) @ w1^T) @ w2^T) from the
    fp32 rows x; with `want_stats` also the (mean, rstd) of the output rows, fp32 [rows, 2].  w2c = `ffn_chunk_w2(w2)`."""
    _dev(x, norm_weight, norm_bias, w1, w2c, mask)
    assert x.dtype == torch.float32 and w1.dtype == torch.bfloat16 and w2c.dtype == torch.bfloat16
    x2 = _rows2d(x)
    R, D = x2.shape
    Fi = w1.shape[0]
    assert w1.shape == (Fi, D) and w1.is_contiguous() and w2c.shape == (Fi // 32, D, 32) and w2c.is_contiguous()
    out = torch.empty((*x.shape[:-1], D), dtype=torch.float32, device=x.device)
    stats = torch.empty((R, 2), dtype=torch.float32, device=x.device) if want_stats else None
    if mask is not None:
        mask = mask.reshape(-1).contiguous()
    nb = x2.numel() * 8 + (w1.numel() + w2c.numel()) * 2 + out.numel() * 4 + (R * 8 if want_stats else 0)
    _launch("ffn2_bf16_kernel<0>", 4.0 * R * D * Fi, float(nb), lib().ispk_ffn_bf16_prenorm2, x2.data_ptr(),
            x2.stride(0), norm_weight.data_ptr(), norm_bias.data_ptr(), norm_eps, w1.data_ptr(), w2c.data_ptr(), _ptr(mask),
            out.data_ptr(), D, R, D, Fi, flags, _ptr(stats), stats_eps, _stream())
    return (out, stats) if want_stats else out


def chunk_k16(w: Tensor) -> Tensor:
    """ispk_chunk_k16_bf16: W bf16 [N, K] -> k-step chunks [K/16, N, 16] (weight staging for attn_out_ffn's q/kv epilogue)."""
    _dev(w)
    assert w.dtype == torch.bfloat16 and w.dim() == 2 and w.stride(1) == 1
    N, K = w.shape
    out = torch.empty((K // 16, N, 16), dtype=torch.bfloat16, device=w.device)
    _launch("chunk_k16_kernel", 0.0, 4.0 * N * K, lib().ispk_chunk_k16_bf16, w.data_ptr(), w.stride(0), N, K, out.data_ptr(),
            _stream())
    return out


def attn_out_ffn(x: Tensor, attn_out: Tensor, woc: Tensor, norm_weight: Tensor, norm_bias: Tensor, w1: Tensor, w2c: Tensor,
                 mask: Optional[Tensor] = None, norm_eps: float = 1e-5, want_stats: bool = False, stats_eps: float = 1e-5,
                 next_qkv: Optional[tuple] = None, final_norm: Optional[tuple] = None, want_out: bool = True):
    """ispk_attn_out_ffn_bf16 (dim 384 = heads * 64): the second half of a pre-norm layer in one kernel,
        x1 = x + [mask] * (attn_out @ Wo^T);  out = [mask] * (x1 + gelu(LN(x1) @ w1^T) @ w2^T)
    from the fp32 residual rows x and the bf16 attention output; woc = `ffn_chunk_w2(Wo)`, w2c = `ffn_chunk_w2(w2)`.  With
    `want_stats` also the (mean, rstd) of the output rows, fp32 [rows, 2].  With `next_qkv` = (norm weight, norm bias, eps,
    `chunk_k16([Wq; Wkv])`) of the NEXT layer (ispk_attn_out_ffn_qkv_bf16) also that layer's q/kv rows, bf16 [..., 512]:
    -> (out, qkv).  With `final_norm` = (weight, bias, eps, apply_mask, dtype) of the STACK's final LayerNorm
    (ispk_attn_out_ffn_norm_bf16) also LN_final(out) [* mask]: -> (out | None, ln); `want_out=False` does not store the raw rows."""
    _dev(x, attn_out, woc, norm_weight, norm_bias, w1, w2c, mask)
    assert x.dtype == torch.float32 and attn_out.dtype == torch.bfloat16 and w1.dtype == torch.bfloat16
    assert woc.dtype == torch.bfloat16 and w2c.dtype == torch.bfloat16
    x2, o2 = _rows2d(x), _rows2d(attn_out)
    R, D = x2.shape
    Fi = w1.shape[0]
    assert o2.shape == (R, D) and woc.shape == (D // 32, D, 32) and woc.is_contiguous()
    assert w1.shape == (Fi, D) and w1.is_contiguous() and w2c.shape == (Fi // 32, D, 32) and w2c.is_contiguous()
    out = torch.empty((*x.shape[:-1], D), dtype=torch.float32, device=x.device)
    stats = torch.empty((R, 2), dtype=torch.float32, device=x.device) if want_stats else None
    flags = 0
    if mask is not None:
        mask = mask.reshape(-1).contiguous()
        flags = EP_MASK_ACC | EP_MASK_OUT
    nb = x2.numel() * 4 + o2.numel() * 2 + (woc.numel() + w1.numel() + w2c.numel()) * 2 + out.numel() * 4 + (R * 8 if want_stats else 0)
    if final_norm is not None:
        assert not want_stats and next_qkv is None
        fw, fb, feps, fmask, fdtype = final_norm
        _dev(fw, fb)
        assert fdtype in (torch.float32, torch.bfloat16)
        ln = torch.empty(x.shape, dtype=fdtype, device=x.device)
        outp = out if want_out else None
        _launch("ffn2_bf16_kernel<50>", 4.0 * R * D * Fi + 2.0 * R * D * D,
                float(nb - (0 if want_out else out.numel() * 4) + ln.numel() * ln.element_size()), lib().ispk_attn_out_ffn_norm_bf16,
                x2.data_ptr(), x2.stride(0), o2.data_ptr(), o2.stride(0), woc.data_ptr(), norm_weight.data_ptr(), norm_bias.data_ptr(),
                norm_eps, w1.data_ptr(), w2c.data_ptr(), _ptr(mask), _ptr(outp), D, R, D, Fi, flags, fw.data_ptr(), fb.data_ptr(), feps,
                int(bool(fmask) and mask is not None), ln.data_ptr(), D, int(fdtype == torch.bfloat16), _stream())
        return outp, ln
    if next_qkv is not None:
        assert not want_stats
        ng, nbeta, neps, wqc = next_qkv
        _dev(ng, nbeta, wqc)
        assert wqc.dtype == torch.bfloat16 and wqc.shape == (D // 16, 512, 16) and wqc.is_contiguous()
        qkv = torch.empty((*x.shape[:-1], 512), dtype=torch.bfloat16, device=x.device)
        _launch("ffn2_bf16_kernel<51>", 4.0 * R * D * Fi + 2.0 * R * D * D + 2.0 * R * D * 512, float(nb + wqc.numel() * 2 + R * 1024),
                lib().ispk_attn_out_ffn_qkv_bf16, x2.data_ptr(), x2.stride(0), o2.data_ptr(), o2.stride(0), woc.data_ptr(),
                norm_weight.data_ptr(), norm_bias.data_ptr(), norm_eps, w1.data_ptr(), w2c.data_ptr(), _ptr(mask), out.data_ptr(), D,
                R, D, Fi, flags, ng.data_ptr(), nbeta.data_ptr(), neps, wqc.data_ptr(), qkv.data_ptr(), 512, _stream())
        return out, qkv
    _launch("ffn2_bf16_kernel<50>", 4.0 * R * D * Fi + 2.0 * R * D * D, float(nb), lib().ispk_attn_out_ffn_bf16, x2.data_ptr(),
            x2.stride(0), o2.data_ptr(), o2.stride(0), woc.data_ptr(), norm_weight.data_ptr(), norm_bias.data_ptr(), norm_eps,
            w1.data_ptr(), w2c.data_ptr(), _ptr(mask), out.data_ptr(), D, R, D, Fi, flags, _ptr(stats), stats_eps, _stream())
    return (out, stats) if want_stats else out


def ffn_prenorm2_split(x: Tensor, norm_weight: Tensor, norm_bias: Tensor, w1: Tensor, w2c: Tensor, mask: Optional[Tensor],
                       splits: int, next_norm: Optional[tuple] = None, norm_eps: float = 1e-5, attn_proj: Optional[tuple] = None):
    """Small-batch form of `ffn_prenorm2` (ispk_ffn_bf16_prenorm2_split + ispk_ffn_combine_ln_f32): the inner dimension split
    over `splits` workgroups per row block, partial products added in split order with the residual and the mask, and -
    `next_norm` = (weight, bias, eps, apply_mask, dtype) - the LayerNorm that consumes the result from the same pass.
    -> (y fp32, LN(y) | None).
    `attn_proj` = (attention output bf16 [..., D], `ffn_chunk_w2(Wo)`): x is the layer's INPUT and every split first forms
    x1 = x + [mask] * (attn_out @ Wo^T) in its accumulators (ispk_attn_out_ffn_split_bf16; split 0's partial product carries x1,
    the combine pass runs without a residual): y = [mask] * (x1 + feed_forward(LN(x1)))."""
    _dev(x, norm_weight, norm_bias, w1, w2c, mask)
    assert x.dtype == torch.float32 and w1.dtype == torch.bfloat16 and w2c.dtype == torch.bfloat16
    x2 = _rows2d(x)
    R, D = x2.shape
    Fi = w1.shape[0]
    parts = torch.empty((splits, R, D), dtype=torch.float32, device=x.device)   # per call: graph instances may run side by side
    if attn_proj is not None:
        o, woc = attn_proj
        _dev(o, woc)
        o2 = _rows2d(o)
        assert o.dtype == torch.bfloat16 and o2.shape == (R, D) and woc.dtype == torch.bfloat16 and woc.shape == (D // 32, D, 32)
        mflat = mask.reshape(-1).contiguous() if mask is not None else None
        _launch("ffn2_bf16_kernel<21>", (4.0 * R * D * Fi) + 2.0 * R * D * D * splits,
                float((x2.numel() * 4 + o2.numel() * 2 + woc.numel() * 2) * splits + (w1.numel() + w2c.numel()) * 2 + splits * R * D * 4),
                lib().ispk_attn_out_ffn_split_bf16, x2.data_ptr(), x2.stride(0), o2.data_ptr(), o2.stride(0), woc.data_ptr(),
                norm_weight.data_ptr(), norm_bias.data_ptr(), norm_eps, w1.data_ptr(), w2c.data_ptr(), _ptr(mflat),
                EP_MASK_ACC if mflat is not None else 0, parts.data_ptr(), R * D, splits, R, D, Fi, _stream())
    else:
        nbytes = x2.numel() * 4 * splits + (w1.numel() + w2c.numel()) * 2 + splits * R * D * 4
        _launch("ffn2_bf16_kernel<20>", 4.0 * R * D * Fi, float(nbytes), lib().ispk_ffn_bf16_prenorm2_split, x2.data_ptr(),
                x2.stride(0), norm_weight.data_ptr(), norm_bias.data_ptr(), norm_eps, w1.data_ptr(), w2c.data_ptr(),
                parts.data_ptr(), R * D, splits, R, D, Fi, _stream())
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    if mask is not None:
        mask = mask.reshape(-1).contiguous()
    ln = None
    nw = nb = None
    neps, nmask, nbf16 = 1e-5, 0, 0
    if next_norm is not None:
        nw, nb, neps, apply_mask, ndtype = next_norm
        ln = torch.empty(x.shape, dtype=ndtype, device=x.device)
        nmask, nbf16 = int(bool(apply_mask) and mask is not None), int(ndtype == torch.bfloat16)
    _launch("ffn_combine_ln_kernel", 0.0, float(R * D * 4 * (2 + splits) + (R * D * ln.element_size() if ln is not None else 0)),
            lib().ispk_ffn_combine_ln_f32, None if attn_proj is not None else x2.data_ptr(), x2.stride(0), parts.data_ptr(), R * D,
            splits, _ptr(mask), y.data_ptr(),
            D, _ptr(nw), _ptr(nb), neps, nmask, _ptr(ln), D, nbf16, R, D, _stream())
    return y, ln


def gemm_lnin(x: Tensor, stats: Optional[Tensor], ln_weight: Tensor, ln_bias: Tensor, w: Tensor,
              bias: Optional[Tensor] = None, mask: Optional[Tensor] = None, flags: int = 0,
              out_dtype: torch.dtype = torch.bfloat16, ln_eps: float = 1e-5) -> Tensor:
    """ispk_gemm_bf16_lnin: C[..., N] = epilogue(bf16(LayerNorm(x)) @ w[N, K]^T) with x fp32 [..., K]; the rows'
    (mean, rstd) come from `stats` (written by `ffn_prenorm` / `ffn_prenorm2`) or, with stats None, are computed by
    the kernel itself."""
    _dev(x, stats, ln_weight, ln_bias, w, bias, mask)
    assert x.dtype == torch.float32 and w.dtype == torch.bfloat16
    x2 = _rows2d(x)
    M, K = x2.shape
    N = w.shape[0]
    assert w.shape == (N, K) and w.stride(1) == 1
    assert stats is None or (stats.dtype == torch.float32 and stats.shape == (M, 2) and stats.is_contiguous())
    if out_dtype == torch.bfloat16:
        flags |= EP_OUT_BF16
    out = torch.empty((*x.shape[:-1], N), dtype=out_dtype, device=x.device)
    if mask is not None:
        mask = mask.reshape(-1).contiguous()
    nb = x2.numel() * 4 + (stats.numel() * 4 if stats is not None else 0) + w.numel() * 2 + out.numel() * out.element_size()
    _launch(f"gemm_bf16_panel_kernel<{K // 64},lnin>", 2.0 * M * N * K, float(nb), lib().ispk_gemm_bf16_lnin, x2.data_ptr(),
            x2.stride(0), _ptr(stats), ln_weight.data_ptr(), ln_bias.data_ptr(), ln_eps, w.data_ptr(), w.stride(0),
            out.data_ptr(), N, _ptr(bias), 0, 0, _ptr(mask), M, N, K, flags, _stream())
    return out


def _gemm_label(bf16: bool, M: int, N: int, K: int) -> str:
    if bf16:
        return "gemm_bf16_kernel"
    t = lib().ispk_gemm_f32_tile(M, N, K)
    return f"gemm_f32_kernel<{t // 10},{t % 10}>"


def _gemm_bytes(a2: Tensor, w: Tensor, out: Tensor, r2: Optional[Tensor]) -> float:
    n = a2.numel() * a2.element_size() + w.numel() * w.element_size() + out.numel() * out.element_size()
    return float(n + (r2.numel() * r2.element_size() if r2 is not None else 0))


def to_mel(dec: Tensor, weight: Tensor, bias: Tensor, mask: Optional[Tensor]) -> Tensor:
    """mel[B, C, T] = mask[b,t] * (dec[B,T,D] @ weight[C,D]^T + bias[C])  — Linear + transpose + mask of
    model.py:167-168 as ONE GEMM with swapped operands: lanes run along the mel-frame axis T, so the transposed
    output is written with coalesced 128-B segments."""
    _dev(dec, weight, bias, mask)
    B, T, D = dec.shape
    C = weight.shape[0]
    x2 = _rows2d(dec)
    out = torch.empty((B, C, T), dtype=torch.float32, device=dec.device)
    if dec.dtype == torch.bfloat16 and D in (256, 384) and C % 4 == 0 and bias is not None:
        # bf16, K = 256 / 384: the panel GEMM with frames as rows and the transposed per-batch store (ISPK_EP_ROWS_T)
        flags = EP_ROWS_T
        if mask is not None:
            mask = mask.reshape(-1).contiguous()
            flags |= EP_MASK_OUT
        _launch("gemm_bf16_kernel", 2.0 * C * B * T * D, _gemm_bytes(x2, weight, out, None), lib().ispk_gemm_bf16,
                x2.data_ptr(), x2.stride(0), weight.data_ptr(), weight.stride(0), out.data_ptr(), T, _ptr(bias), None, 0,
                _ptr(mask), B * T, C, D, flags, T, C * T, _stream())
        return out
    flags = EP_BIAS_ROW | EP_MASK_COL
    if mask is not None:
        mask = mask.reshape(-1).contiguous()
        flags |= EP_MASK_OUT
    fn = lib().ispk_gemm_bf16 if dec.dtype == torch.bfloat16 else lib().ispk_gemm_f32
    _launch(_gemm_label(dec.dtype == torch.bfloat16, C, B * T, D), 2.0 * C * B * T * D, _gemm_bytes(x2, weight, out, None),
            fn, weight.data_ptr(), weight.stride(0), x2.data_ptr(), x2.stride(0), out.data_ptr(), T, _ptr(bias), None, 0,
            _ptr(mask), C, B * T, D, flags, T, C * T, _stream())
    return out


def linear_small(a: Tensor, w: Tensor, bias: Optional[Tensor] = None, resid: Optional[Tensor] = None,
                 act: int = 0) -> Tensor:
    """ispk_linear_small_f32: any K / N, fp32.  `w` may be a column slice of a wider weight (stride kept)."""
    _dev(a, w, bias, resid)
    assert a.dtype == torch.float32 and w.dtype == torch.float32 and w.stride(1) == 1
    a2 = _rows2d(a)
    M, K = a2.shape
    N = w.shape[0]
    assert w.shape[1] == K
    out = torch.empty((*a.shape[:-1], N), dtype=torch.float32, device=a.device)
    r2 = _rows2d(resid) if resid is not None else None
    _launch("linear_small_kernel", 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N), lib().ispk_linear_small_f32,
            a2.data_ptr(), a2.stride(0), w.data_ptr(), w.stride(0), _ptr(bias), _ptr(r2),
            r2.stride(0) if r2 is not None else 0, out.data_ptr(), N, M, N, K, act, _stream())
    return out


def linear(a: Tensor, w: Tensor, bias: Optional[Tensor] = None, act: int = 0) -> Tensor:
    """nn.Linear on the device: MFMA GEMM when the shape allows (K % 8 == 0, enough rows), else the small kernel."""
    K = w.shape[1]
    rows = a.numel() // K
    if a.dtype == torch.float32 and (K % 8 != 0 or rows * w.shape[0] < 64 * 64 or w.stride(0) % 4 != 0
                                     or w.data_ptr() % 16 != 0):
        return linear_small(a, w, bias, None, act)
    return gemm(a, w, bias=bias, flags=act)


# ------------------------------------------------------------------------------------------------- attention
def alibi_mqa_attention_raw(q: Tensor, ldq: int, k: Tensor, v: Tensor, ldkv: int, slopes: Tensor,
                            key_len: Optional[Tensor], B: int, N: int, heads: int, q_tiles: int = 0) -> Tensor:
    """ispk_alibi_mqa_attn_*: q is any tensor whose storage holds [B][N][H*64] rows at leading stride ldq starting at
    q.data_ptr(); k / v likewise [B][N][64] at stride ldkv.  Returns the merged heads [B, N, H*64]."""
    _dev(q, k, v, slopes, key_len)
    out = torch.empty((B, N, heads * 64), dtype=q.dtype, device=q.device)
    if key_len is not None:
        key_len = key_len.to(torch.int64).contiguous()
    slopes = slopes.to(torch.float32).contiguous()
    es = q.element_size()
    label = ("attn_f32_kernel" if es == 4 else "attn_bf16_kernel") + ("<768>" if heads <= 6 else "<1024>")
    flops, nbytes = 256.0 * B * N * N * heads, float(B) * N * (2 * heads * 64 + 128) * es
    if q_tiles:   # explicit query tiles per workgroup (bf16 kernel only; 0 = the launcher's own choice)
        assert es == 2
        _launch(label, flops, nbytes, lib().ispk_alibi_mqa_attn_bf16_tiles, q.data_ptr(), ldq, k.data_ptr(), v.data_ptr(),
                ldkv, slopes.data_ptr(), _ptr(key_len), out.data_ptr(), heads * 64, B, N, heads, q_tiles, _stream())
        return out
    fn = lib().ispk_alibi_mqa_attn_f32 if es == 4 else lib().ispk_alibi_mqa_attn_bf16
    _launch(label, flops, nbytes, fn, q.data_ptr(), ldq, k.data_ptr(), v.data_ptr(), ldkv,
            slopes.data_ptr(), _ptr(key_len), out.data_ptr(), heads * 64, B, N, heads, _stream())
    return out


def alibi_mqa_attention(qkv: Tensor, heads: int, slopes: Tensor, key_len: Optional[Tensor], q_tiles: int = 0) -> Tensor:
    """qkv [B, N, H*64 + 128] = [Q | K | V] (the fused to_q / to_kv projection) -> merged heads [B, N, H*64]."""
    B, N, W = qkv.shape
    assert W == heads * 64 + 128 and qkv.is_contiguous()
    return alibi_mqa_attention_raw(qkv, W, qkv[..., heads * 64:], qkv[..., heads * 64 + 64:], W, slopes, key_len, B, N,
                                   heads, q_tiles)


# ------------------------------------------------------------------------------------------------- split-fp16 (parity-grade fast path)
# A "split" tensor is a torch.float16 tensor [2, *shape]: plane 0 = hi = fp16(v), plane 1 = lo = fp16(v - hi).
def split_f16(x: Tensor) -> Tensor:
    """ispk_split_f16: fp32 [..., C] (unit inner stride) -> split planes fp16 [2, ..., C]."""
    _dev(x)
    assert x.dtype == torch.float32
    x2 = _rows2d(x)
    rows, cols = x2.shape
    out = torch.empty((2, *x.shape), dtype=torch.float16, device=x.device)
    _launch("split_f16_kernel", 0.0, 8.0 * rows * cols, lib().ispk_split_f16, x2.data_ptr(), x2.stride(0), out[0].data_ptr(),
            out[1].data_ptr(), cols, rows, cols, _stream())
    return out


def layernorm_split(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], ada_scale: Optional[Tensor] = None,
                    ada_shift: Optional[Tensor] = None, rows_per_batch: int = 1, row_mask: Optional[Tensor] = None,
                    eps: float = 1e-5) -> Tensor:
    """ispk_layernorm_f32_split: `layernorm` with the result as split planes fp16 [2, ..., D]."""
    _dev(x, gamma, beta, ada_scale, ada_shift, row_mask)
    assert x.dtype == torch.float32
    x2 = _rows2d(x)
    rows, D = x2.shape
    y = torch.empty((2, *x.shape), dtype=torch.float16, device=x.device)
    ada_stride = 0
    if ada_scale is not None:
        ada_scale = ada_scale.reshape(-1, D) if ada_scale.ndim != 2 else ada_scale
        if ada_scale.stride(1) != 1:
            ada_scale = ada_scale.contiguous()
        if ada_shift is not None:
            ada_shift = ada_shift.reshape(-1, D) if ada_shift.ndim != 2 else ada_shift
            if ada_shift.stride(1) != 1 or ada_shift.stride(0) != ada_scale.stride(0):
                ada_scale, ada_shift = ada_scale.contiguous(), ada_shift.contiguous()
        ada_stride = ada_scale.stride(0) if ada_scale.shape[0] > 1 else 0
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).contiguous()
        assert row_mask.dtype == torch.bool and row_mask.numel() == rows
    label = f"layernorm_vec_kernel<{D // 128},split>" if D % 128 == 0 and D <= 512 else f"layernorm_kernel<{D // 64},split>"
    _launch(label, 0.0, float(rows) * D * 8, lib().ispk_layernorm_f32_split, x2.data_ptr(), x2.stride(0), _ptr(gamma),
            _ptr(beta), _ptr(ada_scale), _ptr(ada_shift), ada_stride, rows_per_batch, _ptr(row_mask), y.data_ptr(), D,
            y.stride(0), rows, D, eps, _stream())
    return y


def _split_label(M: int, N: int, K: int) -> str:
    t = lib().ispk_gemm_split_f16_tile(M, N, K)
    return f"gemm_split_f16_kernel<{t // 100},{t // 10 % 10},{t % 10}>"


def gemm_split(a: Tensor, w: Tensor, bias: Optional[Tensor] = None, resid: Optional[Tensor] = None,
               mask: Optional[Tensor] = None, flags: int = 0, out_split: bool = False) -> Tensor:
    """ispk_gemm_split_f16: epilogue(a @ w^T) with a = split planes [2, ..., K], w = split planes [2, N, K].
    Returns fp32 [..., N], or split planes [2, ..., N] with `out_split` (no residual)."""
    _dev(a, w, bias, resid, mask)
    assert a.dtype == torch.float16 and w.dtype == torch.float16 and a.shape[0] == 2 and w.ndim == 3 and w.shape[0] == 2
    assert a.is_contiguous() and w.is_contiguous()
    K, N = a.shape[-1], w.shape[1]
    assert w.shape[2] == K
    M = a[0].numel() // K
    lead = a.shape[1:-1]
    r2 = None
    if resid is not None:
        r2 = _rows2d(resid)
        assert r2.shape == (M, N) and r2.dtype == torch.float32 and not out_split
    if mask is not None:
        mask = mask.reshape(-1).contiguous()
        assert mask.dtype == torch.bool
    if out_split:
        out = torch.empty((2, *lead, N), dtype=torch.float16, device=a.device)
        flags |= EP_OUT_SPLIT
        c_plane, nb_out = out.stride(0), 4.0 * M * N
    else:
        out = torch.empty((*lead, N), dtype=torch.float32, device=a.device)
        c_plane, nb_out = 0, 4.0 * M * N
    nb = 4.0 * M * K + 4.0 * N * K + nb_out + (4.0 * M * N if r2 is not None else 0.0)
    _launch(_split_label(M, N, K), 2.0 * M * N * K, nb, lib().ispk_gemm_split_f16, a.data_ptr(), K, a.stride(0), w.data_ptr(), K,
            w.stride(0), out.data_ptr(), N, c_plane, _ptr(bias), _ptr(r2), r2.stride(0) if r2 is not None else 0, _ptr(mask), M, N,
            K, flags, 0, 0, _stream())
    return out


def to_mel_split(dec: Tensor, w: Tensor, bias: Tensor, mask: Optional[Tensor]) -> Tensor:
    """mel[B, C, T] = mask * (dec @ w^T + bias) from split planes dec [2, B, T, D], w [2, C, D] (ISPK_EP_ROWS_T)."""
    _dev(dec, w, bias, mask)
    _, B, T, D = dec.shape
    C = w.shape[1]
    out = torch.empty((B, C, T), dtype=torch.float32, device=dec.device)
    flags = EP_ROWS_T
    if mask is not None:
        mask = mask.reshape(-1).contiguous()
        flags |= EP_MASK_OUT
    _launch(_split_label(B * T, C, D), 2.0 * C * B * T * D, 4.0 * (B * T * D + C * D + B * C * T), lib().ispk_gemm_split_f16,
            dec.data_ptr(), D, dec.stride(0), w.data_ptr(), D, w.stride(0), out.data_ptr(), T, 0, _ptr(bias), None, 0, _ptr(mask),
            B * T, C, D, flags, T, C * T, _stream())
    return out


def conv5_padded_split(xpad: Tensor, w2d: Tensor, flags: int = 0) -> Tensor:
    """`conv5_padded` on split planes: xpad [2, B, T+4, C], w2d [2, O, k*C] -> fp32 [B, T+4, O] (row t = frame t)."""
    _dev(xpad, w2d)
    _, B, TP, C = xpad.shape
    _, O, K = w2d.shape
    taps = K // C
    assert taps * C == K and taps in (1, 5) and xpad.is_contiguous() and w2d.is_contiguous()
    out = torch.empty((B, TP, O), dtype=torch.float32, device=xpad.device)
    a_ptr = xpad.data_ptr() + (0 if taps == 5 else 2 * C * 2)   # k=1: frame t sits at padded row t+2
    M = B * TP - 4
    _launch(_split_label(M, O, K), 2.0 * M * O * K, 4.0 * (M * C + O * K + M * O), lib().ispk_gemm_split_f16, a_ptr, C,
            xpad.stride(0), w2d.data_ptr(), K, w2d.stride(0), out.data_ptr(), O, 0, None, None, 0, None, M, O, K, flags, 0, 0,
            _stream())
    return out


def alibi_mqa_attention_split(qkv: Tensor, heads: int, slopes: Tensor, key_len: Optional[Tensor], out_split: bool = True) -> Tensor:
    """ispk_alibi_mqa_attn_split_f16: qkv fp32 [B, N, H*64 + 128] -> merged heads as split planes [2, B, N, H*64] (or fp32)."""
    _dev(qkv, slopes, key_len)
    B, N, W = qkv.shape
    assert W == heads * 64 + 128 and qkv.is_contiguous() and qkv.dtype == torch.float32
    if key_len is not None:
        key_len = key_len.to(torch.int64).contiguous()
    slopes = slopes.to(torch.float32).contiguous()
    if out_split:
        out = torch.empty((2, B, N, heads * 64), dtype=torch.float16, device=qkv.device)
        plane = out.stride(0)
    else:
        out = torch.empty((B, N, heads * 64), dtype=torch.float32, device=qkv.device)
        plane = 0
    es = qkv.element_size()
    _launch("attn_split_f16_kernel", 256.0 * B * N * N * heads, float(B) * N * (2 * heads * 64 + 128) * 4,
            lib().ispk_alibi_mqa_attn_split_f16, qkv.data_ptr(), W, qkv.data_ptr() + heads * 64 * es,
            qkv.data_ptr() + (heads * 64 + 64) * es, W, slopes.data_ptr(), _ptr(key_len), out.data_ptr(), heads * 64, plane, B, N,
            heads, _stream())
    return out


# ------------------------------------------------------------------------------------------------- aligner front-end
def pad_rows(x: Tensor, lengths: Tensor, channel_first: bool = False, out_dtype: torch.dtype = torch.float32) -> Tensor:
    """ispk_pad_rows_f32: [B,T,C] (or [B,C,T] with channel_first) -> masked, zero-padded channel-last [B,T+4,C]."""
    _dev(x, lengths)
    assert x.dtype == torch.float32 and x.ndim == 3
    if channel_first:
        B, C, T = x.shape
        sb, sc, st = x.stride()
    else:
        B, T, C = x.shape
        sb, st, sc = x.stride()
    lengths = lengths.to(torch.int64).contiguous()
    split = out_dtype == torch.float16      # split fp16 planes [2, B, T+4, C] (hi, lo): the split-fp16 GEMMs' operand format
    out = torch.empty((2, B, T + 4, C) if split else (B, T + 4, C), dtype=out_dtype, device=x.device)
    _launch("pad_rows_kernel", 0.0, 8.0 * B * T * C, lib().ispk_pad_rows_f32, x.data_ptr(), sb, st, sc,
            lengths.data_ptr(), out.data_ptr(), 2 if split else int(out_dtype == torch.bfloat16), B, T, C, _stream())
    return out


def conv5_padded(xpad: Tensor, w2d: Tensor, flags: int = 0) -> Tensor:
    """Conv1d(kernel k, padding (k-1)/2, no bias) over a padded channel-last buffer as ONE GEMM over overlapping rows.
    xpad [B, T+4, C]; w2d [O, k*C] (= conv.weight.permute(0,2,1).reshape(O, k*C)), k = 5 or 1.  Returns [B, T+4, O] whose
    row t (not t+2) of every utterance is frame t; the last 4 rows per utterance are scratch."""
    _dev(xpad, w2d)
    B, TP, C = xpad.shape
    O, K = w2d.shape
    taps = K // C
    assert taps * C == K and taps in (1, 5) and xpad.is_contiguous() and w2d.is_contiguous()
    assert xpad.dtype == w2d.dtype
    bf16 = xpad.dtype == torch.bfloat16
    out = torch.empty((B, TP, O), dtype=torch.float32, device=xpad.device)      # fp32 out on both paths
    a_ptr = xpad.data_ptr() + (0 if taps == 5 else 2 * C * xpad.element_size())  # k=1: frame t sits at padded row t+2
    M = B * TP - 4
    fn = lib().ispk_gemm_bf16 if bf16 else lib().ispk_gemm_f32
    _launch(_gemm_label(bf16, M, O, K), 2.0 * M * O * K, float(xpad.element_size()) * (M * C + O * K) + 4.0 * M * O, fn,
            a_ptr, C, w2d.data_ptr(), K, out.data_ptr(), O, None, None, 0, None, M, O, K, flags, 0, 0, _stream())
    return out


def masked_instnorm(y: Tensor, weight: Tensor, bias: Tensor, lengths: Tensor, eps: float = 1e-5,
                    out_dtype: torch.dtype = torch.float32) -> Tensor:
    """ispk_masked_instnorm_f32: conv output [B,T+4,C] (row t = frame t) -> normalised, masked, re-padded [B,T+4,C]."""
    _dev(y, weight, bias, lengths)
    B, TP, C = y.shape
    split = out_dtype == torch.float16      # split fp16 planes [2, B, T+4, C]
    out = torch.empty((2, *y.shape) if split else y.shape, dtype=out_dtype, device=y.device)
    lengths = lengths.to(torch.int64).contiguous()
    _launch("masked_instnorm_kernel", 0.0, 16.0 * B * TP * C, lib().ispk_masked_instnorm_f32, y.data_ptr(),
            weight.data_ptr(), bias.data_ptr(), lengths.data_ptr(), out.data_ptr(), 2 if split else int(out_dtype == torch.bfloat16), B,
            TP - 4, C, eps, _stream())
    return out


def aligner_scores(q_enc: Tensor, k_enc: Tensor, text_len: Tensor, mel_len: Tensor, M: int, L: int, fast: bool = False):
    """ispk_aligner_scores_f32: q_enc [B, M+4, 128], k_enc [B, L+4, 128] (row t = frame/token t) ->
    (attn_soft, attn_logits), both [B, M, L].  `fast`: ispk_aligner_scores_fast_f32 (bf16 compute path: split-bf16 score
    products, hardware exp / log)."""
    _dev(q_enc, k_enc, text_len, mel_len)
    B, D = q_enc.shape[0], q_enc.shape[2]
    logits = torch.empty((B, M, L), dtype=torch.float32, device=q_enc.device)
    soft = torch.empty((B, M, L), dtype=torch.float32, device=q_enc.device)
    text_len = text_len.to(torch.int64).contiguous()
    mel_len = mel_len.to(torch.int64).contiguous()
    _launch("aligner_scores_kernel<bf16x3>" if fast else "aligner_scores_kernel", 2.0 * B * M * L * D, 4.0 * B * (M * D + L * D + 2 * M * L),
            lib().ispk_aligner_scores_fast_f32 if fast else lib().ispk_aligner_scores_f32, q_enc.data_ptr(), q_enc.stride(0), k_enc.data_ptr(), k_enc.stride(0),
            text_len.data_ptr(), mel_len.data_ptr(), logits.data_ptr(), soft.data_ptr(), B, M, L, D, _stream())
    return soft, logits


def soft_average(attn_soft: Tensor, pitch: Tensor, energy: Tensor, duration: Optional[Tensor], text_len: Tensor) -> Tensor:
    """ispk_soft_average_f32 -> feats [B, L, 3] = (log1p(duration) - or 0 without durations -, pitch target, energy target)."""
    _dev(attn_soft, pitch, energy, duration, text_len)
    B, M, L = attn_soft.shape
    feats = torch.empty((B, L, 3), dtype=torch.float32, device=attn_soft.device)
    dur = None if duration is None else duration.to(torch.int64).contiguous()
    _launch("soft_average_kernel", 0.0, 4.0 * B * M * L, lib().ispk_soft_average_f32, attn_soft.contiguous().data_ptr(),
            pitch.contiguous().data_ptr(), energy.contiguous().data_ptr(), _ptr(dur),
            text_len.to(torch.int64).contiguous().data_ptr(), feats.data_ptr(), B, M, L, _stream())
    return feats


def flow_mix(x0: Tensor, x1: Tensor, t: Tensor, sigma: float):
    """ispk_flow_mix_f32 -> (x_t, flow), both [B, L, C] fp32."""
    _dev(x0, x1, t)
    B, L, C = x1.shape
    x0c, x1c, tc = x0.float().contiguous(), x1.float().contiguous(), t.float().contiguous()
    xt, flow = torch.empty_like(x1c), torch.empty_like(x1c)
    _launch("flow_mix_kernel", 0.0, 16.0 * B * L * C, lib().ispk_flow_mix_f32, x0c.data_ptr(), x1c.data_ptr(), tc.data_ptr(),
            float(sigma), xt.data_ptr(), flow.data_ptr(), B, L, C, _stream())
    return xt, flow


def flow_finish(pred_raw: Tensor, flow: Tensor, x0: Tensor, mask: Tensor):
    """ispk_flow_finish_f32 -> (pred [B,L,C], duration [B,L], loss_ratio [B], loss = mean(loss_ratio) 0-d)."""
    _dev(pred_raw, flow, x0, mask)
    B, L, C = pred_raw.shape
    assert mask.dtype == torch.bool and mask.shape == (B, L)
    pr, fl, x0c, mk = pred_raw.float().contiguous(), flow.contiguous(), x0.float().contiguous(), mask.contiguous()
    pred = torch.empty_like(pr)
    dur = torch.empty((B, L), dtype=torch.float32, device=pr.device)
    ratio = torch.empty((B,), dtype=torch.float32, device=pr.device)
    loss = torch.empty((), dtype=torch.float32, device=pr.device)
    _launch("flow_finish_kernel", 0.0, 20.0 * B * L * C, lib().ispk_flow_finish_f32, pr.data_ptr(), fl.data_ptr(),
            x0c.data_ptr(), mk.data_ptr(), pred.data_ptr(), dur.data_ptr(), ratio.data_ptr(), loss.data_ptr(), B, L, C,
            _stream())
    return pred, dur, ratio, loss


def flow_head(y: Tensor, norm_weight: Tensor, norm_bias: Tensor, norm_eps: float, weight: Tensor, bias: Tensor, flow: Tensor,
              x0: Tensor, mask: Tensor):
    """ispk_flow_head_f32: the predictor's final LayerNorm (row-masked) + 256 -> 3 linear_layer + `flow_finish` on the stack's raw
    output rows y [B, L, 256] -> (pred [B,L,3], duration [B,L], loss_ratio [B], loss 0-d), two launches instead of three."""
    _dev(y, norm_weight, norm_bias, weight, bias, flow, x0, mask)
    B, L, D = y.shape
    C = weight.shape[0]
    assert y.dtype == torch.float32 and y.stride(2) == 1 and y.stride(0) == L * y.stride(1) and weight.shape == (C, D) and weight.is_contiguous()
    assert mask.dtype == torch.bool and mask.shape == (B, L)
    fl, x0c, mk = flow.contiguous(), x0.float().contiguous(), mask.contiguous()
    pred = torch.empty((B, L, C), dtype=torch.float32, device=y.device)
    dur = torch.empty((B, L), dtype=torch.float32, device=y.device)
    ratio = torch.empty((B,), dtype=torch.float32, device=y.device)
    loss = torch.empty((), dtype=torch.float32, device=y.device)
    ws = torch.empty((2 * B * ((L + 15) // 16),), dtype=torch.float32, device=y.device)
    _launch("flow_head_kernels", 0.0, 4.0 * B * L * D, lib().ispk_flow_head_f32, y.data_ptr(), y.stride(1), norm_weight.data_ptr(),
            norm_bias.data_ptr(), float(norm_eps), weight.data_ptr(), bias.data_ptr(), fl.data_ptr(), x0c.data_ptr(), mk.data_ptr(),
            pred.data_ptr(), dur.data_ptr(), ratio.data_ptr(), loss.data_ptr(), ws.data_ptr(), B, L, D, C, _stream())
    return pred, dur, ratio, loss


def flow_euler(x_t: Tensor, velocity: Tensor, dt: float, mask: Optional[Tensor] = None) -> Tensor:
    """ispk_flow_euler_f32: x_t + velocity * dt [* mask[..., None]] (one Euler step of the flow predictor's `infer`)."""
    _dev(x_t, velocity, mask)
    B, L, C = x_t.shape
    xc, vc = x_t.float().contiguous(), velocity.float().contiguous()
    out = torch.empty_like(xc)
    if mask is not None:
        mask = mask.contiguous()
        assert mask.dtype == torch.bool and mask.shape == (B, L)
    _launch("flow_euler_kernel", 0.0, 12.0 * B * L * C, lib().ispk_flow_euler_f32, xc.data_ptr(), vc.data_ptr(), float(dt),
            _ptr(mask), out.data_ptr(), B, L, C, _stream())
    return out


def infer_features(pred: Tensor, duration_target: Optional[Tensor], pitch_target: Optional[Tensor],
                   energy_target: Optional[Tensor], duration_factor: float = 1.0, pitch_factor: float = 1.0,
                   pitch_delta: float = 0.0, energy_factor: float = 1.0, energy_delta: float = 0.0):
    """ispk_infer_features_f32: pred [B, L, 3] -> (duration fp32 [B, L], features fp32 [B, L, 2])."""
    _dev(pred, duration_target, pitch_target, energy_target)
    B, L, C = pred.shape
    assert C == 3 and pred.dtype == torch.float32
    pc = pred.contiguous()
    dur_f = dur_i = None
    if duration_target is not None:
        assert duration_target.shape == (B, L)
        if duration_target.dtype == torch.int64:
            dur_i = duration_target.contiguous()
        else:
            dur_f = duration_target.float().contiguous()
    pt = None if pitch_target is None else pitch_target.float().reshape(B, L).contiguous()
    et = None if energy_target is None else energy_target.float().reshape(B, L).contiguous()
    duration = torch.empty((B, L), dtype=torch.float32, device=pred.device)
    feats = torch.empty((B, L, 2), dtype=torch.float32, device=pred.device)
    _launch("infer_features_kernel", 0.0, 24.0 * B * L, lib().ispk_infer_features_f32, pc.data_ptr(), _ptr(dur_f), _ptr(dur_i),
            _ptr(pt), _ptr(et), float(duration_factor), float(pitch_factor), float(pitch_delta), float(energy_factor),
            float(energy_delta), duration.data_ptr(), feats.data_ptr(), B, L, _stream())
    return duration, feats


# ------------------------------------------------------------------------------------------------- between the stacks
def embed_tokens(text: Tensor, table: Tensor, text_len: Optional[Tensor] = None, want_mask: bool = True):
    """ispk_embed_tokens_f32: (emb fp32 [B,L,D], mask bool [B,L] | None) - nn.Embedding lookup + the key mask."""
    _dev(text, table, text_len)
    assert text.dtype == torch.int64 and text.ndim == 2 and table.dtype == torch.float32 and table.stride(1) == 1
    B, L = text.shape
    V, D = table.shape
    text = text.contiguous()
    emb = torch.empty((B, L, D), dtype=torch.float32, device=text.device)
    mask = torch.empty((B, L), dtype=torch.bool, device=text.device) if want_mask else None
    if text_len is not None:
        text_len = text_len.to(torch.int64).contiguous()
    _launch("embed_tokens_kernel", 0.0, 8.0 * B * L * D, lib().ispk_embed_tokens_f32, text.data_ptr(), table.data_ptr(),
            table.stride(0), V, _ptr(text_len), emb.data_ptr(), _ptr(mask), B, L, D, _stream())
    return emb, mask


def add_speaker_(x: Tensor, table: Tensor, speaker: Tensor) -> Tensor:
    """ispk_add_speaker_f32: x [B, L, D] += table[speaker] in place, broadcast over L the way the reference's
    `enc_out + self.speaker_embedding(speaker)` broadcasts (model.py:205-207): `speaker` int64 [B, 1] (the collator's field,
    collator.py:59) = one id per utterance, or one element = one id for the whole batch (the notebook's `torch.tensor([id])`)."""
    _dev(x, table, speaker)
    assert x.dtype == torch.float32 and x.ndim == 3 and x.is_contiguous() and table.dtype == torch.float32 and table.stride(1) == 1
    B, L, D = x.shape
    assert table.shape[1] == D and speaker.dtype == torch.int64
    if speaker.numel() == 1:
        stride = 0
    elif speaker.ndim == 2 and tuple(speaker.shape) == (B, 1):
        stride = 1
    else:
        raise ValueError(f"speaker of shape {tuple(speaker.shape)} does not broadcast against enc_out [B={B}, L, D] "
                         "(the reference takes [B, 1] ids or a single id)")
    speaker = speaker.contiguous()
    _launch("add_speaker_kernel", 0.0, 8.0 * B * L * D, lib().ispk_add_speaker_f32, x.data_ptr(), table.data_ptr(), table.stride(0),
            table.shape[0], speaker.data_ptr(), stride, B, L, D, _stream())
    return x


def time_embedding(t: Tensor, inv_freq: Tensor, freq_scale: Tensor, w0: Tensor, b0: Tensor, w1: Tensor, b1: Tensor) -> Tensor:
    """ispk_time_embedding_f32: t [...] -> [..., emb_dim] (sinusoid with the raw position, Linear, SiLU, Linear)."""
    _dev(t, inv_freq, freq_scale, w0, b0, w1, b1)
    tf = t.to(torch.float32).contiguous()
    E, H = w1.shape[0], inv_freq.numel()
    assert w0.shape == (E, 1 + 2 * H) and w1.shape == (E, E) and w0.is_contiguous() and w1.is_contiguous()
    out = torch.empty((*t.shape, E), dtype=torch.float32, device=t.device)
    _launch("time_embedding_kernel", 0.0, 0.0, lib().ispk_time_embedding_f32, tf.data_ptr(), tf.numel(), inv_freq.data_ptr(),
            freq_scale.data_ptr(), H, w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(), E, out.data_ptr(),
            _stream())
    return out


def length_regulate(x: Tensor, durations: Tensor, alignment: Optional[Tensor], frames: int, max_len: int = -1,
                    enc_len: Optional[Tensor] = None, want_mask: bool = True, split_bf16=False):
    """ispk_length_regulate_f32 -> (out fp32 [B, frames, D], dec_len int64 [B], dec_mask bool [B, frames] | None).
    alignment fp32 [B, frames, L] (forward), or None: the soft path generated from the fp32 `durations` (infer).
    `split_bf16`: True = ispk_length_regulate_split_bf16 (the bf16 compute path: three bf16 MFMAs per product, ~2^-16
    relative); "f16" = ispk_length_regulate_split_f16 (the split-fp16 parity path: fp16 terms, fp32-grade)."""
    _dev(x, durations, alignment, enc_len)
    assert x.dtype == torch.float32 and x.ndim == 3
    if x.stride(2) != 1 or x.stride(0) != x.shape[1] * x.stride(1):
        x = x.contiguous()
    B, L, D = x.shape
    if alignment is not None:
        assert alignment.dtype == torch.float32 and alignment.shape == (B, frames, L)
        alignment = alignment.contiguous()
    dur_f = dur_i = None
    dur_cols = L
    if durations.dtype == torch.int64:   # only summed: any [B, cols] with the right row sums (e.g. mel_len as [B, 1])
        assert alignment is not None, "the soft path is generated from fp32 durations"
        dur_i = durations.reshape(B, -1).contiguous()
        dur_cols = dur_i.shape[1]
    else:
        dur_f = durations.to(torch.float32).contiguous()
        assert dur_f.shape == (B, L)
    if enc_len is not None:
        enc_len = enc_len.to(torch.int64).contiguous()
    out = torch.empty((B, frames, D), dtype=torch.float32, device=x.device)
    dec_len = torch.empty((B,), dtype=torch.int64, device=x.device)
    mask = torch.empty((B, frames), dtype=torch.bool, device=x.device) if want_mask else None
    nb = 4.0 * B * (frames * D + L * D + (frames * L if alignment is not None else 0))
    fn = (lib().ispk_length_regulate_split_f16 if split_bf16 == "f16" else
          lib().ispk_length_regulate_split_bf16 if split_bf16 else lib().ispk_length_regulate_f32)
    _launch("length_regulate_kernel<split_f16>" if split_bf16 == "f16" else "length_regulate_kernel<bf16x3>" if split_bf16
            else "length_regulate_kernel", 2.0 * B * frames * L * D, nb, fn, _ptr(alignment),
            _ptr(dur_f), _ptr(dur_i), _ptr(enc_len), x.data_ptr(), x.stride(1), out.data_ptr(), dec_len.data_ptr(),
            _ptr(mask), B, frames, L, D, max_len, dur_cols, _stream())
    return out, dec_len, mask


def cast_bf16(x: Tensor) -> Tensor:
    _dev(x)
    x2 = _rows2d(x)
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _launch("cast_bf16_kernel", 0.0, 6.0 * x2.numel(), lib().ispk_cast_f32_bf16, x2.data_ptr(), x2.stride(0),
            y.data_ptr(), x2.shape[1], x2.shape[0], x2.shape[1], _stream())
    return y


# ------------------------------------------------------------------------------------------------- data movement (csrc/util.hip)
class _Segment(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("n", ctypes.c_int64), ("mode", ctypes.c_int32)]


SEG_COPY, SEG_ADD, SEG_BF16 = 0, 1, 2


class _Stage(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("ld_dst", ctypes.c_int64), ("flags", ctypes.c_int32)]


stage_calls = 0     # how many staging passes have been launched (train/graph.py checks that a capture recorded one)


def stage_weights(items) -> None:
    """ispk_stage_weights: items = [(src fp32 contiguous [rows, cols], dst 2-D view with unit column stride (fp32 or bf16),
    transposed: bool, exp: bool)], 16 per launch.  dst is [rows, cols], or [cols, rows] when transposed."""
    global stage_calls
    if not items:
        return
    stage_calls += 1
    arr = (_Stage * len(items))()
    nbytes = 0.0
    for k, (src, dst, tr, ex) in enumerate(items):
        _dev(src, dst)
        src = src.detach()
        assert src.dtype == torch.float32 and src.ndim == 2 and src.is_contiguous() and dst.ndim == 2 and dst.stride(1) == 1
        assert tuple(dst.shape) == ((src.shape[1], src.shape[0]) if tr else tuple(src.shape)) and dst.dtype in (torch.float32, torch.bfloat16)
        arr[k].src, arr[k].dst, arr[k].rows, arr[k].cols = src.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1]
        arr[k].ld_dst = dst.stride(0)
        arr[k].flags = (1 if tr else 0) | (2 if dst.dtype == torch.bfloat16 else 0) | (4 if ex else 0)
        nbytes += src.numel() * (4.0 + dst.element_size())
    _launch("stage_kernel", 0.0, nbytes, lib().ispk_stage_weights, ctypes.cast(arr, ctypes.c_void_p), len(items), _stream())


def segments(items) -> None:
    """ispk_segments_f32: items = [(src fp32 contiguous, dst contiguous view, mode)], any number, 32 per launch:
    SEG_COPY dst = src, SEG_ADD dst += src, SEG_BF16 dst(bf16) = src."""
    if not items:
        return
    arr = (_Segment * len(items))()
    nbytes = 0.0
    for k, (src, dst, mode) in enumerate(items):
        _dev(src, dst)
        assert src.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
        assert dst.dtype == (torch.bfloat16 if mode == SEG_BF16 else torch.float32)
        arr[k].src, arr[k].dst, arr[k].n, arr[k].mode = src.data_ptr(), dst.data_ptr(), src.numel(), mode
        nbytes += src.numel() * (4.0 + dst.element_size() * (2 if mode == SEG_ADD else 1))
    _launch("segments_kernel", 0.0, nbytes, lib().ispk_segments_f32, ctypes.cast(arr, ctypes.c_void_p), len(items), _stream())


def cat0(tensors, dtype: torch.dtype = torch.float32) -> Tensor:
    """torch.cat(tensors, 0).to(dtype) of contiguous fp32 tensors as one ispk_segments_f32 launch (weights that change every
    training step: the fused [to_q; to_kv] image, the adaptive norms' stacked projections)."""
    srcs = [t.detach() for t in tensors]
    assert dtype in (torch.float32, torch.bfloat16) and all(t.dtype == torch.float32 and t.is_contiguous() for t in srcs)
    rows = sum(t.shape[0] for t in srcs)
    out = torch.empty((rows, *srcs[0].shape[1:]), dtype=dtype, device=srcs[0].device)
    items, r = [], 0
    for t in srcs:
        items.append((t, out[r:r + t.shape[0]], SEG_BF16 if dtype == torch.bfloat16 else SEG_COPY))
        r += t.shape[0]
    segments(items)
    return out


def deliver_grads(pairs) -> list:
    """pairs = [(parameter, gradient | None)] -> the list of gradients to hand to autograd.  A gradient whose parameter's .grad
    is a buffer of an optimizer arena (`FlatParameters` marks it `_ispk_grad_arena`) is written - or added, if something has
    been delivered since the arena was zeroed - into it by ONE segments launch for the whole list, and autograd gets None:
    no AccumulateGrad add per parameter."""
    out, items, seen = [], [], set()
    for p, g in pairs:
        if g is not None and not p.requires_grad:
            # frozen AFTER the arena was built (model.freeze(), row f3): autograd would have dropped this gradient - so do we,
            # instead of writing it into the arena where AdamW would apply it
            out.append(None)
            continue
        buf = p.grad if g is not None else None
        if buf is not None and getattr(buf, "_ispk_grad_arena", False) and g.is_cuda:
            g = g.detach()
            g = g if g.dtype == torch.float32 and g.is_contiguous() else g.float().contiguous()
            if buf.data_ptr() in seen:
                # the same parameter twice in one call: two segments of ONE launch writing one buffer would race - flush first
                segments(items)
                items, seen = [], set()
            seen.add(buf.data_ptr())
            items.append((g, buf, SEG_ADD if getattr(buf, "_ispk_dirty", False) else SEG_COPY))
            buf._ispk_dirty = True
            out.append(None)
        else:
            out.append(g)
    segments(items)
    return out


def zero_(t: Tensor) -> Tensor:
    """ispk_fill_zero on a contiguous tensor."""
    _dev(t)
    assert t.is_contiguous()
    _launch("fill_zero_kernel", 0.0, float(t.numel() * t.element_size()), lib().ispk_fill_zero, t.data_ptr(),
            t.numel() * t.element_size(), _stream())
    return t


def zeros(shape, dtype: torch.dtype = torch.float32, device=None) -> Tensor:
    return zero_(torch.empty(shape, dtype=dtype, device=device))


def scale_(x: Tensor, s_dev: Optional[Tensor] = None, s_host: float = 1.0) -> Tensor:
    """ispk_scale_f32: x *= s_dev[0] * s_host in place (s_dev: a one-element fp32 device tensor or None)."""
    _dev(x, s_dev)
    assert x.dtype == torch.float32 and x.is_contiguous() and (s_dev is None or (s_dev.dtype == torch.float32 and s_dev.numel() == 1))
    _launch("scale_kernel", 0.0, 8.0 * x.numel(), lib().ispk_scale_f32, x.data_ptr(), x.numel(), _ptr(s_dev), float(s_host), _stream())
    return x


def sum_scalars(terms, weights=None) -> Tensor:
    """ispk_sum_scalars_f32 -> 0-dim fp32: sum_i weights[i] * terms[i] (one-element fp32 device tensors), in index order."""
    terms = list(terms)
    _dev(*terms)
    assert 1 <= len(terms) <= 8 and all(t.dtype == torch.float32 and t.numel() == 1 for t in terms)
    ptrs = (ctypes.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
    ws = (ctypes.c_float * len(terms))(*([1.0] * len(terms) if weights is None else [float(w) for w in weights]))
    out = torch.empty((1,), dtype=torch.float32, device=terms[0].device)
    _launch("sum_scalars_kernel", 0.0, 0.0, lib().ispk_sum_scalars_f32, ctypes.cast(ptrs, ctypes.c_void_p),
            ctypes.cast(ws, ctypes.c_void_p), len(terms), out.data_ptr(), _stream())
    return out.reshape(())


def exp_pad(src: Tensor, total: Optional[int] = None) -> Tensor:
    """ispk_exp_pad_f32: exp(src) (fp32, flattened), zero-padded to `total` elements."""
    _dev(src)
    src = src.detach().reshape(-1)
    assert src.dtype == torch.float32 and src.is_contiguous()
    total = src.numel() if total is None else total
    out = torch.empty((total,), dtype=torch.float32, device=src.device)
    _launch("unary_kernel", 0.0, 0.0, lib().ispk_exp_pad_f32, src.data_ptr(), out.data_ptr(), src.numel(), total, _stream())
    return out


def sqrt_scale(src: Tensor, scale: float = 1.0) -> Tensor:
    """ispk_sqrt_scale_f32: sqrt(src) * scale (fp32)."""
    _dev(src)
    assert src.dtype == torch.float32 and src.is_contiguous()
    out = torch.empty_like(src)
    _launch("unary_kernel", 0.0, 0.0, lib().ispk_sqrt_scale_f32, src.data_ptr(), out.data_ptr(), src.numel(), float(scale), _stream())
    return out


def copy2d(src: Tensor, dst: Tensor) -> Tensor:
    """ispk_copy2d_f32: dst[:, :] = src for 2-D fp32 views with unit column stride."""
    _dev(src, dst)
    assert src.dtype == torch.float32 and dst.dtype == torch.float32 and src.ndim == 2 and src.shape == dst.shape
    assert src.stride(1) == 1 and dst.stride(1) == 1
    _launch("copy2d_kernel", 0.0, 8.0 * src.numel(), lib().ispk_copy2d_f32, src.data_ptr(), src.stride(0), dst.data_ptr(),
            dst.stride(0), src.shape[0], src.shape[1], _stream())
    return dst


def permute021(src: Tensor) -> Tensor:
    """ispk_permute021_f32: [A, B, C] fp32 contiguous -> contiguous [A, C, B]."""
    _dev(src)
    src = src.detach()
    assert src.dtype == torch.float32 and src.ndim == 3 and src.is_contiguous()
    A, B, C = src.shape
    out = torch.empty((A, C, B), dtype=torch.float32, device=src.device)
    _launch("permute021_kernel", 0.0, 8.0 * src.numel(), lib().ispk_permute021_f32, src.data_ptr(), out.data_ptr(), A, B, C, _stream())
    return out


def conv_weight_flip(w: Tensor) -> Tensor:
    """ispk_conv_weight_flip_f32: Conv1d weight [O, C, K] -> [C, K * O] with wf[c][(K-1-k) O + o] = w[o][c][k]."""
    _dev(w)
    w = w.detach()
    assert w.dtype == torch.float32 and w.ndim == 3 and w.is_contiguous()
    O, C, K = w.shape
    out = torch.empty((C, K * O), dtype=torch.float32, device=w.device)
    _launch("conv_flip_kernel", 0.0, 8.0 * w.numel(), lib().ispk_conv_weight_flip_f32, w.data_ptr(), out.data_ptr(), O, C, K, _stream())
    return out


def draw_seed() -> int:
    """A 62-bit seed for one launch group's dropout masks from torch's CPU generator (`torch.manual_seed(s)` reproduces a
    run): a host-side draw, no device tensor and no device round trip."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


# ------------------------------------------------------------------------------------------------- training step (row f2)
def transpose(x: Tensor) -> Tensor:
    """ispk_transpose_f32: y[c, r] = x[r, c] (fp32 matrix; weights for dX = dY . W through the NT GEMM)."""
    _dev(x)
    assert x.dtype == torch.float32 and x.ndim == 2 and x.stride(1) == 1
    y = torch.empty((x.shape[1], x.shape[0]), dtype=torch.float32, device=x.device)
    _launch("transpose_kernel", 0.0, 8.0 * x.numel(), lib().ispk_transpose_f32, x.data_ptr(), x.stride(0), y.data_ptr(),
            y.stride(0), x.shape[0], x.shape[1], _stream())
    return y


_TN_WORKSPACE_FLOATS = 48 << 20     # 192 MB: up to 64+ row ranges of the largest weight (1536 x 384)
_workspaces: dict = {}


def drop_workspace(key) -> None:
    """Forget the scratch buffer of one (device index, stream) - a HIP graph's capture stream when the graph is destroyed."""
    _workspaces.pop(key, None)


def workspace(device, floats: int) -> Tensor:
    """Scratch for the backward kernels' partial sums, one buffer per (device, CURRENT STREAM): launches on one stream use it
    in order; a backward node that autograd runs on another stream (its forward ran there) gets its own buffer instead of
    racing on the partial sums.  Grown on demand, reused."""
    key = (torch.device(device).index or 0, torch.cuda.current_stream(device).cuda_stream if torch.cuda.is_available() else 0)
    w = _workspaces.get(key)
    if w is None or w.numel() < floats:
        w = _workspaces[key] = torch.empty((max(floats, _TN_WORKSPACE_FLOATS),), dtype=torch.float32, device=device)
    return w


def gemm_tn(a: Tensor, b: Tensor, row_mask: Optional[Tensor] = None, out: Optional[Tensor] = None,
            accumulate: bool = False, bf16: bool = False) -> Tensor:
    """ispk_gemm_tn_f32: C[N1, N2] (+)= sum_m mask[m] a[m, N1] b[m, N2] - the weight gradient dY^T . X of a Linear.
    `bf16`: ispk_gemm_tn_bf16, the operands rounded to bf16 in flight (autocast's weight gradient), fp32 accumulation."""
    _dev(a, b, row_mask, out)
    a2, b2 = _rows2d(a), _rows2d(b)
    in16 = a2.dtype == torch.bfloat16
    assert a2.dtype == b2.dtype and a2.dtype in (torch.float32, torch.bfloat16) and a2.shape[0] == b2.shape[0]
    M, N1 = a2.shape
    N2 = b2.shape[1]
    if out is None:
        assert not accumulate
        out = torch.empty((N1, N2), dtype=torch.float32, device=a.device)
    assert out.shape == (N1, N2) and out.stride(1) == 1 and out.dtype == torch.float32
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).contiguous()
        assert row_mask.dtype == torch.bool and row_mask.numel() == M
    ws = workspace(a.device, N1 * N2)
    fn = lib().ispk_gemm_tn_b16 if in16 else (lib().ispk_gemm_tn_bf16 if bf16 else lib().ispk_gemm_tn_f32)
    _launch(f"gemm_tn_{'b16_' if in16 else ('bf16_' if bf16 else '')}kernel<{N1}x{N2}>", 2.0 * M * N1 * N2,
            float(a2.element_size()) * (a2.numel() + b2.numel()) + 4.0 * out.numel(), fn, a2.data_ptr(), a2.stride(0), b2.data_ptr(), b2.stride(0), out.data_ptr(), out.stride(0), M,
            N1, N2, _ptr(row_mask), int(accumulate), ws.data_ptr(), ws.numel(), _stream())
    return out


def gemm_gelu_train(x: Tensor, w: Tensor, dropout_p: float = 0.0, seed: int = 0):
    """ispk_gemm_bf16_gelu_train -> (u, a): u = x @ w^T and a = dropout(gelu(u)), both bf16, from ONE launch (the first Linear
    of a feed-forward block in an AMP training step: `gemm(x, w)` followed by `gelu(u, dropout_p, seed)`, bit for bit).
    x bf16 [..., K], w bf16 [N, K], K = 256 / 384."""
    _dev(x, w)
    x2 = _rows2d(x)
    M, K = x2.shape
    N = w.shape[0]
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.shape[1] == K and w.stride(1) == 1
    u = torch.empty((*x.shape[:-1], N), dtype=torch.bfloat16, device=x.device)
    a = torch.empty_like(u)
    _launch(f"gemm_bf16_panel_kernel<{K // 64},gelu_train>", 2.0 * M * N * K, 2.0 * (M * K + N * K + 2 * M * N), lib().ispk_gemm_bf16_gelu_train,
            x2.data_ptr(), x2.stride(0), w.data_ptr(), w.stride(0), u.data_ptr(), N, a.data_ptr(), N, M, N, K, dropout_p,
            seed & 0xFFFFFFFFFFFFFFFF, _stream())
    return u, a


def gemm_gelu_bwd(dy: Tensor, w2_t: Tensor, u: Tensor, mask: Optional[Tensor] = None, dropout_p: float = 0.0, seed: int = 0) -> Tensor:
    """ispk_gemm_bf16_gelu_bwd -> du = (mask dy @ w2_t^T) * gelu'(u) * [keep / (1 - p)] (bf16): the feed-forward backward's
    `gemm(dy, w2_t, mask=mask, flags=EP_MASK_OUT)` + `gelu_bwd(da, u, dropout_p=, seed=)` as ONE launch, bit for bit.
    dy bf16 [..., K], w2_t bf16 [N, K] (= W2^T rows), u bf16 [..., N]."""
    _dev(dy, w2_t, u, mask)
    d2, u2 = _rows2d(dy), _rows2d(u)
    M, K = d2.shape
    N = w2_t.shape[0]
    assert dy.dtype == torch.bfloat16 and w2_t.dtype == torch.bfloat16 and u.dtype == torch.bfloat16 and u2.shape == (M, N) and u2.is_contiguous()
    if mask is not None:
        mask = mask.reshape(-1).contiguous()
        assert mask.dtype == torch.bool and mask.numel() == M
    du = torch.empty_like(u)
    _launch(f"gemm_bf16_panel_kernel<{K // 64},gelu_bwd>", 2.0 * M * N * K, 2.0 * (M * K + N * K + 2 * M * N), lib().ispk_gemm_bf16_gelu_bwd,
            d2.data_ptr(), d2.stride(0), w2_t.data_ptr(), w2_t.stride(0), u2.data_ptr(), N, du.data_ptr(), N, _ptr(mask), M, N, K,
            dropout_p, seed & 0xFFFFFFFFFFFFFFFF, _stream())
    return du


def gemm_batched(a: Tensor, w: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """ispk_gemm_f32_batched: C[i] = a[i] @ w[i]^T for a [batch, M, K], w [batch, N, K] (fp32, unit column strides)
    -> [batch, M, N] (`out`: a view with unit column stride)."""
    _dev(a, w, out)
    assert a.dtype == torch.float32 and w.dtype == torch.float32 and a.ndim == 3 and w.ndim == 3
    assert a.shape[0] == w.shape[0] and a.shape[2] == w.shape[2]
    if a.stride(2) != 1:
        a = a.contiguous()
    if w.stride(2) != 1:
        w = w.contiguous()
    batch, M, K = a.shape
    N = w.shape[1]
    if out is None:
        out = torch.empty((batch, M, N), dtype=torch.float32, device=a.device)
    assert out.shape == (batch, M, N) and out.stride(2) == 1 and out.dtype == torch.float32
    _launch("gemm_f32_kernel<batched>", 2.0 * batch * M * N * K, 4.0 * (a.numel() + w.numel() + out.numel()),
            lib().ispk_gemm_f32_batched, a.data_ptr(), a.stride(1), a.stride(0), w.data_ptr(), w.stride(1), w.stride(0),
            out.data_ptr(), out.stride(1), out.stride(0), batch, M, N, K, _stream())
    return out


def gemm_tn_batched(a: Tensor, b: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """ispk_gemm_tn_batched_f32: C[i] = a[i]^T b[i] for a [batch, M, N1], b [batch, M, N2] (fp32; any batch / row strides,
    unit column stride) -> [batch, N1, N2] (`out`: a view with the same freedom)."""
    _dev(a, b, out)
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.ndim == 3 and b.ndim == 3 and a.shape[:2] == b.shape[:2]
    if a.stride(2) != 1:
        a = a.contiguous()
    if b.stride(2) != 1:
        b = b.contiguous()
    batch, M, N1 = a.shape
    N2 = b.shape[2]
    if out is None:
        out = torch.empty((batch, N1, N2), dtype=torch.float32, device=a.device)
    assert out.shape == (batch, N1, N2) and out.stride(2) == 1 and out.dtype == torch.float32
    ws = workspace(a.device, batch * N1 * N2)
    _launch("gemm_tn_kernel<batched>", 2.0 * batch * M * N1 * N2, 4.0 * (a.numel() + b.numel() + out.numel()),
            lib().ispk_gemm_tn_batched_f32, a.data_ptr(), a.stride(1), a.stride(0), b.data_ptr(), b.stride(1), b.stride(0),
            out.data_ptr(), out.stride(1), out.stride(0), batch, M, N1, N2, None, 0, ws.data_ptr(), ws.numel(), _stream())
    return out


def aligner_scores_bwd(attn_logits: Tensor, attn_soft: Tensor, d_soft: Optional[Tensor], d_logits: Optional[Tensor],
                       text_len: Tensor, mel_len: Tensor, scale: float):
    """ispk_aligner_scores_bwd_f32 -> (dS [B, M, L4], dSt [B, L, M4]) zero-padded to multiples of 4 columns."""
    _dev(attn_logits, attn_soft, d_soft, d_logits, text_len, mel_len)
    B, M, L = attn_logits.shape
    L4, M4 = (L + 3) // 4 * 4, (M + 3) // 4 * 4
    dS = zeros((B, M, L4), torch.float32, attn_logits.device)
    dSt = zeros((B, L, M4), torch.float32, attn_logits.device)
    cg = lambda t: None if t is None else t.float().contiguous()       # noqa: E731
    d_soft, d_logits = cg(d_soft), cg(d_logits)
    _launch("aligner_scores_bwd_kernel", 0.0, 4.0 * B * M * L * 6, lib().ispk_aligner_scores_bwd_f32, attn_logits.contiguous().data_ptr(),
            attn_soft.contiguous().data_ptr(), _ptr(d_soft), _ptr(d_logits), text_len.to(torch.int64).contiguous().data_ptr(),
            mel_len.to(torch.int64).contiguous().data_ptr(), dS.data_ptr(), L4, dSt.data_ptr(), M4, B, M, L, scale, _stream())
    return dS, dSt


def masked_instnorm_bwd(y: Tensor, d_out: Tensor, weight: Tensor, lengths: Tensor, eps: float = 1e-5):
    """ispk_masked_instnorm_bwd_f32: y, d_out [B, T+4, C] (row t = frame t) -> (d_y like y, d_weight [C], d_bias [C])."""
    _dev(y, d_out, weight, lengths)
    B, TP, C = y.shape
    assert y.is_contiguous() and d_out.is_contiguous() and d_out.shape == y.shape and y.dtype == torch.float32
    d_y = torch.empty_like(y)
    dw, db = torch.empty((C,), dtype=torch.float32, device=y.device), torch.empty((C,), dtype=torch.float32, device=y.device)
    ws = workspace(y.device, 2 * B * C)
    _launch("masked_instnorm_bwd_kernel", 0.0, 4.0 * y.numel() * 5, lib().ispk_masked_instnorm_bwd_f32, y.data_ptr(), d_out.data_ptr(),
            weight.data_ptr(), lengths.to(torch.int64).contiguous().data_ptr(), d_y.data_ptr(), dw.data_ptr(), db.data_ptr(),
            ws.data_ptr(), ws.numel(), B, TP - 4, C, eps, _stream())
    return d_y, dw, db


def soft_average_bwd(attn_soft: Tensor, pitch: Tensor, energy: Tensor, d_feats: Tensor, text_len: Tensor) -> Tensor:
    """ispk_soft_average_bwd_f32 -> d attn_soft [B, M, L]."""
    _dev(attn_soft, pitch, energy, d_feats, text_len)
    B, M, L = attn_soft.shape
    d = torch.empty_like(attn_soft)
    ws = workspace(attn_soft.device, 3 * B * L)
    _launch("soft_average_bwd_kernels", 0.0, 4.0 * attn_soft.numel() * 3, lib().ispk_soft_average_bwd_f32,
            attn_soft.contiguous().data_ptr(), pitch.float().contiguous().data_ptr(), energy.float().contiguous().data_ptr(),
            d_feats.float().contiguous().data_ptr(), text_len.to(torch.int64).contiguous().data_ptr(), ws.data_ptr(), ws.numel(),
            d.data_ptr(), 0, B, M, L, _stream())
    return d


def layernorm_bwd(x: Tensor, dy: Tensor, gamma: Optional[Tensor], row_mask: Optional[Tensor] = None,
                  dx: Optional[Tensor] = None, add_to_dx: bool = False, want_param_grads: bool = True, eps: float = 1e-5,
                  bf16_copy: bool = False):
    """ispk_layernorm_bwd_f32 -> (dx, dgamma | None, dbeta | None).  `dx` given + add_to_dx: accumulated in place (the
    residual branch's gradient is already there).  `bf16_copy` (ispk_layernorm_bwd_dual_f32): a fourth result, dx once more
    as bf16 rows - the operand an AMP step's next dX GEMM and weight gradient take, without a cast launch."""
    _dev(x, dy, gamma, row_mask, dx)
    x2, dy2 = _rows2d(x), _rows2d(dy)
    rows, D = x2.shape
    assert x2.dtype == torch.float32 and dy2.dtype == torch.float32 and dy2.shape == x2.shape
    if dx is None:
        assert not add_to_dx
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    dx2 = _rows2d(dx)
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).contiguous()
        assert row_mask.dtype == torch.bool and row_mask.numel() == rows
    dg = db = None
    ws = None
    if want_param_grads:
        dg = torch.empty((D,), dtype=torch.float32, device=x.device)
        db = torch.empty((D,), dtype=torch.float32, device=x.device)
        ws = workspace(x.device, ((rows + 63) // 64) * 2 * D)
    if bf16_copy:
        dx16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        _launch(f"layernorm_bwd_kernel<{D // 64}>", 0.0, 4.0 * rows * D * (3.5 + int(add_to_dx)), lib().ispk_layernorm_bwd_dual_f32,
                x2.data_ptr(), x2.stride(0), dy2.data_ptr(), dy2.stride(0), _ptr(gamma), _ptr(row_mask), dx2.data_ptr(),
                dx2.stride(0), int(add_to_dx), _ptr(dg), _ptr(db), _ptr(ws), ws.numel() if ws is not None else 0, rows, D, eps,
                dx16.data_ptr(), D, _stream())
        return dx, dg, db, dx16
    _launch(f"layernorm_bwd_kernel<{D // 64}>", 0.0, 4.0 * rows * D * (3 + int(add_to_dx)), lib().ispk_layernorm_bwd_f32,
            x2.data_ptr(), x2.stride(0), dy2.data_ptr(), dy2.stride(0), _ptr(gamma), _ptr(row_mask), dx2.data_ptr(),
            dx2.stride(0), int(add_to_dx), _ptr(dg), _ptr(db), _ptr(ws), ws.numel() if ws is not None else 0, rows, D, eps,
            _stream())
    return dx, dg, db


def gelu(u: Tensor, dropout_p: float = 0.0, seed: int = 0, out_dtype: torch.dtype = torch.float32) -> Tensor:
    """ispk_gelu_f32 / ispk_gelu_f32_bf16: exact-erf GELU as its own pass (the training forward keeps u), optionally followed
    by dropout; `out_dtype=torch.bfloat16`: the result as the bf16 operand an AMP step's second Linear takes."""
    _dev(u)
    assert u.dtype in (torch.float32, torch.bfloat16) and u.is_contiguous() and out_dtype in (torch.float32, torch.bfloat16)
    if u.dtype == torch.bfloat16:      # ispk_gelu_bf16: the pre-activation itself is bf16 (autocast's Linear output)
        assert out_dtype == torch.bfloat16
        fn = lib().ispk_gelu_bf16
    else:
        fn = lib().ispk_gelu_f32 if out_dtype == torch.float32 else lib().ispk_gelu_f32_bf16
    a = torch.empty(u.shape, dtype=out_dtype, device=u.device)
    _launch("gelu_fwd_kernel", 0.0, float(u.element_size() + a.element_size()) * u.numel(), fn, u.data_ptr(), a.data_ptr(), u.numel(),
            dropout_p, seed & 0xFFFFFFFFFFFFFFFF, _stream())
    return a


def dropout_mask(n: int, dropout_p: float, seed: int, device) -> Tensor:
    """ispk_dropout_mask_u8: the keep mask the kernels evaluate for element indices 0 .. n-1 (bool [n])."""
    out = torch.empty((n,), dtype=torch.bool, device=device)
    _dev(out)
    _check(lib().ispk_dropout_mask_u8(out.data_ptr(), n, dropout_p, seed & 0xFFFFFFFFFFFFFFFF, _stream()), "ispk_dropout_mask_u8")
    return out


def alibi_mqa_attention_train(qkv: Tensor, heads: int, slopes: Tensor, key_len: Optional[Tensor], dropout_p: float, seed: int):
    """-> (o [B, N, heads*64] in qkv's dtype, lse fp32 [B, heads, N]): attention with dropped probabilities, row statistics
    kept for the backward.  fp32 qkv: ispk_alibi_mqa_attn_train_f32.  bf16 qkv (the step under autocast):
    ispk_alibi_mqa_attn_train_bf16 - bf16 MFMAs, K / V staged in LDS, bf16 o."""
    _dev(qkv, slopes, key_len)
    B, N, W = qkv.shape
    b16 = qkv.dtype == torch.bfloat16
    assert W == heads * 64 + 128 and qkv.dtype in (torch.float32, torch.bfloat16) and qkv.is_contiguous()
    slopes = slopes.to(torch.float32).contiguous()
    if key_len is not None:
        key_len = key_len.to(torch.int64).contiguous()
    o = torch.empty((B, N, heads * 64), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, heads, N), dtype=torch.float32, device=qkv.device)
    _launch("attn_train_fwd_bf16_kernel" if b16 else "attn_train_fwd_kernel", 4.0 * B * heads * N * N * 64,
            float(qkv.element_size()) * (qkv.numel() + o.numel()),
            lib().ispk_alibi_mqa_attn_train_bf16 if b16 else lib().ispk_alibi_mqa_attn_train_f32, qkv.data_ptr(), W, slopes.data_ptr(),
            _ptr(key_len), o.data_ptr(), heads * 64, lse.data_ptr(), B, N, heads, dropout_p, seed & 0xFFFFFFFFFFFFFFFF, _stream())
    return o, lse


def gelu_bwd(da: Tensor, u: Tensor, out: Optional[Tensor] = None, dropout_p: float = 0.0, seed: int = 0) -> Tensor:
    """ispk_gelu_bwd_f32: du = da * [keep / (1 - p)] * gelu'(u) (exact erf); `out` may alias `da`."""
    _dev(da, u, out)
    assert da.dtype in (torch.float32, torch.bfloat16) and u.dtype in (torch.float32, torch.bfloat16) and da.is_contiguous() and u.is_contiguous()
    assert da.shape == u.shape and (u.dtype == torch.float32 or da.dtype == torch.bfloat16)
    if out is None:
        out = torch.empty_like(da)
    assert out.dtype == da.dtype
    b16 = da.dtype == torch.bfloat16      # ispk_gelu_bwd_bf16: da and du are bf16 GEMM operands of an AMP step (_b16: u bf16 too)
    fn = (lib().ispk_gelu_bwd_b16 if u.dtype == torch.bfloat16 else lib().ispk_gelu_bwd_bf16) if b16 else lib().ispk_gelu_bwd_f32
    _launch("gelu_bwd_kernel", 0.0, float(u.element_size() + 2 * da.element_size()) * da.numel(), fn,
            da.data_ptr(), u.data_ptr(), out.data_ptr(), da.numel(), dropout_p, seed & 0xFFFFFFFFFFFFFFFF, _stream())
    return out


def alibi_mqa_attention_bwd(qkv: Tensor, o: Tensor, d_o: Tensor, heads: int, slopes: Tensor, key_len: Optional[Tensor],
                            lse: Optional[Tensor] = None, dropout_p: float = 0.0, seed: int = 0):
    """-> (dqkv like qkv, dlogslopes fp32 [heads]).  fp32 tensors: ispk_alibi_mqa_attn_bwd_f32 (`lse` from the training
    forward saves the statistics pass).  bf16 tensors (the step under autocast): ispk_alibi_mqa_attn_bwd_bf16, `lse` required.
    dropout_p / seed must be the forward's."""
    _dev(qkv, o, d_o, slopes, key_len, lse)
    B, N, W = qkv.shape
    b16 = qkv.dtype == torch.bfloat16
    assert W == heads * 64 + 128 and qkv.dtype in (torch.float32, torch.bfloat16) and qkv.is_contiguous()
    assert o.shape == (B, N, heads * 64) and d_o.shape == o.shape and o.dtype == qkv.dtype and d_o.dtype == qkv.dtype
    o, d_o = o.contiguous(), d_o.contiguous()
    slopes = slopes.to(torch.float32).contiguous()
    if key_len is not None:
        key_len = key_len.to(torch.int64).contiguous()
    dqkv = torch.empty_like(qkv)
    dls = torch.empty((heads,), dtype=torch.float32, device=qkv.device)
    if b16:
        assert lse is not None and lse.dtype == torch.float32 and lse.shape == (B, heads, N) and lse.is_contiguous()
        ws = workspace(qkv.device, B * heads * N + 2 * heads * B * ((N + 63) // 64))
        _launch("attn_bwd_bf16_kernels", 10.0 * B * heads * N * N * 64, 2.0 * (2 * qkv.numel() + 2 * o.numel()),
                lib().ispk_alibi_mqa_attn_bwd_bf16, qkv.data_ptr(), W, o.data_ptr(), d_o.data_ptr(), heads * 64, slopes.data_ptr(),
                _ptr(key_len), lse.data_ptr(), dqkv.data_ptr(), dls.data_ptr(), ws.data_ptr(), ws.numel(), B, N, heads, dropout_p,
                seed & 0xFFFFFFFFFFFFFFFF, _stream())
        return dqkv, dls
    tiles = (N + 31) // 32
    ws = workspace(qkv.device, 2 * B * heads * N + heads * B * tiles)
    _launch("attn_bwd_kernels", 10.0 * B * heads * N * N * 64, 4.0 * (2 * qkv.numel() + 2 * o.numel()),
            lib().ispk_alibi_mqa_attn_bwd_f32, qkv.data_ptr(), W, o.data_ptr(), d_o.data_ptr(), heads * 64, slopes.data_ptr(),
            _ptr(key_len), dqkv.data_ptr(), dls.data_ptr(), ws.data_ptr(), ws.numel(), B, N, heads, _ptr(lse), dropout_p,
            seed & 0xFFFFFFFFFFFFFFFF, _stream())
    return dqkv, dls


def mel_loss(mel_out: Tensor, mel_target: Tensor, mel_len: Tensor, want_grad: bool = False, grad_out: float = 1.0):
    """ispk_mel_loss_f32 -> (loss fp32 [1], grad fp32 like mel_out | None)."""
    _dev(mel_out, mel_target, mel_len)
    assert mel_out.dtype == torch.float32 and mel_target.dtype == torch.float32 and mel_out.shape == mel_target.shape
    mel_out, mel_target = mel_out.contiguous(), mel_target.contiguous()
    B, C, T = mel_out.shape
    mel_len = mel_len.to(torch.int64).contiguous()
    ratio = torch.empty((B,), dtype=torch.float32, device=mel_out.device)
    loss = torch.empty((1,), dtype=torch.float32, device=mel_out.device)
    grad = torch.empty_like(mel_out) if want_grad else None
    _launch("mel_loss_kernel", 0.0, 4.0 * mel_out.numel() * (2 + int(want_grad)), lib().ispk_mel_loss_f32, mel_out.data_ptr(),
            mel_target.data_ptr(), mel_len.data_ptr(), ratio.data_ptr(), loss.data_ptr(), _ptr(grad), grad_out, B, C, T,
            _stream())
    return loss, grad


def flow_loss_bwd(pred_raw: Tensor, flow: Tensor, mask: Tensor, grad_out: float = 1.0) -> Tensor:
    """ispk_flow_loss_bwd_f32: gradient of the flow loss wrt the predictor's raw output [B, L, C]."""
    _dev(pred_raw, flow, mask)
    pred_raw, flow, mask = pred_raw.contiguous(), flow.contiguous(), mask.contiguous()
    B, L, C = pred_raw.shape
    assert mask.dtype == torch.bool and mask.shape == (B, L) and flow.shape == pred_raw.shape
    d = torch.empty_like(pred_raw)
    _launch("flow_loss_bwd_kernel", 0.0, 12.0 * pred_raw.numel(), lib().ispk_flow_loss_bwd_f32, pred_raw.data_ptr(), flow.data_ptr(),
            mask.data_ptr(), grad_out, d.data_ptr(), B, L, C, _stream())
    return d


def adaln_bwd(x: Tensor, dy: Tensor, scale: Tensor, row_mask: Optional[Tensor], dx: Optional[Tensor], add_to_dx: bool,
              dscale: Tensor, dshift: Tensor, eps: float = 1e-5) -> Tensor:
    """ispk_adaln_bwd_f32: x, dy [B, L, D]; scale / dscale / dshift [B, D] rows (any row stride, unit column stride)."""
    _dev(x, dy, scale, row_mask, dx, dscale, dshift)
    B, L, D = x.shape
    x2, dy2 = _rows2d(x), _rows2d(dy)
    if dx is None:
        assert not add_to_dx
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    dx2 = _rows2d(dx)
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).contiguous()
    assert scale.stride(1) == 1 and dscale.stride(1) == 1 and dshift.stride(1) == 1 and dscale.stride(0) == dshift.stride(0)
    _launch(f"adaln_bwd_kernel<{D // 64}>", 0.0, 4.0 * x2.numel() * (3 + int(add_to_dx)), lib().ispk_adaln_bwd_f32, x2.data_ptr(),
            x2.stride(0), dy2.data_ptr(), dy2.stride(0), scale.data_ptr(), scale.stride(0), _ptr(row_mask), dx2.data_ptr(),
            dx2.stride(0), int(add_to_dx), dscale.data_ptr(), dshift.data_ptr(), dscale.stride(0), B, L, D, eps, _stream())
    return dx


def time_embedding_bwd(t: Tensor, inv_freq: Tensor, freq_scale: Tensor, w0: Tensor, b0: Tensor, w1: Tensor, d_out: Tensor):
    """ispk_time_embedding_bwd_f32 -> (dw0, db0, dw1, db1)."""
    _dev(t, inv_freq, freq_scale, w0, b0, w1, d_out)
    t = t.reshape(-1).float().contiguous()
    d_out = d_out.reshape(t.numel(), -1).float().contiguous()
    E, H = w1.shape[0], inv_freq.numel()
    dw0, db0 = torch.empty_like(w0, dtype=torch.float32), torch.empty((E,), dtype=torch.float32, device=t.device)
    dw1, db1 = torch.empty((E, E), dtype=torch.float32, device=t.device), torch.empty((E,), dtype=torch.float32, device=t.device)
    _launch("time_embedding_bwd_kernel", 0.0, 0.0, lib().ispk_time_embedding_bwd_f32, t.data_ptr(), t.numel(),
            inv_freq.contiguous().data_ptr(), freq_scale.data_ptr(), H, w0.contiguous().data_ptr(), b0.data_ptr(),
            w1.contiguous().data_ptr(), E, d_out.data_ptr(), dw0.data_ptr(), db0.data_ptr(), dw1.data_ptr(), db1.data_ptr(), _stream())
    return dw0, db0, dw1, db1


def attn_ctc_loss(attn_logits: Tensor, text_len: Tensor, mel_len: Tensor, blank_logprob: float = -1.0,
                  want_grad: bool = False, grad_out: float = 1.0):
    """ispk_attn_ctc_loss_f32 -> (loss fp32 [1], grad fp32 like attn_logits | None)."""
    _dev(attn_logits, text_len, mel_len)
    assert attn_logits.dtype == torch.float32
    lg = attn_logits.reshape(-1, *attn_logits.shape[-2:]).contiguous()
    B, M, L = lg.shape
    text_len, mel_len = text_len.to(torch.int64).contiguous(), mel_len.to(torch.int64).contiguous()
    s_pad = (2 * L + 1 + 63) // 64 * 64
    ws = workspace(lg.device, B * M + B + 2 * B * M * s_pad)
    loss = torch.empty((1,), dtype=torch.float32, device=lg.device)
    grad = torch.empty_like(lg) if want_grad else None
    _launch("ctc_loss_kernels", 0.0, 4.0 * (lg.numel() * (2 + int(want_grad)) + 4 * B * M * s_pad), lib().ispk_attn_ctc_loss_f32,
            lg.data_ptr(), text_len.data_ptr(), mel_len.data_ptr(), blank_logprob, ws.data_ptr(), ws.numel(), loss.data_ptr(),
            _ptr(grad), grad_out, B, M, L, _stream())
    return loss, (grad.view(attn_logits.shape) if grad is not None else None)


def attn_bin_loss(attn_soft: Tensor, attn_hard: Tensor, eps: float = 1e-6, want_grad: bool = False, grad_out: float = 1.0):
    """ispk_attn_bin_loss_f32 -> (loss fp32 [2] = (loss, number of path cells), grad fp32 like attn_soft | None)."""
    _dev(attn_soft, attn_hard)
    assert attn_soft.dtype == torch.float32 and attn_hard.dtype == torch.int16 and attn_soft.shape == attn_hard.shape
    attn_soft, attn_hard = attn_soft.contiguous(), attn_hard.contiguous()
    B, M, L = attn_soft.shape[0], attn_soft.shape[-2], attn_soft.shape[-1]
    loss = torch.empty((2,), dtype=torch.float32, device=attn_soft.device)
    grad = zeros(attn_soft.shape, attn_soft.dtype, attn_soft.device) if want_grad else None
    ws = workspace(attn_soft.device, 2048)
    _launch("bin_loss_kernels", 0.0, 6.0 * attn_soft.numel(), lib().ispk_attn_bin_loss_f32, attn_soft.data_ptr(),
            attn_hard.data_ptr(), eps, ws.data_ptr(), loss.data_ptr(), _ptr(grad), grad_out, B, M, L, _stream())
    return loss, grad


def mel_grad_rows(dmel: Tensor, mask: Optional[Tensor]) -> Tensor:
    """ispk_mel_grad_rows_f32: [B, C, T] gradient of the mel output -> masked rows [B, T, C] for to_mel's backward."""
    _dev(dmel, mask)
    assert dmel.dtype == torch.float32 and dmel.ndim == 3
    dmel = dmel.contiguous()
    B, C, T = dmel.shape
    if mask is not None:
        mask = mask.contiguous()
        assert mask.dtype == torch.bool and mask.shape == (B, T)
    g = torch.empty((B, T, C), dtype=torch.float32, device=dmel.device)
    _launch("mel_grad_rows_kernel", 0.0, 8.0 * dmel.numel(), lib().ispk_mel_grad_rows_f32, dmel.data_ptr(), _ptr(mask),
            g.data_ptr(), B, C, T, _stream())
    return g


def colsum(x: Tensor, row_mask: Optional[Tensor] = None) -> Tensor:
    """ispk_colsum_f32: column sums of a [rows, cols] fp32 matrix (bias gradients) over the rows `row_mask` keeps, fixed order."""
    _dev(x, row_mask)
    x2 = _rows2d(x)
    assert x2.dtype == torch.float32
    rows, cols = x2.shape
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).contiguous()
        assert row_mask.dtype == torch.bool and row_mask.numel() == rows
    out = torch.empty((cols,), dtype=torch.float32, device=x.device)
    ws = workspace(x.device, 256 * cols)
    _launch("colsum_kernels", 0.0, 4.0 * x2.numel(), lib().ispk_colsum_f32, x2.data_ptr(), x2.stride(0), rows, cols,
            _ptr(row_mask), ws.data_ptr(), ws.numel(), out.data_ptr(), _stream())
    return out


def smallk_wgrad(g: Tensor, x: Tensor) -> Tensor:
    """ispk_smallk_wgrad_f32: out[n, k] = sum_r g[r, n] x[r, k] for a Linear with K <= 8 input features."""
    _dev(g, x)
    g2, x2 = _rows2d(g), _rows2d(x)
    assert g2.dtype == torch.float32 and x2.dtype == torch.float32 and g2.shape[0] == x2.shape[0] and x2.shape[1] <= 8
    rows, N = g2.shape
    K = x2.shape[1]
    out = torch.empty((N, K), dtype=torch.float32, device=g.device)
    ws = workspace(g.device, 256 * N * K)
    _launch("smallk_wgrad_kernels", 2.0 * rows * N * K, 4.0 * (g2.numel() + x2.numel()), lib().ispk_smallk_wgrad_f32, g2.data_ptr(),
            g2.stride(0), x2.data_ptr(), x2.stride(0), rows, N, K, ws.data_ptr(), ws.numel(), out.data_ptr(), _stream())
    return out


def embedding_bwd(ids: Tensor, d_emb: Tensor, vocab: int, padding_idx: int = 0) -> Tensor:
    """ispk_embedding_bwd_f32 -> d_table fp32 [vocab, D]."""
    _dev(ids, d_emb)
    ids = ids.reshape(-1).to(torch.int64).contiguous()
    d2 = _rows2d(d_emb).contiguous()
    assert d2.dtype == torch.float32 and d2.shape[0] == ids.numel()
    out = torch.empty((vocab, d2.shape[1]), dtype=torch.float32, device=d_emb.device)
    _launch("embedding_bwd_kernel", 0.0, 4.0 * d2.numel(), lib().ispk_embedding_bwd_f32, ids.data_ptr(), d2.data_ptr(), ids.numel(),
            d2.shape[1], vocab, padding_idx, out.data_ptr(), out.stride(0), _stream())
    return out


def grad_sqnorm(g: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """ispk_grad_sqnorm_f32: sum of squares of a flat fp32 arena -> fp32 [1] (device)."""
    _dev(g, out)
    assert g.dtype == torch.float32 and g.ndim == 1 and g.is_contiguous()
    if out is None:
        out = torch.empty((1,), dtype=torch.float32, device=g.device)
    part = workspace(g.device, 2048)
    _launch("sqnorm_kernels", 0.0, 4.0 * g.numel(), lib().ispk_grad_sqnorm_f32, g.data_ptr(), g.numel(), part.data_ptr(),
            out.data_ptr(), _stream())
    return out


def set_seed_source(word: Optional[Tensor]) -> None:
    """ispk_set_dropout_seed_source: while set (a one-element int64 DEVICE tensor the caller keeps alive), every dropout kernel
    launched by this process - from the calling thread or from autograd's backward worker - folds that word into its seed
    when it runs: what lets a captured training step draw fresh masks on every replay.  None switches it off."""
    if word is not None:
        _dev(word)
        assert word.dtype == torch.int64 and word.numel() == 1
    _check(lib().ispk_set_dropout_seed_source(None if word is None else word.data_ptr()), "set_dropout_seed_source")


def adam_args(lr: float, betas: tuple, eps: float, weight_decay: float, step: int, max_norm: float = 1.0,
              grad_scale: float = 1.0) -> Tensor:
    """ispk_adam_args_f32 -> the 10 fp32 factors of AdamW step `step` as a pinned host tensor (for a copy to the device
    record that ispk_adamw_f32_dev reads)."""
    buf = (ctypes.c_float * 10)()
    _check(lib().ispk_adam_args_f32(lr, betas[0], betas[1], eps, weight_decay, step, max_norm, grad_scale,
                                    ctypes.cast(buf, ctypes.c_void_p)), "adam_args")
    t = torch.tensor(list(buf), dtype=torch.float32)
    return t.pin_memory() if torch.cuda.is_available() else t


def adamw_dev(p: Tensor, g: Tensor, m: Tensor, v: Tensor, n_decay: int, args_dev: Tensor, grad_sqnorm: Optional[Tensor] = None) -> None:
    """ispk_adamw_f32_dev: `adamw` with the step's factors read from the device record `args_dev` (fp32 [10], adam_args)."""
    _dev(p, g, m, v, args_dev, grad_sqnorm)
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.ndim == 1 and t.is_contiguous() and t.numel() == p.numel()
    assert args_dev.dtype == torch.float32 and args_dev.numel() == 10 and args_dev.is_contiguous()
    _launch("adamw_kernel", 0.0, 28.0 * p.numel(), lib().ispk_adamw_f32_dev, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(),
            p.numel(), n_decay, args_dev.data_ptr(), _ptr(grad_sqnorm), _stream())


def adamw(p: Tensor, g: Tensor, m: Tensor, v: Tensor, n_decay: int, lr: float, betas: tuple, eps: float, weight_decay: float,
          step: int, grad_sqnorm: Optional[Tensor] = None, max_norm: float = 1.0, grad_scale: float = 1.0) -> None:
    """ispk_adamw_f32 over flat fp32 arenas (in place)."""
    _dev(p, g, m, v, grad_sqnorm)
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.ndim == 1 and t.is_contiguous() and t.numel() == p.numel()
    _launch("adamw_kernel", 0.0, 28.0 * p.numel(), lib().ispk_adamw_f32, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(),
            p.numel(), n_decay, lr, betas[0], betas[1], eps, weight_decay, step, _ptr(grad_sqnorm), max_norm, grad_scale,
            _stream())
