"""Length/mask helpers with the reference's names and semantics (tts/utils/functions.py:28-79)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor


def min_dtype_value(tensor: Tensor) -> float:
    """functions.py:28-33."""
    return -65504.0 if tensor.dtype == torch.float16 else -3.4028234663852886e+38


def max_dtype_value(tensor: Tensor) -> float:
    """functions.py:36-41."""
    return 65504.0 if tensor.dtype == torch.float16 else 3.4028234663852886e+38


_ARANGE: dict = {}


def _arange(n: int, device) -> Tensor:
    """0 .. n-1 on `device`, built once (a constant; one launch less per mask on the hot path)."""
    key = (n, str(device))
    if key not in _ARANGE:
        _ARANGE[key] = torch.arange(n, device=device)
    return _ARANGE[key]


def get_mask_from_lengths(lengths: Tensor, max_len: Optional[int] = None) -> Tensor:
    """functions.py:61-65.  `max_len=None` reads `lengths.max()` back to the host exactly like the reference;
    pass `max_len` on the hot path to stay asynchronous."""
    max_len = int(lengths.max().item()) if max_len is None else max_len
    return _arange(max_len, lengths.device)[None, :] < lengths[:, None]


def get_float_mask_from_lengths(lengths: Tensor, max_len: Optional[int] = None) -> Tensor:
    """functions.py:68-73: clamp(lengths - index, 0, 1) for fractional lengths."""
    max_len = int(lengths.max().item()) if max_len is None else max_len
    ids = torch.arange(max_len, device=lengths.device)
    return (lengths.unsqueeze(1) - ids).clamp(0., 1.).to(torch.float32)


def get_mask_3d(widths: Tensor, heights: Tensor, max_w: Optional[int] = None, max_h: Optional[int] = None) -> Tensor:
    """functions.py:76-80."""
    return get_mask_from_lengths(widths, max_w).unsqueeze(2) & get_mask_from_lengths(heights, max_h).unsqueeze(1)


def masked_mean(tensor: Tensor, mask: Tensor) -> Tensor:
    """functions.py:44-58."""
    if tensor.ndim == 3 and mask.ndim == 2:
        mask = mask[..., None].expand_as(tensor)
    tensor = tensor.masked_fill(~mask, 0.)
    if tensor.ndim == 3:
        num, den = tensor.sum(-1).sum(-1), mask.sum(-1).sum(-1)
    else:
        num, den = tensor.sum(-1), mask.sum(-1)
    return (num / den.clamp(min=1e-5)).mean()
