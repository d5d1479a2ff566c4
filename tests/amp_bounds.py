"""The stated bounds of the bf16-AMP training step against the fp32 reference / oracle, shared by
tests/test_gpu_train_loop.py (the HIP step) and tests/test_oracle_goldens.py (torch's own bf16 autocast over the oracle on the
CPU: the same bounds hold for it, i.e. they are autocast-grade, not slack for the kernels).

Under AMP every Linear / Conv1d / attention operand is rounded to bf16 (2^-9 relative), sums are fp32.  Measured on MI355X
(tools/measure_amp_bounds.py; B = 2 fixture / B = 64 x 100 x 512) and, for comparison, torch.autocast("cpu", bf16) over the
oracle on the B = 2 fixture:

    quantity                                              HIP AMP step         torch CPU autocast
    four losses, total (relative)                         <= 7e-4              <= 1.5e-3
    gradients outside aligner / slopes (relative RMS)     <= 4.3e-2            <= 1.8e-2
    aligner front-end gradients (relative RMS)            3e-2 / 9.6e-2        8.2e-2
    learned_logslopes (ALiBi) gradients                   up to 0.56 of a      0.17 of a
                                                          4.8e-5 gradient      7e-4 gradient

The aligner's gradients pass through a softmax over sharply peaked bf16-operand scores (and, at B = 64, longer sums); a slope
gradient is a distance-weighted sum over every (query, key) pair of a stack that nearly cancels - its error is governed by the
size of the terms, not of the net, so it is bounded against the LARGEST slope gradient of the model instead of its own norm.
"""
from __future__ import annotations

AMP_LOSS_RTOL = 3e-3               # each loss term and the total, relative
AMP_GRAD_REL_RMS = 6e-2            # ||g - g_ref|| / ||g_ref|| per tensor, outside the two classes below
AMP_GRAD_REL_RMS_ALIGNER = 1.5e-1  # aligner.* tensors
AMP_SLOPE_FAMILY = 3e-2            # learned_logslopes: ||g - g_ref|| <= this x max over all slope tensors of ||g_ref||
# fixture form (tests/golden/train.npz holds per-tensor norms and 192 sampled entries, not whole gradients)
AMP_GRAD_NORM_RTOL = 6e-2          # | ||g|| - ||g_ref|| | / ||g_ref||
AMP_GRAD_SAMPLE = 2.5e-1           # sampled entries: max |g - g_ref| / max |g_ref|


def tensor_class(name: str) -> str:
    return "slope" if name.endswith("learned_logslopes") else ("aligner" if name.startswith("aligner.") else "general")


def check_full_gradients(grads: dict, ref_grads: dict, what: str = "") -> str:
    """grads / ref_grads: name -> tensor (any device).  Asserts the bounds above; -> a one-line summary of the worst cases."""
    slope_scale = max(float(ref_grads[n].double().norm()) for n in ref_grads if tensor_class(n) == "slope")
    worst = {"general": (0.0, ""), "aligner": (0.0, ""), "slope": (0.0, "")}
    for n, ref in ref_grads.items():
        g, ref = grads[n].detach().double().cpu(), ref.detach().double().cpu()
        err = float((g - ref).norm())
        cls = tensor_class(n)
        rel = err / slope_scale if cls == "slope" else err / max(float(ref.norm()), 1e-30)
        worst[cls] = max(worst[cls], (rel, n))
        bound = {"general": AMP_GRAD_REL_RMS, "aligner": AMP_GRAD_REL_RMS_ALIGNER, "slope": AMP_SLOPE_FAMILY}[cls]
        assert rel <= bound, (what, n, cls, rel, bound)
    return (f"{what}: worst relative RMS {worst['general'][0]:.2e} ({worst['general'][1]}), aligner {worst['aligner'][0]:.2e} "
            f"({worst['aligner'][1]}), slopes {worst['slope'][0]:.2e} of the largest slope gradient ({worst['slope'][1]})")
