"""Kernel-ready weight images of a transformer stack for a training step, all layers in ONE pass (ispk_stage_weights, 16 images
per launch): the fused [to_q; to_kv] rows, the transposed images the NT GEMM wants for dX = dY W, the ALiBi slopes - in bf16
under AMP.  A training step re-stages after every optimizer update; image by image (cat, transpose, cast: ~11 launches per
layer) that was 176 small launches of the step.  Cached per (stack, amp) on the parameters' versions, so the backward of a
step finds the forward's images."""
from __future__ import annotations

import torch

from .. import runtime
from ..staging import StagedWeights


def layer_images(tr, amp: bool) -> list:
    """-> per layer {"wqkv", "wo", "w1", "w2", "wqkv_t", "wo_t", "w1_t", "w2_t": bf16 under AMP else fp32; "slopes": fp32}."""
    layers = list(tr.layers)
    ps = []
    for layer in layers:
        att, ff = layer.attention, layer.feed_forward
        ps += [att.to_q.weight, att.to_kv.weight, att.to_out.weight, ff.net[0].weight, ff.net[3].weight, att.rel_pos.learned_logslopes]

    def build():
        dt = torch.bfloat16 if amp else torch.float32
        items, out = [], []
        for layer in layers:
            att, ff = layer.attention, layer.feed_forward
            wq, wkv, wo, w1, w2 = (p.detach() for p in (att.to_q.weight, att.to_kv.weight, att.to_out.weight, ff.net[0].weight,
                                                        ff.net[3].weight))
            dev, hq, nkv, D = wq.device, wq.shape[0], wkv.shape[0], wq.shape[1]
            new = lambda *shape: torch.empty(shape, dtype=dt, device=dev)        # noqa: E731
            im = {"wqkv": new(hq + nkv, D), "wqkv_t": new(D, hq + nkv), "wo_t": new(wo.shape[1], wo.shape[0]),
                  "w1_t": new(w1.shape[1], w1.shape[0]), "w2_t": new(w2.shape[1], w2.shape[0])}
            items += [(wq, im["wqkv"][:hq], False, False), (wkv, im["wqkv"][hq:], False, False),
                      (wq, im["wqkv_t"][:, :hq], True, False), (wkv, im["wqkv_t"][:, hq:], True, False),
                      (wo, im["wo_t"], True, False), (w1, im["w1_t"], True, False), (w2, im["w2_t"], True, False)]
            if amp:
                im.update(wo=new(*wo.shape), w1=new(*w1.shape), w2=new(*w2.shape))
                items += [(wo, im["wo"], False, False), (w1, im["w1"], False, False), (w2, im["w2"], False, False)]
            else:
                im.update(wo=wo, w1=w1, w2=w2)
            logs = att.rel_pos.learned_logslopes.detach().reshape(1, -1)
            total = max(att.heads, logs.shape[1])
            sl = torch.empty((1, total), dtype=torch.float32, device=dev) if total == logs.shape[1] else \
                runtime.zeros((1, total), torch.float32, dev)
            items.append((logs.contiguous(), sl[:, :logs.shape[1]], False, True))
            im["slopes"] = sl.reshape(-1)
            out.append(im)
        runtime.stage_weights(items)
        return out
    cache = tr.__dict__.get("_train_images")
    if cache is None:
        cache = tr.__dict__["_train_images"] = StagedWeights()
    return cache.get(("images", amp), ps, build)
