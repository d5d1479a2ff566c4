#!/usr/bin/env python3
"""Sensitivity experiment (WRONG results on purpose): how much of the graphed bf16 step is the LENGTH of the flow predictor's side
chain?  The predictor's AdaptiveLayerNorm launches (two per layer) are replaced by a constant tensor - no launch - and the step is
timed against the real one, interleaved.  Says what folding those norms into the consuming GEMM's prologue could gain at most."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth
from isp_tts_amd.acoustic import AcousticModel
from isp_tts_amd.config import AcousticDims
from isp_tts_amd.graph import GraphedForward

B = int(os.environ.get("B", 64))
model = AcousticModel.init(AcousticDims().model_config()).eval()
model.load_state_dict(synth.make_state_dict(), strict=True)
model = model.to("cuda").requires_grad_(False)
model.set_compute_dtype(torch.bfloat16)
d = {k: v.to("cuda") for k, v in synth.make_inputs(B, 100, 512).items()}
args = (d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"], d["flow_x0"], d["flow_t"])
real = GraphedForward(model, *args)
orig = runtime.layernorm
cache = {}
skipped = [0]


def fake(x, w, b, scale=None, shift=None, *a, **kw):
    if scale is None:
        return orig(x, w, b, scale, shift, *a, **kw)
    skipped[0] += 1
    key = (tuple(x.shape), kw.get("out_dtype", a[3] if len(a) > 3 else torch.float32))
    if key not in cache:
        cache[key] = torch.zeros(x.shape, dtype=key[1], device=x.device)
    return cache[key]


runtime.layernorm = fake
fakeg = GraphedForward(model, *args)
runtime.layernorm = orig
print("adaptive LayerNorm launches skipped per forward (incl. warm-up passes):", skipped[0])
res = {"real": [], "no adaptive-LN launches": []}
for rnd in range(6):
    for name, g in (("real", real), ("no adaptive-LN launches", fakeg)):
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 20 * 1e3)
for name, v in res.items():
    v = sorted(v)
    print(f"{name}: median {v[len(v) // 2]:.4f} ms, min {v[0]:.4f} ms")
