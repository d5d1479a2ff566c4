#!/usr/bin/env python3
"""Same-box A/B of the graphed bf16 forward (B = 64 x 100 x 512, one batch in flight) with a module attribute toggled:
    python tools/ab_forward.py temporal_adaptor.predictor.fused_head
prints ms per step for attribute = True / False, interleaved rounds."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import synth
from isp_tts_amd.acoustic import AcousticModel
from isp_tts_amd.config import AcousticDims
from isp_tts_amd.graph import GraphedForward

path = sys.argv[1].split(".")
B = int(os.environ.get("B", 64))
model = AcousticModel.init(AcousticDims().model_config()).eval()
model.load_state_dict(synth.make_state_dict(), strict=True)
model = model.to("cuda").requires_grad_(False)
model.set_compute_dtype(torch.bfloat16)
d = {k: v.to("cuda") for k, v in synth.make_inputs(B, 100, 512).items()}
obj = model
if path[0] in ("TransformerLayer", "FeedForward", "Attention"):      # a class attribute: every layer of every stack
    from isp_tts_amd.modules.transformer import attention, feedforward, transformer
    obj = {"TransformerLayer": transformer.TransformerLayer, "FeedForward": feedforward.FeedForward,
           "Attention": attention.Attention}[path[0]]
    path = path[1:]
for n in path[:-1]:
    obj = getattr(obj, n)
graphs = {}
VALS = [int(v) for v in os.environ["VALS"].split(",")] if os.environ.get("VALS") else [True, False]
for val in VALS:
    setattr(obj, path[-1], val)
    graphs[val] = GraphedForward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"], d["flow_x0"], d["flow_t"])
res = {v: [] for v in VALS}
for rnd in range(6):
    for val, g in graphs.items():
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        res[val].append((time.perf_counter() - t0) / 20 * 1e3)
for val, v in res.items():
    v = sorted(v)
    print(f"{sys.argv[1]} = {val}: median {v[len(v) // 2]:.4f} ms, min {v[0]:.4f} ms")
