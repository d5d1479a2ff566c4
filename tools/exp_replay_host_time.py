#!/usr/bin/env python3
"""How long does the HOST spend in one replay of the forward's HIP graph (hipGraphLaunch), against the device time of the step?
If the two are of the same order the device runs packets as the host submits them, and the order in which the runtime submits the
graph's branches decides when the side branch starts (profiles: it starts ~130 us after its dependencies are met)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import synth
from isp_tts_amd.acoustic import AcousticModel
from isp_tts_amd.config import AcousticDims
from isp_tts_amd.graph import GraphedForward

model = AcousticModel.init(AcousticDims().model_config()).eval()
model.load_state_dict(synth.make_state_dict(), strict=True)
model = model.to("cuda").requires_grad_(False)
model.set_compute_dtype(torch.bfloat16)
d = {k: v.to("cuda") for k, v in synth.make_inputs(64, 100, 512).items()}
args = (d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"], d["flow_x0"], d["flow_t"])
for name, overlap in (("one graph with side branches", True), ("linear graph", False)):
    model.overlap_streams = overlap
    g = GraphedForward(model, *args)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    host = []
    t_all = time.perf_counter()
    for _ in range(30):
        t0 = time.perf_counter()
        g.replay()
        host.append(time.perf_counter() - t0)
    t_sub = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    host.sort()
    print(f"{name}: host time per replay median {host[15] * 1e6:.0f} us (min {host[0] * 1e6:.0f}, max {host[-1] * 1e6:.0f}); "
          f"30 replays submitted in {t_sub * 1e3:.2f} ms, finished in {t_tot * 1e3:.2f} ms ({t_tot / 30 * 1e3:.3f} ms per step)")
