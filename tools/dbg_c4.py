#!/usr/bin/env python3
"""Debug harness: several GraphedForward instances of different shapes replayed back to back (what bench.py's config 4 does)."""
import os, sys, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import synth
from isp_tts_amd.acoustic import AcousticModel
from isp_tts_amd.config import AcousticDims
from isp_tts_amd.dist import plan_micro_batches
from isp_tts_amd.graph import GraphedForward
mode = os.environ.get("MODE", "")
dev = torch.device("cuda", 0)
model = AcousticModel.init(AcousticDims().model_config()).eval()
model.load_state_dict(synth.make_state_dict(), strict=True)
model = model.to(dev)
model.set_compute_dtype(torch.bfloat16)
if "nooverlap" in mode:
    model.overlap_streams = False
if "lanes" in mode:     # what bench.py does before config 4: graph instances created, replayed and destroyed
    import gc
    from isp_tts_amd.graph import GraphedForwardLanes
    keep = []
    for nl in (1, 2):
        d0 = {k: v.to(dev) for k, v in synth.make_inputs(64, 100, 512).items()}
        ln = GraphedForwardLanes(model, d0["text"], d0["text_len"], d0["mel"], d0["mel_len"], d0["pitch"], d0["energy"],
                                 d0["flow_x0"], d0["flow_t"], lanes=nl, calibrate=nl > 1)
        for _ in range(10):
            g, st = ln.next_lane()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                g.replay()
        torch.cuda.synchronize()
        if "keep" in mode:
            keep.append(ln)
        del ln, g, st
        gc.collect()
    print("lanes done", flush=True)
full = synth.make_inputs(256, 200, 1024, variable=True)
shards, plans = plan_micro_batches(full["mel_len"].tolist(), full["text_len"].tolist(), 1, 32768)
graphs = []
for idx, m_pad, l_pad in plans[0]:
    ii = torch.tensor(idx)
    mb = {"text": full["text"][ii, :l_pad], "text_len": full["text_len"][ii], "mel": full["mel"][ii, :, :m_pad],
          "mel_len": full["mel_len"][ii], "pitch": full["pitch"][ii, :m_pad], "energy": full["energy"][ii, :m_pad],
          "flow_x0": full["flow_x0"][ii, :l_pad], "flow_t": full["flow_t"][ii]}
    d = {k: v.to(dev) for k, v in mb.items()}
    if "eager" in mode:
        graphs.append(d)
    else:
        graphs.append(GraphedForward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"], d["flow_x0"], d["flow_t"]))
    print("captured", len(idx), m_pad, l_pad, flush=True)
for it in range(8):
    for g in graphs:
        if "eager" in mode:
            o = model(g["text"], g["text_len"], g["mel"], g["mel_len"], g["pitch"], g["energy"], flow_noise=g["flow_x0"], flow_time=g["flow_t"])
        else:
            o = g.replay()
        if "sync" in mode:
            torch.cuda.synchronize()
    print("iter", it, flush=True)
torch.cuda.synchronize()
print("ok", mode, float(o.mel.abs().sum()))
