import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import synth
from isp_tts_amd.acoustic import AcousticModel
from isp_tts_amd.config import AcousticDims
from isp_tts_amd.graph import GraphedForward, SegmentedForward
model = AcousticModel.init(AcousticDims().model_config()).eval()
model.load_state_dict(synth.make_state_dict(), strict=True)
model = model.to("cuda").requires_grad_(False)
model.set_compute_dtype(torch.bfloat16)
d = {k: v.to("cuda") for k, v in synth.make_inputs(64, 100, 512).items()}
args = (d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"], d["flow_x0"], d["flow_t"])
gs = {"one graph": GraphedForward(model, *args), "three graphs": SegmentedForward(model, *args),
      "three graphs, side stream high priority": SegmentedForward(model, *args, side_priority=-1)}
res = {k: [] for k in gs}
for rnd in range(6):
    for k, g in gs.items():
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); res[k].append((time.perf_counter() - t0) / 20 * 1e3)
for k, v in res.items():
    v = sorted(v); print(f"{k}: median {v[len(v)//2]:.4f} ms, min {v[0]:.4f} ms")
