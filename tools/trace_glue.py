#!/usr/bin/env python3
"""Which Python lines launch the PyTorch (non-libispk) kernels of one benchmark-shape forward: torch.profiler with stacks."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import synth
from isp_tts_amd.acoustic.model import AcousticModel
from isp_tts_amd.config import AcousticDims
model = AcousticModel.init(AcousticDims().model_config()).eval()
model.load_state_dict(synth.make_state_dict(), strict=True)
model = model.to("cuda")
model.set_compute_dtype(torch.bfloat16)
model.overlap_streams = False
d = {k: v.to("cuda") for k, v in synth.make_inputs(64, 100, 512).items()}
def step():
    with torch.no_grad():
        return model(d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"], flow_noise=d["flow_x0"],
                     flow_time=d["flow_t"])
step(); step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.Counter(); cnt = collections.Counter()
for ka in prof.key_averages(group_by_stack_n=12):
    t = getattr(ka, "self_device_time_total", 0) or 0
    if t <= 0:
        continue
    frames = [f for f in (ka.stack or []) if "isp_tts_amd" in f and "runtime.py" not in f]
    where = frames[0].split("isp_tts_amd/")[-1] if frames else (ka.stack[0] if ka.stack else "?")
    agg[(where, ka.key)] += t
    cnt[(where, ka.key)] += ka.count
for (where, name), t in agg.most_common(70):
    print(f"{t:8.1f} us  x{cnt[(where, name)]:3d}  {name[:36]:36s} {where[:100]}")
