#!/usr/bin/env python3
"""Per-call timing and shapes of linear_small / layernorm / bmm-like small launches in one forward (eager, HIP events)."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth
from isp_tts_amd.acoustic.model import AcousticModel
from isp_tts_amd.config import AcousticDims
model = AcousticModel.init(AcousticDims().model_config()).eval()
model.load_state_dict(synth.make_state_dict(), strict=True)
model = model.to("cuda")
model.set_compute_dtype(torch.bfloat16)
model.overlap_streams = False
d = {k: v.to("cuda") for k, v in synth.make_inputs(64, 100, 512).items()}
rec = []
orig = runtime._launch
want = sys.argv[1] if len(sys.argv) > 1 else "linear_small"
def spy(label, flops, nbytes, fn, *args):
    if want in label or (want == "gemm" and label == "gemm_bf16_kernel"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(label, flops, nbytes, fn, *args); e1.record()
        if label == "gemm_bf16_kernel":
            label = f"gemm variant {runtime.lib().ispk_gemm_bf16_last_variant()}"
        rec.append((label, [a for a in args if isinstance(a, int) and 0 < a < 10**7], e0, e1))
    else:
        orig(label, flops, nbytes, fn, *args)
runtime._launch = spy
def step():
    with torch.no_grad():
        model(d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"], flow_noise=d["flow_x0"], flow_time=d["flow_t"])
step(); rec.clear(); step(); torch.cuda.synchronize()
for label, ints, e0, e1 in rec:
    print(f"{e0.elapsed_time(e1) * 1e3:8.1f} us  {label}  {ints}")
