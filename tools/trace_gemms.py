#!/usr/bin/env python3
"""Lists the distinct (M, N, K, flags, bias?, resid?, variant) of every bf16 GEMM launch in one benchmark-shape forward."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth
from isp_tts_amd.acoustic.model import AcousticModel
from isp_tts_amd.config import AcousticDims
model = AcousticModel.init(AcousticDims().model_config()).eval()
model.load_state_dict(synth.make_state_dict(), strict=True)
model = model.to("cuda")
model.set_compute_dtype(torch.bfloat16)
d = {k: v.to("cuda") for k, v in synth.make_inputs(64, 100, 512).items()}
seen = collections.Counter()
orig = runtime._launch
def spy(label, flops, nbytes, fn, *args):
    orig(label, flops, nbytes, fn, *args)
    if label == "gemm_bf16_kernel":
        v = runtime.lib().ispk_gemm_bf16_last_variant()
        seen[(args[10], args[11], args[12], args[13], bool(args[6]), bool(args[7]), args[14], v)] += 1
    elif label.startswith("ffn"):
        seen[(label,)] += 1
runtime._launch = spy
with torch.no_grad():
    model(d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"], flow_noise=d["flow_x0"],
          flow_time=d["flow_t"])
torch.cuda.synchronize()
print("count  M N K flags bias resid cpb variant")
for k, c in sorted(seen.items(), key=lambda kv: str(kv[0])):
    print(c, *k)
