#!/usr/bin/env python3
"""One AMP training step at the headline batch (64 x 100 x 512): wall time per step, and per-kernel time of one step from HIP
events around every libispk launch (eager).  GPU box."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth, train
from isp_tts_amd.acoustic import AcousticModel
from isp_tts_amd.config import AcousticDims

B = int(os.environ.get("B", 64))
torch.set_num_threads(8)
dev = "cuda"
model = AcousticModel.init(AcousticDims().model_config())
model.load_state_dict(synth.make_state_dict(), strict=True)
model = model.to(dev).train()
d = {k: v.to(dev) for k, v in synth.make_inputs(B, 100, 512).items()}
opt = train.FlatAdamW(list(model.parameters()), lr=2e-4, weight_decay=1e-2, grad_clip=1.0)
opt.check_finite = False
amp = os.environ.get("AMP", "1") == "1"


def step():
    _, total, _ = train.acoustic_train_forward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"],
                                               flow_noise=d["flow_x0"], flow_time=d["flow_t"], amp=amp)
    opt.step(total)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
prof = runtime.LaunchProfiler()
runtime.set_profiler(prof)
step()
torch.cuda.synchronize()
runtime.set_profiler(None)
summ = prof.summary()
tot = sum(v["total_ms"] for v in summ.values())
nl = sum(v["launches"] for v in summ.values())
print(f"B={B} amp={amp}: wall {wall * 1e3:.2f} ms/step; libispk launches {nl}, summed kernel time {tot:.2f} ms")
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])[:40]:
    print(f"  {k:48s} x{v['launches']:4d}  {v['total_ms']:7.3f} ms  avg {v['avg_us']:7.1f} us")
# fp32 GEMMs by shape (algorithmic FLOPs identify the shape)
by = {}
for label, flops, nbytes, e0, e1 in prof.records:
    if label.startswith("gemm_f32") or label.startswith("gemm_tn_kernel"):
        rec = by.setdefault((label, flops, nbytes), [0, 0.0])
        rec[0] += 1
        rec[1] += e0.elapsed_time(e1)
print("fp32 GEMMs by shape:")
for (label, flops, nbytes), (n, ms) in sorted(by.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {label:28s} {flops / 1e9:8.3f} GFLOP {nbytes / 1e6:8.2f} MB  x{n:3d} {ms:7.3f} ms")
# the same step as one HIP graph
if amp:
    batch = {k: d[k] for k in ("text", "text_len", "mel", "mel_len", "pitch", "energy", "flow_x0", "flow_t")}
    g = train.GraphedTrainStep(model, opt, batch, amp=True, warmup=2)
    for _ in range(3):
        g()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g()
    torch.cuda.synchronize()
    print(f"graph replay: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/step")
