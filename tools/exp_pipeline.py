#!/usr/bin/env python3
"""Experiment: two graph instances in flight + variants of the per-step exchange (single process, world_size 1 RCCL)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from isp_tts_amd import synth
from isp_tts_amd.acoustic import AcousticModel
from isp_tts_amd.config import AcousticDims
from isp_tts_amd.dist import MelGatherPipeline
from isp_tts_amd.graph import GraphedForward
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
model = AcousticModel.init(AcousticDims().model_config()).eval()
model.load_state_dict(synth.make_state_dict(), strict=True)
model = model.to(dev); model.set_compute_dtype(torch.bfloat16)
d = {k: v.to(dev) for k, v in synth.make_inputs(64, 100, 512).items()}
lanes = [(GraphedForward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"], d["flow_x0"], d["flow_t"]),
          torch.cuda.Stream(device=dev)) for _ in range(2)]
def run(mode, steps=30):
    gather = MelGatherPipeline(64, 80, 512, dev)
    stage = [torch.empty(64, 80, 512, device=dev) for _ in range(2)]
    outb = [torch.empty(64, 80, 512, device=dev) for _ in range(2)]
    works = [None, None]
    def step(k):
        g, st = lanes[k % 2]
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            out = g.replay()
            if mode == "pipeline":
                gather.submit(out.mel, out.adaptor_output.dec_lengths)
            elif mode == "copyonly":
                stage[k % 2].copy_(out.mel)
            elif mode == "one_gather":
                if works[k % 2] is not None: works[k % 2].wait()
                stage[k % 2].copy_(out.mel)
                works[k % 2] = dist.all_gather_into_tensor(outb[k % 2], stage[k % 2], async_op=True)
            elif mode == "sync_gather":
                dist.all_gather_into_tensor(outb[k % 2], out.mel.contiguous())
    for k in range(6): step(k)
    gather.wait(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps): step(k)
    gather.wait()
    for w in works:
        if w is not None: w.wait()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{mode:12s} {dt * 1e3:.3f} ms/step")
for mode in ("pipeline", "none"):
    run(mode)
print("after dist.barrier():")
dist.barrier(); torch.cuda.synchronize()
for mode in ("pipeline", "none"):
    run(mode)
dist.destroy_process_group()
