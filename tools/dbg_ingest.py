"""Times the phases of ingest.BatchIngest on the GPU box (pageable -> pinned memcpy, H2D on the copy stream)."""
import sys, time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import ingest, synth
dev = "cuda"
B, L, M = 64, 100, 512
i_ = synth.make_inputs(B, L, M)
hb = {"text_vector": i_["text"], "text_vector_len": i_["text_len"], "mel": i_["mel"], "mel_len": i_["mel_len"],
      "pitch": i_["pitch"], "energy": i_["energy"]}
pin = torch.empty(B * 80 * M, dtype=torch.float32, pin_memory=True)
page = torch.empty(B * 80 * M, dtype=torch.float32)
d = torch.empty(B * 80 * M, dtype=torch.float32, device=dev)
src = i_["mel"].reshape(-1)
for name, dst in (("pageable->pinned", pin), ("pageable->pageable", page)):
    for _ in range(2):
        dst.copy_(src)
    t0 = time.perf_counter()
    for _ in range(10):
        dst.copy_(src)
    print(name, f"{(time.perf_counter() - t0) * 100:.3f} ms per 10.5 MB", "threads", torch.get_num_threads())
torch.set_num_threads(1)
t0 = time.perf_counter()
for _ in range(10):
    pin.copy_(src)
print("pageable->pinned, 1 thread", f"{(time.perf_counter() - t0) * 100:.3f} ms")
import numpy as np
t0 = time.perf_counter()
for _ in range(10):
    np.copyto(pin.numpy(), src.numpy())
print("numpy copyto pinned", f"{(time.perf_counter() - t0) * 100:.3f} ms")
for _ in range(2):
    d.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    d.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
print("pinned->device", f"{(time.perf_counter() - t0) * 100:.3f} ms")
t0 = time.perf_counter()
for _ in range(10):
    d.copy_(src)
torch.cuda.synchronize()
print("pageable->device", f"{(time.perf_counter() - t0) * 100:.3f} ms")
ing = ingest.BatchIngest(dev, B, L, M, 80, slots=2)
for _ in range(3):
    ing.submit(hb); ing.get(); ing.done()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    ing.submit(hb); ing.get(); ing.done()
torch.cuda.synchronize()
print("submit+get+done", f"{(time.perf_counter() - t0) * 100:.3f} ms")

# ---- with the graphed forward
from isp_tts_amd.acoustic import AcousticModel
from isp_tts_amd.config import AcousticDims
from isp_tts_amd.graph import GraphedForward
torch.set_num_threads(16)
model = AcousticModel.init(AcousticDims().model_config()).eval()
model.load_state_dict(synth.make_state_dict(), strict=True)
model = model.to(dev)
model.set_compute_dtype(torch.bfloat16)
dd = {k: v.to(dev) for k, v in i_.items()}
g = GraphedForward(model, dd["text"], dd["text_len"], dd["mel"], dd["mel_len"], dd["pitch"], dd["energy"], dd["flow_x0"], dd["flow_t"])
n = 20


def loop(tag, body):
    body(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    body()
    torch.cuda.synchronize()
    print(tag, f"{(time.perf_counter() - t0) / n * 1e3:.3f} ms/step")


def replay_only():
    for _ in range(n):
        g.replay()


def overlapped():
    ing.submit(hb)
    for k in range(n):
        if k + 1 < n:
            ing.submit(hb)
        g(**ingest.model_inputs(ing.get()))
        ing.done()


def overlapped_no_static_copy():
    ing.submit(hb)
    for k in range(n):
        if k + 1 < n:
            ing.submit(hb)
        ing.get()
        g.replay()
        ing.done()


def same_stream():
    for k in range(n):
        for name, _ in ingest.FIELDS:
            ing.dev[0][name].copy_(ing.host[0][name], non_blocking=True)
        g.replay()


def phases():
    ts = [0.0] * 4
    ing.submit(hb)
    for k in range(n):
        t0 = time.perf_counter()
        if k + 1 < n:
            ing.submit(hb)
        t1 = time.perf_counter()
        got = ingest.model_inputs(ing.get())
        t2 = time.perf_counter()
        g(**got)
        t3 = time.perf_counter()
        ing.done()
        t4 = time.perf_counter()
        for i, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4))):
            ts[i] += b - a
    print("host time per step: submit %.3f get %.3f call %.3f done %.3f ms" % tuple(1e3 * t / n for t in ts))


loop("replay only", replay_only)
loop("same stream copies + replay", same_stream)
loop("overlapped (no copy into static)", overlapped_no_static_copy)
loop("overlapped", overlapped)
phases()
torch.cuda.synchronize()
