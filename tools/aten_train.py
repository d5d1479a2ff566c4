#!/usr/bin/env python3
"""Which PyTorch (ATen) compute ops does one AMP training step still issue?  Prints op -> count.  GPU box."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.utils._python_dispatch import TorchDispatchMode
from isp_tts_amd import synth, train
from isp_tts_amd.acoustic import AcousticModel
from isp_tts_amd.config import AcousticDims

HARMLESS = ("aten.view", "aten.empty", "aten._unsafe_view", "aten.transpose", "aten.slice", "aten.select", "aten.unsqueeze",
            "aten.expand", "aten.detach", "aten.alias", "aten.t.", "aten.permute", "aten.squeeze", "aten.reshape", "aten.as_strided",
            "aten.is_", "aten.size", "aten.stride", "aten.lift_fresh", "aten._reshape_alias", "aten.split", "aten.unbind", "aten.sym_",
            "aten.empty_like", "aten.new_empty", "aten.record_stream")
B = int(os.environ.get("B", 8))
dev = "cuda"
model = AcousticModel.init(AcousticDims().model_config())
model.load_state_dict(synth.make_state_dict(), strict=True)
model = model.to(dev).train()
d = {k: v.to(dev) for k, v in synth.make_inputs(B, 100, 512).items()}
opt = train.FlatAdamW(list(model.parameters()), lr=2e-4, weight_decay=1e-2, grad_clip=1.0)
opt.check_finite = False


def step():
    _, total, _ = train.acoustic_train_forward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"],
                                               flow_noise=d["flow_x0"], flow_time=d["flow_t"], amp=True)
    opt.step(total)


for _ in range(2):
    step()
seen = collections.Counter()
where = {}


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(HARMLESS):
            seen[name] += 1
            if name not in where:
                import traceback
                fr = [f for f in traceback.extract_stack() if "isp_tts_amd" in f.filename]
                where[name] = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:])
        return func(*args, **(kwargs or {}))


with Spy():
    step()
torch.cuda.synchronize()
print(f"{sum(seen.values())} ATen compute ops in one step")
for k, v in seen.most_common():
    print(f"  {v:5d}  {k:45s} {where[k]}")
