"""Per-tensor error of the bf16-AMP training step against (a) the reference's fixture (tests/golden/train.npz, B = 2) and (b)
autograd over the oracle at the bench shape - the numbers behind the bounds stated in tests/test_gpu_train_loop.py.
    python tools/measure_amp_bounds.py [--bench-shape]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from isp_tts_amd import synth, train  # noqa: E402
import test_gpu_train_loop as T  # noqa: E402

sd = synth.make_state_dict()
g = np.load(os.path.join(ROOT, "tests", "golden", "train.npz"))
args, x0, t = T._fixture_batch(g)
names = [str(n) for n in g["names"]]
for amp in (False, True):
    model = T._model(sd)
    params = dict(model.named_parameters())
    _, total, losses = train.acoustic_train_forward(model, *args, flow_noise=x0, flow_time=t, amp=amp)
    total.backward()
    print(f"== fixture B=2, amp={amp}")
    for k, v in list(losses.items()) + [("total", total)]:
        ref = float(g["loss_" + k.replace("/", "_")])
        print(f"  loss {k:26s} {float(v):.6f} vs {ref:.6f}  rel {abs(float(v) - ref) / max(abs(ref), 1e-3):.2e}")
    rows = []
    for i, n in enumerate(names):
        gr = params[n].grad
        f = gr.detach().reshape(-1)
        s = f[::max(1, -(-f.numel() // 192))].cpu()
        ref_norm, scale = float(g["grad_norm"][i]), float(g["grad_absmax"][i])
        rows.append((abs(gr.double().norm().item() - ref_norm) / max(ref_norm, 1e-12),
                     (s - torch.from_numpy(g[f"g{i}"])).abs().max().item() / max(scale, 1e-12), n, ref_norm))
    for e_n, e_s, n, rn in sorted(rows, reverse=True)[:14]:
        print(f"  norm err {e_n:.2e}  sample err {e_s:.2e}  |g| {rn:.2e}  {n}")
    ns = [r for r in rows if "logslopes" not in r[2]]
    print(f"  without slopes: worst norm err {max(r[0] for r in ns):.2e}, worst sample err {max(r[1] for r in ns):.2e}")

if "--bench-shape" in sys.argv:
    inp, ref_terms, ref_total, ref_grads = T._bench_shape_reference(sd)
    d = {k: v.to("cuda") for k, v in inp.items()}
    for amp in (False, True):
        model = T._model(sd)
        _, total, losses = train.acoustic_train_forward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"], d["energy"],
                                                        flow_noise=d["flow_x0"], flow_time=d["flow_t"], amp=amp)
        total.backward()
        print(f"== bench shape B=64, amp={amp}")
        for k, v in ref_terms.items():
            print(f"  loss {k:26s} {float(losses[k]):.6f} vs {v:.6f}  rel {abs(float(losses[k]) - v) / max(abs(v), 1e-3):.2e}")
        rows = []
        for n, p in model.named_parameters():
            gr, ref = p.grad.double().cpu(), ref_grads[n].double()
            rows.append((float((gr - ref).norm() / ref.norm().clamp_min(1e-30)), n, float(ref.norm())))
        for e, n, rn in sorted(rows, reverse=True)[:14]:
            print(f"  rel RMS {e:.2e}  |g| {rn:.2e}  {n}")
        ns = [r for r in rows if "logslopes" not in r[1]]
        print(f"  without slopes: worst rel RMS {max(r[0] for r in ns):.2e}")
