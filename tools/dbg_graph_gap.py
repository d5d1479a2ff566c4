"""Graph-structure experiments for the headline step (B = 64 x 512, bf16): where does the ~129 us turnaround between graph
replays come from?  Prints ms per STEP (one forward of one batch) for each variant."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.set_grad_enabled(False)
from isp_tts_amd import synth
from isp_tts_amd.acoustic import AcousticModel
from isp_tts_amd.config import AcousticDims
from isp_tts_amd.graph import GraphedCall, GraphedForward
dev = "cuda"
torch.set_num_threads(16)
model = AcousticModel.init(AcousticDims().model_config()).eval()
model.load_state_dict(synth.make_state_dict(), strict=True)
model = model.to(dev)
model.set_compute_dtype(torch.bfloat16)
i_ = synth.make_inputs(64, 100, 512)
dd = {k: v.to(dev) for k, v in i_.items()}
args = (dd["text"], dd["text_len"], dd["mel"], dd["mel_len"], dd["pitch"], dd["energy"], dd["flow_x0"], dd["flow_t"])
n = 30


def timeit(tag, body, steps_per_call=1):
    for _ in range(5):
        body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        body()
    torch.cuda.synchronize()
    print(f"{tag:64s} {(time.perf_counter() - t0) / n / steps_per_call * 1e3:.3f} ms/step", flush=True)


def fwd():
    return model(*args[:6], flow_noise=args[6], flow_time=args[7])


g0 = GraphedForward(model, *args)
timeit("one graph per step, replayed back to back", g0.replay)

g2 = GraphedCall(lambda: (fwd(), fwd())[1])
timeit("TWO forwards captured in one graph", g2.replay, 2)
g4 = GraphedCall(lambda: (fwd(), fwd(), fwd(), fwd())[3])
timeit("FOUR forwards captured in one graph", g4.replay, 4)

g1 = GraphedForward(model, *args)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
e = [torch.cuda.Event(), torch.cuda.Event()]
e[1].record()
flip = [0]


def alternate_streams():
    k = flip[0]
    s, g = (s0, g0) if k == 0 else (s1, g1)
    s.wait_event(e[k ^ 1])
    with torch.cuda.stream(s):
        g.replay()
        e[k].record(s)
    flip[0] ^= 1


timeit("two instances on two streams, chained by events (sequential)", alternate_streams)
torch.cuda.synchronize()
timeit("eager launches, no graph", fwd)

# ---- does an eagerly launched tail hide the turnaround of the next graph launch?
x = torch.randn(64, 512, 384, device=dev)
dmask = torch.ones(64, 512, dtype=torch.bool, device=dev)
dlen = torch.full((64,), 512, dtype=torch.int64, device=dev)


def dec_eager():
    return model.decoder(x, mask=dmask, key_len=dlen, out_dtype=torch.bfloat16).out


timeit("decoder stack alone, eager launches", dec_eager)
gd = GraphedCall(dec_eager)
timeit("decoder stack alone, graph", gd.replay)


def mix():
    g0.replay()
    dec_eager()


def mix_g():
    g0.replay()
    gd.replay()


timeit("full-forward graph + eager decoder (sum of both)", mix)
timeit("full-forward graph + decoder graph (sum of both)", mix_g)

# ---- lower bound for "side branch fully hidden": the same graph without MAS and without the flow predictor
out0 = g0.replay()
torch.cuda.synchronize()
hard, dur = out0.aligner_output.attn_hard.clone(), out0.aligner_output.attn_hard_duration.clone()
pred_mod = model.temporal_adaptor.predictor
cached = None
orig_pred, orig_mas = pred_mod.forward, model.aligner.binarize_attention_parallel


def fake_pred(*a, **k):
    return cached


def fake_mas(*a, **k):
    return hard, dur


# run once eagerly to obtain the predictor's outputs, then freeze them
res = {}


def spy_pred(*a, **k):
    r = orig_pred(*a, **k)
    res["r"] = r
    return r


pred_mod.forward = spy_pred
fwd()
torch.cuda.synchronize()
cached = res["r"]
pred_mod.forward = fake_pred
model.aligner.binarize_attention_parallel = fake_mas
gn = GraphedForward(model, *args)
timeit("graph WITHOUT MAS and flow predictor (lower bound)", gn.replay)
pred_mod.forward = orig_pred
gm = GraphedForward(model, *args)
timeit("graph without MAS only", gm.replay)
model.aligner.binarize_attention_parallel = orig_mas

from isp_tts_amd.graph import SegmentedForward
sg = SegmentedForward(model, *args)
timeit("THREE graphs: front | side (MAS, predictor) | back", sg.replay)
o_seg, o_one = sg.replay(), g0.replay()
torch.cuda.synchronize()
for name in ("mel",):
    assert torch.equal(o_seg.mel, o_one.mel)
assert torch.equal(o_seg.aligner_output.attn_hard, o_one.aligner_output.attn_hard)
assert torch.equal(o_seg.adaptor_output.log_duration, o_one.adaptor_output.log_duration)
assert torch.equal(o_seg.adaptor_output.losses["flow_loss"] if "flow_loss" in o_seg.adaptor_output.losses else o_seg.adaptor_output.duration,
                   o_one.adaptor_output.losses["flow_loss"] if "flow_loss" in o_one.adaptor_output.losses else o_one.adaptor_output.duration)
print("segmented == single graph: outputs identical")
