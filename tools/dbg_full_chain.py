"""Debug: gradients arriving at attn_soft / attn_logits in the full training chain, product against the oracle in fp32 and fp64."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import synth, train
from isp_tts_amd.acoustic import AcousticModel
from isp_tts_amd.config import AcousticDims
from oracle import acoustic_oracle as orc, train_oracle as torc, mas_oracle

_ls, _sm = F.log_softmax, F.softmax
F.log_softmax = lambda x, dim, dtype=None: _ls(x, dim=dim, dtype=x.dtype if x.dtype == torch.float64 else dtype)
F.softmax = lambda x, dim, dtype=None: _sm(x, dim=dim, dtype=x.dtype if x.dtype == torch.float64 else dtype)
state_dict = synth.make_state_dict()
inp = synth.make_inputs(2, 40, 150, variable=True, seed=35)


def reference(dtype):
    text, text_len, mel, mel_len, pitch, energy = (inp[k] for k in ("text", "text_len", "mel", "mel_len", "pitch", "energy"))
    cast = lambda v: v.to(dtype) if v.is_floating_point() else v
    mel, pitch, energy = cast(mel), cast(pitch), cast(energy)
    sd = {k: (cast(v).clone().requires_grad_() if v.is_floating_point() and not k.endswith("freq_scale") else cast(v).clone())
          for k, v in state_dict.items()}
    emb = F.embedding(text, sd["text_embedding.weight"], padding_idx=0)
    enc_mask = torch.arange(text.shape[1])[None, :] < text_len[:, None]
    m3 = enc_mask[..., None]
    enc_out = orc.transformer(sd, "encoder", emb, enc_mask)
    soft, logits = orc.conv_attention(sd, mel, enc_out.detach().transpose(1, 2), mel_len, text_len)
    soft.retain_grad(); logits.retain_grad()
    soft_all = soft
    soft_avg, soft_lr, soft_kl = soft * 1, soft * 1, soft * 1
    for t in (soft_avg, soft_lr, soft_kl):
        t.retain_grad()
    hard = torch.from_numpy(mas_oracle.b_mas(logits.detach().float().numpy(), text_len.numpy(), mel_len.numpy())).to(dtype)
    dur = hard.sum(dim=1)
    pt = orc.soft_average(pitch[:, None], soft_avg).transpose(1, 2) * m3
    et = orc.soft_average(energy[:, None], soft_avg).transpose(1, 2) * m3
    targets = torch.cat([torch.log1p(dur)[..., None], pt, et], dim=-1)
    _, flow_loss = orc.predictor_forward(sd, enc_out, targets.detach(), enc_mask, cast(inp["flow_x0"]), cast(inp["flow_t"]))
    x = enc_out + orc.embedding_module(sd, torch.cat([pt, et], dim=-1), enc_mask)
    dec_in = soft_lr @ x
    dec_mask = torch.arange(mel.shape[2])[None, :] < mel_len[:, None]
    dec = orc.transformer(sd, "decoder", dec_in, dec_mask)
    mel_ref = F.linear(dec, sd["to_mel.weight"], sd["to_mel.bias"]).transpose(1, 2) * dec_mask[:, None]
    total = (torc.mel_loss(mel_ref, mel, mel_len) + flow_loss + torc.attention_ctc_loss(logits, text_len, mel_len)
             + torc.attention_binarization_loss(soft_kl, hard))
    total.backward()
    soft.parts = {'avg': soft_avg.grad, 'lr': soft_lr.grad, 'kl': soft_kl.grad}
    return sd, soft, logits, hard


sd32, s32, l32, h32 = reference(torch.float32)
sd64, s64, l64, h64 = reference(torch.float64)
print("hard alignments equal fp32/fp64:", bool(torch.equal(h32.double(), h64)))
rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max())
print("oracle fp32 vs fp64: d soft", rel(s32.grad, s64.grad), " d logits", rel(l32.grad, l64.grad))
names = [k for k in sd64 if k.startswith("aligner.") and sd64[k].is_floating_point() and sd64[k].grad is not None]
for k in names:
    print(f"  oracle fp32 vs fp64 {k:48s} {rel(sd32[k].grad, sd64[k].grad):.2e}")

model = AcousticModel.init(AcousticDims().model_config())
model.load_state_dict(state_dict, strict=True)
model = model.to("cuda").eval()
d = {k: v.to("cuda") for k, v in inp.items()}
grabbed = {}
from isp_tts_amd.train import aligner as tal
orig = tal.ConvAttentionFunction.backward
def spy(ctx, d_soft, d_logits):
    grabbed["soft"], grabbed["logits"] = d_soft.clone(), d_logits.clone()
    return orig(ctx, d_soft, d_logits)
tal.ConvAttentionFunction.backward = staticmethod(spy)
o_avg = tal.SoftAverageFunction.backward
def spy_avg(ctx, d_feats):
    r = o_avg(ctx, d_feats)
    grabbed["avg"], grabbed["d_feats"] = r[0].clone(), d_feats.clone()
    return r
tal.SoftAverageFunction.backward = staticmethod(spy_avg)
o_lr = train.LengthRegulateFunction.backward
def spy_lr(ctx, d_out, a, b):
    r = o_lr(ctx, d_out, a, b)
    grabbed["lr"] = r[1].clone()
    return r
train.LengthRegulateFunction.backward = staticmethod(spy_lr)
mel_out, total, losses = train.acoustic_train_forward(model, d["text"], d["text_len"], d["mel"], d["mel_len"], d["pitch"],
                                                      d["energy"], flow_noise=d["flow_x0"], flow_time=d["flow_t"], train_aligner=True)
total.backward()
print("product vs fp64: d soft", rel(grabbed["soft"], s64.grad), " d logits", rel(grabbed["logits"], l64.grad))
print("parts vs fp64: avg", rel(grabbed["avg"], s64.parts["avg"]), " lr", rel(grabbed["lr"], s64.parts["lr"]), " kl",
      rel(grabbed["soft"] - grabbed["avg"] - grabbed["lr"], s64.parts["kl"]))
print("scales: avg", float(s64.parts["avg"].abs().max()), " lr", float(s64.parts["lr"].abs().max()), " kl", float(s64.parts["kl"].abs().max()))
for k in names:
    p = dict(model.named_parameters())[k]
    print(f"  product vs fp64 {k:48s} {rel(p.grad, sd64[k].grad):.2e}   scale {float(sd64[k].grad.abs().max()):.2e}")
