"""CPU study (test infrastructure): how many bf16 terms per fp32 operand does the parity path need?

Runs the oracle's forward with every Linear / attention / convolution product replaced by an emulation of split-bf16 MFMA
products (operands split into bf16 hi / lo [/ lo2] terms, each bf16 x bf16 product exact, fp32 accumulation) and reports mel
L-inf and the number of identical MAS paths against the plain fp32 oracle.  usage: python tools/split_numerics.py [B] [terms]"""
import sys
import types

import torch
import torch.nn.functional as TF

sys.path.insert(0, ".")
from isp_tts_amd import synth  # noqa: E402
from oracle import acoustic_oracle as orc  # noqa: E402


DT = torch.bfloat16
FLUSH = False


def split(x, terms):
    out, r = [], x
    for _ in range(terms):
        h = r.to(DT).float()
        if FLUSH and DT == torch.float16:   # denormal inputs flushed to zero
            h = torch.where(h.abs() < 6.103515625e-05, torch.zeros_like(h), h)
        out.append(h)
        r = r - h
    return out


def make_mm(terms, products):
    def mm(a, bt):
        """a [.., M, K] @ bt [.., K, N] through split products; `products` = list of (i, j) term pairs."""
        sa, sb = split(a, terms), split(bt, terms)
        acc = None
        for i, j in products:
            p = torch.matmul(sa[i], sb[j])
            acc = p if acc is None else acc + p
        return acc
    return mm


def patched(terms, products, exp_fast=False):
    mm = make_mm(terms, products)
    ns = types.SimpleNamespace(**{k: getattr(TF, k) for k in dir(TF) if not k.startswith("__")})

    def linear(x, w, b=None):
        y = mm(x, w.t())
        return y if b is None else y + b

    def sdpa(q, k, v, attn_mask=None):
        s = mm(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5) + attn_mask
        p = torch.softmax(s, dim=-1)
        return mm(p, v)

    def conv1d(x, w, bias, padding=0):
        # [B, C, T] conv as a GEMM over unfolded windows
        k = w.shape[2]
        xp = TF.pad(x, (padding, padding))
        cols = xp.unfold(2, k, 1)                       # [B, C, T, k]
        a = cols.permute(0, 2, 1, 3).reshape(x.shape[0], -1, x.shape[1] * k)
        y = mm(a, w.reshape(w.shape[0], -1).t())
        return y.transpose(1, 2)

    ns.linear, ns.scaled_dot_product_attention, ns.conv1d = linear, sdpa, conv1d
    return ns, mm


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    sd = synth.make_state_dict()
    inp = synth.make_inputs(B, 100, 512, variable=True)
    args = (inp["text"], inp["text_len"], inp["mel"], inp["mel_len"], inp["pitch"], inp["energy"], inp["flow_x0"], inp["flow_t"])
    ref = orc.acoustic_forward(sd, *args)
    sd64 = {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}
    print("mel abs max", ref.mel.abs().max().item(), "rms", ref.mel.pow(2).mean().sqrt().item())
    global DT, FLUSH
    if len(sys.argv) > 2 and sys.argv[2].startswith("f16"):
        DT = torch.float16
        FLUSH = sys.argv[2] == "f16flush"
    cases = {
        "2 terms, 3 products": (2, [(0, 0), (0, 1), (1, 0)]),
        "2 terms, 4 products": (2, [(0, 0), (0, 1), (1, 0), (1, 1)]),
        "3 terms, 6 products": (3, [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]),
        "1 term (bf16)": (1, [(0, 0)]),
    }
    realF = orc.F
    for name, (terms, products) in cases.items():
        ns, mm = patched(terms, products)
        orc.F = ns
        realmm = torch.matmul
        try:
            out = orc.acoustic_forward(sd, *args)
        finally:
            orc.F = realF
        d = (out.mel - ref.mel).abs().max().item()
        dl = (out.aligner.attn_logits - ref.aligner.attn_logits).abs().max().item()
        ds = (out.aligner.attn_soft - ref.aligner.attn_soft).abs().max().item()
        de = (out.enc_out - ref.enc_out).abs().max().item()
        same = int((out.aligner.attn_hard == ref.aligner.attn_hard).flatten(1).all(1).sum())
        print(f"{name:22s}: mel Linf {d:.3e}  enc_out {de:.3e}  attn_logits {dl:.3e}  attn_soft {ds:.3e}  MAS identical {same}/{B}")


if __name__ == "__main__":
    main()
