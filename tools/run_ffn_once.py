#!/usr/bin/env python3
"""Launches the fused feed-forward kernels a few times at the decoder's shape (for rocprofv3 --pmc passes): the eight-wave kernel
alone (ffn2<0>), with the attention block's output projection as its prologue (<50>), with the next layer's q/kv projection as
its epilogue too (<51>), and the round-1 four-wave kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, synth
R, D, Fi = 32768, 384, 1536
dev = "cuda"
x = synth._normal("b/ffn/x", (R, D), 1.5, 0.4).to(dev)
w1 = synth._normal("b/ffn/w1", (Fi, D), D ** -0.5).to(dev).to(torch.bfloat16)
w2 = synth._normal("b/ffn/w2", (D, Fi), Fi ** -0.5).to(dev).to(torch.bfloat16)
g, b = synth._normal("b/ffn/g", (D,), 0.1, 1.0).to(dev), synth._normal("b/ffn/b", (D,), 0.1).to(dev)
mask = (torch.arange(R, device=dev) % 7 != 3)
w2c, w2p = runtime.ffn_chunk_w2(w2), runtime.ffn_pack_w2(w2)
o_att = synth._normal("b/ffn/o", (R, D), 1.0).to(dev).to(torch.bfloat16)
woc = runtime.ffn_chunk_w2(synth._normal("b/ffn/wo", (D, D), D ** -0.5).to(dev).to(torch.bfloat16))
wqc = runtime.chunk_k16(synth._normal("b/ffn/wq", (512, D), D ** -0.5).to(dev).to(torch.bfloat16))
for _ in range(6):
    runtime.attn_out_ffn(x, o_att, woc, g, b, w1, w2c, mask=mask, want_stats=True)
    runtime.attn_out_ffn(x, o_att, woc, g, b, w1, w2c, mask=mask, next_qkv=(g, b, 1e-5, wqc))
    runtime.ffn_prenorm2(x, g, b, w1, w2c, mask=mask, flags=runtime.EP_MASK_OUT, want_stats=True)
    runtime.ffn_prenorm(x, g, b, w1, w2p, mask=mask, flags=runtime.EP_MASK_OUT, want_stats=True)
torch.cuda.synchronize()
