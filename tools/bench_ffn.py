#!/usr/bin/env python3
"""Times the fused feed-forward kernels at the decoder's shape (32,768 x 384 x 1536): four-wave vs eight-wave, interleaved
rounds in one process (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import build, runtime, synth
if os.environ.get("ISPK_FFN2_ABLATE") or os.environ.get("EXP"):   # ablations live in the experiments build only
    runtime.LIB_PATH = build.LIB_EXP
if os.environ.get("BENCH_LIB"):      # A/B: time another build of the library (path relative to the package)
    runtime.LIB_PATH = os.path.join(os.path.dirname(runtime.__file__), os.environ["BENCH_LIB"])
R, D, Fi = int(os.environ.get("R", 32768)), 384, 1536
dev = "cuda"
x = synth._normal("b/ffn/x", (R, D), 1.5, 0.4).to(dev)
w1 = synth._normal("b/ffn/w1", (Fi, D), D ** -0.5).to(dev).to(torch.bfloat16)
w2 = synth._normal("b/ffn/w2", (D, Fi), Fi ** -0.5).to(dev).to(torch.bfloat16)
g, b = synth._normal("b/ffn/g", (D,), 0.1, 1.0).to(dev), synth._normal("b/ffn/b", (D,), 0.1).to(dev)
mask = (torch.arange(R, device=dev) % 7 != 3)
w2p, w2c = runtime.ffn_pack_w2(w2), runtime.ffn_chunk_w2(w2)
fl = runtime.EP_MASK_OUT
def ablated(code):
    def run():
        os.environ["ISPK_FFN2_ABLATE"] = code
        runtime.ffn_prenorm2(x, g, b, w1, w2c, mask=mask, flags=fl, want_stats=True)
        os.environ.pop("ISPK_FFN2_ABLATE")
    return run


variants = {
    "four-wave (ispk_ffn_bf16_prenorm)": lambda: runtime.ffn_prenorm(x, g, b, w1, w2p, mask=mask, flags=fl, want_stats=True),
    "eight-wave (ispk_ffn_bf16_prenorm2)": lambda: runtime.ffn_prenorm2(x, g, b, w1, w2c, mask=mask, flags=fl, want_stats=True),
}
if runtime.LIB_PATH == build.LIB_EXP:
    variants["eight-wave, no weight DMA after group 1 (compute only)"] = ablated("1")
    variants["eight-wave, DMA + barriers only (no products)"] = ablated("2")
    variants["eight-wave, DMA issued one by one inside the matrix stages"] = ablated("5")
    variants["eight-wave without setprio on the matrix stages"] = ablated("6")
    variants["eight-wave + tanh-form GELU"] = ablated("7")
    variants["eight-wave + DMA issued by half 0 only"] = ablated("8")
    variants["eight-wave + all three"] = ablated("9")
    variants["eight-wave + DMA by half 0 (12)"] = ablated("12")
    variants["eight-wave without the finish stage (no GELU / exchange)"] = ablated("10")
    variants["eight-wave, matrix stages without operand reads"] = ablated("11")
for f in variants.values():
    for _ in range(3):
        f()
torch.cuda.synchronize()
res = {k: [] for k in variants}
for rnd in range(5):
    for k, f in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
flops = 4.0 * R * D * Fi
for k, v in res.items():
    v = sorted(v)
    print(f"{k:40s} median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f} us   {flops / v[len(v)//2] / 1e6:7.1f} TF/s  ({flops / v[len(v)//2] / 1e6 / 2500:.3f} of 2.5 PF)")

if runtime.LIB_PATH == build.LIB_EXP:
    # in-kernel stamps (s_memtime): where a wave's cycles go
    nwg = (R + 127) // 128
    dbg = torch.zeros((nwg * 8, 8), dtype=torch.int64, device=dev)
    os.environ["ISPK_FFN2_ABLATE"] = "3"
    os.environ["ISPK_FFN2_STAMP"] = hex(dbg.data_ptr())
    for _ in range(3):
        runtime.ffn_prenorm2(x, g, b, w1, w2c, mask=mask, flags=fl, want_stats=True)
    torch.cuda.synchronize()
    t = dbg.cpu().double().view(nwg, 8, 8)
    names = ["prologue", "barrier wait", "DMA issue", "finish(+prefetch1)", "product1", "product2(+prefetch2)", "epilogue", "total"]
    for half in (0, 1):
        m = t[:, 4 * half:4 * half + 4].reshape(-1, 8).median(0).values
        print(f"half {half}: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, m.tolist())))

    # determinism of each build
    os.environ.pop("ISPK_FFN2_STAMP", None)
    for code in ("0", "5", "8", "9"):
        os.environ["ISPK_FFN2_ABLATE"] = code
        a = runtime.ffn_prenorm2(x, g, b, w1, w2c, mask=mask, flags=fl)
        bb = runtime.ffn_prenorm2(x, g, b, w1, w2c, mask=mask, flags=fl)
        torch.cuda.synchronize()
        print(f"ablate {code}: two runs equal = {bool(torch.equal(a, bb))}, max diff {(a - bb).abs().max().item():.3e}")
