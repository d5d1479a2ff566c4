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


HAVE3 = runtime.LIB_PATH == build.LIB_EXP      # csrc/ffn3.hip (one wave per SIMD) exists in the experiments build only


def ffn_prenorm3(x, g, b, w1, w2p, mask=None, flags=0, want_stats=False):
    """ispk_ffn_bf16_prenorm3 (experiments build; arguments as ispk_ffn_bf16_prenorm2 with ispk_ffn_pack_w2_bf16's W2 image)."""
    import ctypes
    fn = runtime.lib().ispk_ffn_bf16_prenorm3
    P, I64, I32, U32, F32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint32, ctypes.c_float
    fn.argtypes = [P, I64, P, P, F32, P, P, P, P, I64, I32, I32, I32, U32, P, F32, P]
    fn.restype = I32
    out = torch.empty_like(x)
    stats = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device) if want_stats else None
    rc = fn(x.data_ptr(), x.stride(0), g.data_ptr(), b.data_ptr(), 1e-5, w1.data_ptr(), w2p.data_ptr(),
            None if mask is None else mask.data_ptr(), out.data_ptr(), out.stride(0), x.shape[0], x.shape[1], w1.shape[0], flags,
            None if stats is None else stats.data_ptr(), 1e-5, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, runtime.lib().ispk_last_error_string()
    return (out, stats) if want_stats else out


def ablated3(code):
    def run():
        os.environ["ISPK_FFN3_ABLATE"] = code
        ffn_prenorm3(x, g, b, w1, w2p, mask=mask, flags=fl, want_stats=True)
        os.environ.pop("ISPK_FFN3_ABLATE")
    return run


variants = {
    "four-wave (ispk_ffn_bf16_prenorm)": lambda: runtime.ffn_prenorm(x, g, b, w1, w2p, mask=mask, flags=fl, want_stats=True),
    "eight-wave (ispk_ffn_bf16_prenorm2)": lambda: runtime.ffn_prenorm2(x, g, b, w1, w2c, mask=mask, flags=fl, want_stats=True),
}
o_att = synth._normal("b/ffn/o", (R, D), 1.0).to(dev).to(torch.bfloat16)
wo = synth._normal("b/ffn/wo", (D, D), D ** -0.5).to(dev).to(torch.bfloat16)
woc = runtime.ffn_chunk_w2(wo)


def two_launches():
    x1 = runtime.gemm(o_att, wo, resid=x, mask=mask, flags=runtime.EP_MASK_ACC, out_dtype=torch.float32)
    return runtime.ffn_prenorm2(x1, g, b, w1, w2c, mask=mask, flags=fl, want_stats=True)


variants["to_out GEMM (+ residual), then eight-wave: the two launches"] = two_launches
variants["projection + feed-forward, one kernel (ispk_attn_out_ffn_bf16)"] = lambda: runtime.attn_out_ffn(x, o_att, woc, g, b, w1, w2c, mask=mask, want_stats=True)
wq = synth._normal("b/ffn/wq", (512, D), D ** -0.5).to(dev).to(torch.bfloat16)
wqc = runtime.chunk_k16(wq)


def three_launches():
    y, _ = two_launches()
    return runtime.gemm_lnin(y, None, g, b, wq)


variants["... and the next layer's q/kv GEMM (own statistics): three launches"] = three_launches
variants["projection + feed-forward + next q/kv, one kernel (ispk_attn_out_ffn_qkv_bf16)"] = lambda: runtime.attn_out_ffn(x, o_att, woc, g, b, w1, w2c, mask=mask, next_qkv=(g, b, 1e-5, wqc))
if HAVE3:
    variants["single wave per SIMD (ispk_ffn_bf16_prenorm3, experiments build)"] = lambda: ffn_prenorm3(x, g, b, w1, w2p, mask=mask, flags=fl, want_stats=True)
    variants["ffn3, no weight DMA in the main loop (compute only)"] = ablated3("1")
    variants["ffn3, no GELU work in the gaps"] = ablated3("2")
if HAVE3:
    variants["eight-wave, TWO-SLOT schedule, DMA 8/4 (30)"] = ablated("30")
    variants["eight-wave, two-slot, DMA even (31)"] = ablated("31")
    variants["eight-wave, DMA between the GELU levels of the finish stage (40)"] = ablated("40")
    variants["eight-wave, two-slot, NO weight DMA after group 1 (35)"] = ablated("35")
    variants["eight-wave, default schedule, W1 half of the DMA bytes only (36)"] = ablated("36")
    variants["eight-wave, no weight DMA after group 1 (compute only)"] = ablated("1")
    variants["eight-wave, two-slot, DMA 9/3 (32)"] = ablated("32")
    variants["eight-wave, two-slot, DMA 10/2 (33)"] = ablated("33")
if runtime.LIB_PATH == build.LIB_EXP and os.environ.get("FFN2_ABL"):
    variants["eight-wave, no weight DMA after group 1 (compute only)"] = ablated("1")
    variants["eight-wave, DMA + barriers only (no products)"] = ablated("2")
    variants["eight-wave, DMA issued one by one inside the matrix stages"] = ablated("5")
    variants["eight-wave without setprio on the matrix stages"] = ablated("6")
    variants["eight-wave + tanh-form GELU"] = ablated("7")
    variants["eight-wave + DMA issued by half 0 only"] = ablated("8")
    variants["eight-wave + all three"] = ablated("9")
    variants["eight-wave + DMA by half 0 (12)"] = ablated("12")
    variants["eight-wave, DMA split 8/4 between the halves (13)"] = ablated("13")
    variants["eight-wave, DMA split 9/3 (14)"] = ablated("14")
    variants["eight-wave, DMA split 10/2 (15)"] = ablated("15")
    variants["eight-wave, static prio 1 for half 1, no per-stage setprio (16)"] = ablated("16")
    variants["eight-wave, static prio + DMA split 8/4 (17)"] = ablated("17")
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
    nwg = (R + 127) // 128
    dbg3 = torch.zeros((nwg * 4, 5), dtype=torch.int64, device=dev)
    os.environ["ISPK_FFN3_ABLATE"] = "3"
    os.environ["ISPK_FFN3_STAMP"] = hex(dbg3.data_ptr())
    for _ in range(3):
        ffn_prenorm3(x, g, b, w1, w2p, mask=mask, flags=fl, want_stats=True)
    torch.cuda.synchronize()
    os.environ.pop("ISPK_FFN3_ABLATE"); os.environ.pop("ISPK_FFN3_STAMP")
    m = dbg3.cpu().double().median(0).values.tolist()
    print("ffn3 stamps (cycles per wave, median): " + ", ".join(f"{n} {v:.0f}" for n, v in zip(["prologue", "fill", "main loop", "epilogue", "total"], m)))
    ref0 = runtime.ffn_prenorm2(x, g, b, w1, w2c, mask=mask, flags=fl, want_stats=True)
    for code in ("30", "31", "32", "33", "40"):
        os.environ["ISPK_FFN2_ABLATE"] = code
        got = runtime.ffn_prenorm2(x, g, b, w1, w2c, mask=mask, flags=fl, want_stats=True)
        again = runtime.ffn_prenorm2(x, g, b, w1, w2c, mask=mask, flags=fl, want_stats=True)
        os.environ.pop("ISPK_FFN2_ABLATE")
        torch.cuda.synchronize()
        print(f"two-slot ({code}) vs the default schedule: out equal = {bool(torch.equal(got[0], ref0[0]))}, stats equal = {bool(torch.equal(got[1], ref0[1]))}, "
              f"max diff {(got[0] - ref0[0]).abs().max().item():.3e}; two runs equal = {bool(torch.equal(got[0], again[0]))}")
    dbg = torch.zeros((nwg * 8, 8), dtype=torch.int64, device=dev)
    for code in ("3", "34"):
        os.environ["ISPK_FFN2_ABLATE"] = code
        os.environ["ISPK_FFN2_STAMP"] = hex(dbg.data_ptr())
        for _ in range(3):
            runtime.ffn_prenorm2(x, g, b, w1, w2c, mask=mask, flags=fl, want_stats=True)
        torch.cuda.synchronize()
        t = dbg.cpu().double().view(nwg, 8, 8)
        names = ["prologue", "barrier wait", "DMA issue", "finish(+prefetch1)", "product1", "product2(+prefetch2)", "epilogue", "total"]
        for half in (0, 1):
            m = t[:, 4 * half:4 * half + 4].reshape(-1, 8).median(0).values
            print(f"stamps {'default' if code == '3' else 'two-slot'} half {half}: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, m.tolist())))
    os.environ.pop("ISPK_FFN2_ABLATE"); os.environ.pop("ISPK_FFN2_STAMP")
    a3 = ffn_prenorm3(x, g, b, w1, w2p, mask=mask, flags=fl)
    a2 = runtime.ffn_prenorm2(x, g, b, w1, w2c, mask=mask, flags=fl)
    print(f"ffn3 vs ffn2: max diff {(a3 - a2).abs().max().item():.3e}, rms {(a3 - a2).pow(2).mean().sqrt().item():.3e}; two runs equal = {bool(torch.equal(a3, ffn_prenorm3(x, g, b, w1, w2p, mask=mask, flags=fl)))}")
if runtime.LIB_PATH == build.LIB_EXP and os.environ.get("FFN2_ABL"):
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
    for code in ("0", "5", "8", "9", "13", "14", "15"):
        os.environ["ISPK_FFN2_ABLATE"] = code
        a = runtime.ffn_prenorm2(x, g, b, w1, w2c, mask=mask, flags=fl)
        bb = runtime.ffn_prenorm2(x, g, b, w1, w2c, mask=mask, flags=fl)
        torch.cuda.synchronize()
        print(f"ablate {code}: two runs equal = {bool(torch.equal(a, bb))}, max diff {(a - bb).abs().max().item():.3e}")
