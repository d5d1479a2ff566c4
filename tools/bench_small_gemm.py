#!/usr/bin/env python3
"""Long-K / skinny bf16 GEMMs at SMALL row counts (the shapes a rank sees under strong scaling: 8 - 16 utterances per GPU): the
row-block kernel (gemm_bf16_wide_kernel) against the generic tile kernel (ISPK_WIDE_MIN_M=2048 restores the old rule).
Experiments build (GPU box):  EXP=1 python tools/bench_small_gemm.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import build, runtime, synth
runtime.LIB_PATH = build.LIB_EXP
dev = "cuda"
shapes = [(800, 384, 1536), (800, 256, 1024), (832, 768, 1920), (832, 128, 768), (1600, 384, 1536), (400, 384, 1536), (100, 384, 1536),
          (4160, 128, 768), (4160, 768, 1920)]
for M, N, K in shapes:
    a = synth._normal(f"b/sg/a{M}{K}", (M, K)).to(dev).to(torch.bfloat16)
    w = synth._normal(f"b/sg/w{N}{K}", (N, K), K ** -0.5).to(dev).to(torch.bfloat16)
    r = synth._normal(f"b/sg/r{M}{N}", (M, N)).to(dev)
    mask = (torch.arange(M, device=dev) % 7 != 3)
    res = {}
    outs = {}
    out_direct = torch.empty((M, N), device=dev)

    def direct():      # ispk_gemm_bf16 itself (no split-K)
        rc = runtime.lib().ispk_gemm_bf16(a.data_ptr(), K, w.data_ptr(), K, out_direct.data_ptr(), N, None, r.data_ptr(), N, mask.data_ptr(),
                                         M, N, K, runtime.EP_MASK_OUT, 0, 0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return out_direct
    ks = runtime.lib().ispk_gemm_bf16_splitk_plan(M, N, K, runtime.EP_MASK_OUT)
    for name, env in ((f"split-K x{ks} (runtime.gemm)", None), ("row-block kernel, unsplit", "direct"), ("generic tile kernel (old rule)", "2048")):
        if env == "2048": os.environ["ISPK_WIDE_MIN_M"] = env
        else: os.environ.pop("ISPK_WIDE_MIN_M", None)
        f = direct if env else (lambda: runtime.gemm(a, w, resid=r, mask=mask, flags=runtime.EP_MASK_OUT, out_dtype=torch.float32))
        for _ in range(3): outs[name] = f().clone()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()          # (eager launches of kernels this small are host-bound: time a graph of 20)
        with torch.cuda.graph(g):
            for _ in range(20): f()
        g.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        res[name] = sorted(ts)[2]
        del g
    os.environ.pop("ISPK_WIDE_MIN_M", None)
    o1, o2 = list(outs.values())[0], list(outs.values())[1].clone()
    ref = (a.double() @ w.double().t() + r.double()) * mask[:, None]
    print(f"{M:5d} x {N:4d} x {K:4d}: " + ", ".join(f"{k} {v:6.1f} us" for k, v in res.items()) +
          f"; max diff between them {(o1 - o2).abs().max().item():.2e}, vs float64 {(o1.double() - ref).abs().max().item():.2e}")
