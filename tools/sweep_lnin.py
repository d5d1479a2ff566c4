"""Times ispk_gemm_bf16_lnin (LayerNorm + q/kv projection, 32,768 x 512 x 384) and the plain panel GEMM for
ISPK_PANEL_NSPLIT = 1, 2, 4 (experiments build).  usage: sweep_lnin.py"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from isp_tts_amd import build, runtime
    runtime.LIB_PATH = build.LIB_EXP
    dev = "cuda"
    M, N, K = 32768, 512, 384
    x = torch.randn(M, K, device=dev)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    g, b = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    stats = torch.stack([x.mean(1), 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)], dim=1).contiguous()
    xb = x.to(torch.bfloat16)
    wo = (torch.randn(384, 384, device=dev) * K ** -0.5).to(torch.bfloat16)
    res = torch.randn(M, 384, device=dev)

    def t(fn, n=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    print(f"NSPLIT={os.environ.get('ISPK_PANEL_NSPLIT', 'default')}: lnin(stats) {t(lambda: runtime.gemm_lnin(x, stats, g, b, w)):.1f} us  "
          f"lnin(self) {t(lambda: runtime.gemm_lnin(x, None, g, b, w)):.1f} us  plain qkv {t(lambda: runtime.gemm(xb, w)):.1f} us  "
          f"out-proj {t(lambda: runtime.gemm(xb, wo, resid=res, out_dtype=torch.float32)):.1f} us")
else:
    for ns in ("", "1", "2", "3", "4", "8"):
        env = dict(os.environ)
        if ns:
            env["ISPK_PANEL_NSPLIT"] = ns
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
