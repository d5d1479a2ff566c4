import torch, time
torch.set_num_threads(16)
src = torch.randn(64, 80, 512)
dst = torch.empty_like(src).pin_memory()
for thr in (1, 4, 16):
    torch.set_num_threads(thr)
    for _ in range(3): dst.copy_(src)
    t0 = time.perf_counter()
    for _ in range(20): dst.copy_(src)
    dt = (time.perf_counter() - t0) / 20
    print(f"threads {thr}: {dt*1e3:.3f} ms  {src.numel()*4/dt/1e9:.1f} GB/s")
import numpy as np
a = src.numpy(); b = dst.numpy()
t0 = time.perf_counter()
for _ in range(20): np.copyto(b, a)
dt = (time.perf_counter() - t0) / 20
print(f"numpy copyto: {dt*1e3:.3f} ms")
d = torch.empty(64, 80, 512, device="cuda")
for _ in range(3):
    d.copy_(dst, non_blocking=True); d.copy_(src)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): d.copy_(dst, non_blocking=True)
torch.cuda.synchronize()
print(f"H2D pinned: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
t0 = time.perf_counter()
for _ in range(20): d.copy_(src)
torch.cuda.synchronize()
print(f"H2D pageable: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
