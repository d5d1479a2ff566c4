"""Debug: soft-average backward and the length regulator's alignment gradient against float64 autograd."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isp_tts_amd import runtime, train
from isp_tts_amd.train import aligner as tal
DEV = "cuda"
g = torch.Generator().manual_seed(1)
B, M, L, D = 3, 150, 40, 384
mel_len, text_len = torch.tensor([150, 77, 120]), torch.tensor([40, 21, 33])
A = torch.softmax(torch.randn(B, M, L, generator=g) * 2, -1)
A = A * (torch.arange(M)[None, :] < mel_len[:, None])[..., None] * (torch.arange(L)[None, :] < text_len[:, None])[:, None, :]
pitch, energy = torch.randn(B, M, generator=g), torch.randn(B, M, generator=g)
gf = torch.randn(B, L, 3, generator=g)
A64 = A.double().requires_grad_()
m3 = (torch.arange(L)[None, :] < text_len[:, None])[..., None]
pt = (pitch.double()[:, None] @ A64 / (A64.sum(1, keepdim=True) + 1e-5)).transpose(1, 2) * m3
et = (energy.double()[:, None] @ A64 / (A64.sum(1, keepdim=True) + 1e-5)).transpose(1, 2) * m3
(torch.cat([torch.zeros_like(pt), pt, et], -1) * gf.double()).sum().backward()
Ag = A.to(DEV).requires_grad_()
f = tal.SoftAverageFunction.apply(Ag, pitch.to(DEV), energy.to(DEV), text_len.to(DEV))
print("soft average fwd err", float((f.cpu()[..., 1:] - torch.cat([pt, et], -1).detach()).abs().max()))
(f * gf.to(DEV)).sum().backward()
print("soft average dA rel err", float((Ag.grad.cpu() - A64.grad).abs().max() / A64.grad.abs().max()))
x = torch.randn(B, L, D, generator=g)
dout = torch.randn(B, M, D, generator=g) * (torch.arange(M)[None, :] < mel_len[:, None])[..., None]
A64 = A.double().requires_grad_(); x64 = x.double().requires_grad_()
(torch.bmm(A64, x64) * dout.double()).sum().backward()
Ag = A.to(DEV).requires_grad_(); xg = x.to(DEV).requires_grad_()
out, _, _ = train.LengthRegulateFunction.apply(xg, Ag, mel_len.view(-1, 1).to(DEV), M)
(out * dout.to(DEV)).sum().backward()
print("LR dA rel err", float((Ag.grad.cpu() - A64.grad).abs().max() / A64.grad.abs().max()), " dx rel err",
      float((xg.grad.cpu() - x64.grad).abs().max() / x64.grad.abs().max()))
