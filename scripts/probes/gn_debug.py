import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from simpleaicv_pytorch_training_examples_amd import _lib
L = _lib.lib()
torch.manual_seed(0)
N, C, H, W, G = 2, 256, 12, 10, 32
cpg = C // G
gamma = torch.rand(C) + 0.5; beta = torch.randn(C) * 0.3
x = torch.randn(N, C, H, W) * 1.7 + 0.4; dy = torch.randn(N, C, H, W)
X = x.permute(0, 2, 3, 1).reshape(N, H * W, C).contiguous(); DY = dy.permute(0, 2, 3, 1).reshape(N, H * W, C).contiguous()
S = X.sum(1); Q = (X * X).sum(1); cnt = H * W * cpg
mean = S.view(N, G, cpg).sum(2) / cnt; var = (Q.view(N, G, cpg).sum(2) / cnt - mean * mean).clamp(min=0); rstd = (var + 1e-5).rsqrt()
meanc = mean.repeat_interleave(cpg, 1); rstdc = rstd.repeat_interleave(cpg, 1)
a = rstdc * gamma; b = beta - meanc * a
gate = (X * a[:, None, :] + b[:, None, :]) > 0
DYg = torch.where(gate, DY, torch.zeros_like(DY))
A = DYg.sum(1); B = (DYg * X).sum(1)
s1 = (gamma * A).view(N, G, cpg).sum(2).repeat_interleave(cpg, 1); s2 = (gamma * rstdc * (B - meanc * A)).view(N, G, cpg).sum(2).repeat_interleave(cpg, 1)
p = rstdc * gamma; q = -rstdc * rstdc * s2 / cnt; r = -rstdc * s1 / cnt - q * meanc
DX = p[:, None, :] * DYg + q[:, None, :] * X + r[:, None, :]
d = lambda t: t.cuda().contiguous()
xd, dyd, gd, bd = d(X), d(DY), d(gamma), d(beta)
y = torch.empty_like(xd); mr = torch.empty(2, N, G, device='cuda'); ab = torch.empty(2, N, C, device='cuda'); ws = torch.empty(5 * N * C, device='cuda')
_lib.check(L.saicv_groupnorm_fwd(_lib.F32, xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), y.data_ptr(), mr.data_ptr(), ab.data_ptr(), ws.data_ptr(), N, H * W, C, G, 1e-5, 1, None), 'f')
torch.cuda.synchronize()
print('ab err', float((ab[0].cpu() - a).abs().max()), float((ab[1].cpu() - b).abs().max()), 'mr err', float((mr[0].cpu() - mean).abs().max()), float((mr[1].cpu() - rstd).abs().max()))
dx = torch.empty_like(xd); dg = torch.zeros(C, device='cuda'); db = torch.zeros(C, device='cuda'); ws2 = torch.empty(5 * N * C, device='cuda')
_lib.check(L.saicv_groupnorm_bwd(_lib.F32, dyd.data_ptr(), xd.data_ptr(), gd.data_ptr(), mr.data_ptr(), ab.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws2.data_ptr(), N, H * W, C, G, 1, None), 'b')
torch.cuda.synchronize()
w = ws2.cpu().view(5, N, C)
print('A err', float((w[0] - A).abs().max()), 'B err', float((w[1] - B).abs().max()), 'A scale', float(A.abs().max()))
print('p err', float((w[2] - p).abs().max()), 'q err', float((w[3] - q).abs().max()), 'r err', float((w[4] - r).abs().max()))
print('dx err', float((dx.cpu() - DX).abs().max()), 'dx if all gated', float((dx.cpu() - (q[:, None, :] * X + r[:, None, :])).abs().max()))
print('ab after err', float((ab[0].cpu() - a).abs().max()), float((ab[1].cpu() - b).abs().max()))
