import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import cases, oracle_lib as O, rlshaders_amd as R
from gpu_util import ggx_oracle
ctx = R.Context(0)
n = 1 << 26
wo, N, T = R.gen_frame(ctx, 1234, 0, n)
u = lambda stream, lo=0.0, hi=1.0: R.gen_uniform(ctx, 1234, 0, n, stream, lo, hi)
Ks = torch.stack([u(8 + j) for j in range(3)])
rough, ior, aniso = u(5, 0.05, 1.0), u(6, 1.05, 2.55), R.gen_aniso(ctx, 1234, 0, n)
xi = [u(11 + j) for j in range(4)]
g = R.GgxSampler(ctx, wo, N, T, specColor=Ks, ior=ior, roughness=rough, anisotropic=aniso)
wi, f, pdf, F, wt, w = g.reflectRefract(*xi)
fin = torch.isfinite(pdf) & torch.isfinite(f).all(dim=0) & torch.isfinite(wi).all(dim=0)
print('nonfinite count', int((~fin).sum()))
neg = ((f < 0).any(dim=0)) & fin
print('neg count', int(neg.sum()))
idx = torch.nonzero(neg | ~fin).flatten()[:12]
sub = lambda t: t[..., idx].contiguous().cpu().numpy()
c = dict(wo=sub(wo), N=sub(N), T=sub(T), KsColor=sub(Ks), roughness=sub(rough), ior=sub(ior), anisotropic=sub(aniso))
ref = ggx_oracle(O, c, nthreads=1).reflect_refract(*[sub(t) for t in xi])
np.set_printoptions(precision=9, linewidth=200)
print('idx', idx.cpu().numpy())
print('gpu f', sub(f)); print('ref f', ref[1])
print('gpu wi', sub(wi)); print('ref wi', ref[0])
print('gpu pdf', sub(pdf)); print('ref pdf', ref[2])
print('rough', c['roughness'], 'aniso', c['anisotropic'], 'xi', sub(xi[0]), sub(xi[1]))
print('wo.N', (c['wo']*c['N']).sum(0), 'wi.N gpu', (sub(wi)*c['N']).sum(0))
