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
ln = torch.linalg.vector_norm(wi.double(), dim=0)
bad = fin & ((ln - 1).abs() > 1e-5)
print('nonfinite', int((~fin).sum()), 'nonunit', int(bad.sum()))
idx = torch.nonzero(bad | ~fin).flatten()[:8]
sub = lambda t: t[..., idx].contiguous().cpu().numpy()
c = dict(wo=sub(wo), N=sub(N), T=sub(T), KsColor=sub(Ks), roughness=sub(rough), ior=sub(ior), anisotropic=sub(aniso))
ref = ggx_oracle(O, c, nthreads=1).reflect_refract(*[sub(t) for t in xi])
np.set_printoptions(precision=9, linewidth=200)
print('idx', idx.cpu().numpy())
print('gpu wi', sub(wi)); print('ref wi', ref[0])
print('gpu f', sub(f)); print('ref f', ref[1])
print('rough', c['roughness'], 'aniso', c['anisotropic'], 'xi', sub(xi[0]), sub(xi[1]))
print('wo.N', (c['wo']*c['N']).sum(0))
# overall bit-exactness on a strided subset
idx = torch.arange(0, n, 257, device='cuda')
c = dict(wo=sub(wo), N=sub(N), T=sub(T), KsColor=sub(Ks), roughness=sub(rough), ior=sub(ior), anisotropic=sub(aniso))
ref = ggx_oracle(O, c, nthreads=8).reflect_refract(*[sub(t) for t in xi])
for nm, a, b in zip(("wi", "f", "pdf", "fresnel", "wt", "weight"), (wi, f, pdf, F, wt, w), ref):
    a = sub(a)
    neq = (a.view(np.uint32) != b.view(np.uint32))
    neq = neq.any(axis=0) if neq.ndim == 2 else neq
    print(nm, 'points', a.shape[-1], 'bit-mismatches', int(neq.sum()), 'max rel', float(cases.rel_err(a, b).max()))
