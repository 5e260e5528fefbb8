"""Time every GGX entry point on 2^26 synthetic points (HIP events on the launch stream)."""
import sys
sys.path.insert(0, ".")
import torch
import rlshaders_amd as R

n = 1 << 26
ctx = R.Context(0)
wo, N, T = R.gen_frame(ctx, 1234, 0, n)
u = lambda s, lo=0.0, hi=1.0: R.gen_uniform(ctx, 1234, 0, n, s, lo, hi)
Ks = torch.stack([u(8 + j) for j in range(3)])
g = R.GgxSampler(ctx, wo, N, T, specColor=Ks, ior=u(6, 1.05, 2.55), roughness=u(5, 0.05, 1.0), anisotropic=R.gen_aniso(ctx, 1234, 0, n))
xi = [u(11 + j) for j in range(4)]
wi, F = g.evalSample(xi[0], xi[1])
ops = {
    "sample": lambda: g.evalSample(xi[0], xi[1], out_wi=wi, out_fresnel=F),
    "eval": lambda: g.evalBrdf(wi),
    "pdf": lambda: g.evalPdf(wi),
    "fused": lambda: g.sampleEvalPdf(xi[0], xi[1]),
    "refract": lambda: g.refractSample(xi[2], xi[3]),
    "reflect_refract": lambda: g.reflectRefract(*xi),
    "microfacet_vndf": lambda: g.microfacet(xi[0], xi[1], R.RLS_KERNEL_VNDF),
    "microfacet_ndf": lambda: g.microfacet(xi[0], xi[1], R.RLS_KERNEL_NDF),
}
for name, fn in ops.items():
    fn(); torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(5):
        fn()
    ctx.timer_stop()
    print(f"{name:18s} {ctx.timer_elapsed_ms() / 5:.3f} ms")
