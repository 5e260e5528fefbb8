"""One pass of every closure family over n device-generated shading points, GPU against oracle, counted in
output WORDS that differ at all.  Used by tests/test_gpu_integrate_gen.py (2^22 points) and
tools/parity_soak.py (2^24 points x several seeds, summary committed under profiles/)."""
import os

import numpy as np
import torch

import cases
import oracle_lib as O
import rlshaders_amd as R
from gpu_util import ggx_oracle


def _words(got, ref):
    diff = total = beyond = 0
    worst = 0.0
    for a, b in zip(got, ref):
        bad = a.view(np.uint32) != b.view(np.uint32)
        bad &= ~(np.isnan(a) & np.isnan(b))
        diff += int(bad.sum())
        total += a.size
        if bad.any():
            e = cases.rel_err(a[bad], b[bad])
            worst = max(worst, float(e.max()))
            beyond += int((e > 1e-5).sum())
    return dict(words_differing=diff, words=total, max_rel_err=worst, beyond_1e5=beyond)


def sweep(ctx, n: int, seed: int, spp_n: int = 2, verbose: bool = True, group: str = '1') -> dict:
    th = O.hardware_threads()
    hostf = lambda t: t.contiguous().cpu().numpy()
    wo, N, T = R.gen_frame(ctx, seed, 0, n)
    u = lambda stream, lo=0.0, hi=1.0: R.gen_uniform(ctx, seed, 0, n, stream, lo, hi)
    report = {}

    def tally(name, got, ref):
        report[name] = _words(got, ref)
        if verbose:
            r = report[name]
            print(f"sweep seed {seed} {name}: {r['words_differing']} of {r['words']} words differ, "
                  f"max rel err {r['max_rel_err']:.3g}, beyond 1e-5: {r['beyond_1e5']}", flush=True)

    Ks = torch.stack([u(8 + j) for j in range(3)])
    rough, ior, aniso = u(5, 0.05, 1.0), u(6, 1.05, 2.55), R.gen_aniso(ctx, seed, 0, n)
    xi = [u(11 + j) for j in range(6)]
    hxi = [hostf(t) for t in xi]
    c = dict(wo=hostf(wo), N=hostf(N), T=hostf(T), KsColor=hostf(Ks), roughness=hostf(rough), ior=hostf(ior),
             anisotropic=hostf(aniso))
    # --- rlGgx reflect + refract, n^2-spp integrator, direct lighting
    g = R.GgxSampler(ctx, wo, N, T, specColor=Ks, ior=ior, roughness=rough, anisotropic=aniso)
    og = ggx_oracle(O, c, nthreads=th)
    tally("ggx reflect+refract", [hostf(t) for t in g.reflectRefract(*xi[:4])], og.reflect_refract(*hxi[:4]))
    os.environ["RLS_INTEGRATE_GROUP"] = group        # lanes per point: the sums grow in sample order whatever the width (fold)
    try:
        tally("ggx integrate", [hostf(t) for t in g.integrate(spp_n, seed)], og.integrate(spp_n, seed))
        P = torch.stack([u(40, 0, 4), u(41, 0, 4), u(42, 0, 1)])
        lt = O.make_light(center=(2.0, 2.0, 3.0), radius=1.25, radiance=(3.0, 2.0, 1.0))
        lg = R._capi.SphereLight.from_buffer_copy(bytes(lt))
        kd, kdr, ks = u(43), u(44), u(45)
        tally("ggx direct lighting",
              [hostf(t) for t in g.directLighting(P, lg, spp_n, seed, KdColor=Ks, Kd=kd, diffuseRoughness=kdr, Ks=ks)],
              og.direct_lighting(hostf(P), lt, spp_n, seed, Kd_color=c["KsColor"], Kd=hostf(kd), Kd_roughness=hostf(kdr),
                                 Ks=hostf(ks)))
        # the whole shader_evaluate of rlGgx under two lights
        lt2 = [lt, O.make_light(center=(-3.0, 1.0, 2.5), radius=0.5, radiance=(0.5, 4.0, 2.0), mis_mode=2)]
        lg2 = [R._capi.SphereLight.from_buffer_copy(bytes(l)) for l in lt2]
        ktc, kt = torch.stack([u(58), u(59), u(60)]), u(61)
        keys = ("direct_diffuse", "direct_specular", "refraction", "indirect_diffuse", "indirect_specular", "out")
        gs = g.shade(P, lg2, spp_n, seed, KdColor=Ks, Kd=kd, diffuseRoughness=kdr, Ks=ks, KtColor=ktc, Kt=kt, env=(1.0, 0.9, 0.8))
        rs = og.shade(hostf(P), lt2, spp_n, seed, Kd_color=c["KsColor"], Kd=hostf(kd), Kd_roughness=hostf(kdr), Ks=hostf(ks),
                      Kt_color=hostf(ktc), Kt=hostf(kt), env=(1.0, 0.9, 0.8))
        tally("ggx shader_evaluate", [hostf(gs[k]) for k in keys], [rs[k] for k in keys])
        tally("ggx integrateRefract", [hostf(t) for t in g.integrateRefract(spp_n, seed, want_tir=True)],
              og.integrate_refract(spp_n, seed))
        # --- rlSss integrateScatter on the unit sphere
        dsmall = torch.stack([u(32 + j, 0.02, 0.3) for j in range(3)])
        ss = R.SssSampler(ctx, N, T, Ks, dsmall)
        so = O.make_scene("sphere", sphere_radius=0.35, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
        sg = R._capi.SssScene.from_buffer_copy(bytes(so))
        Psph = (N * 0.35).contiguous()
        osss = O.Sss(n, hostf(dsmall), c["KsColor"], N=c["N"], T=c["T"], nthreads=th)
        tally("sss integrateScatter", [hostf(t) for t in ss.integrateScatter(Psph, sg, spp_n, seed, want_depth=True)],
              O.integrate_scatter(osss, hostf(Psph), so, spp_n, seed))
    finally:
        del os.environ["RLS_INTEGRATE_GROUP"]
    # --- rlDisney both lobes
    sc = {k: u(32 + j) for j, k in enumerate(R._capi.DISNEY_SCALARS)}
    d = R.DisneySampler(ctx, wo, N, T, base_color=Ks, **sc)
    od = O.Disney(c["wo"], c["N"], c["T"], base_color=c["KsColor"], nthreads=th, **{k: hostf(v) for k, v in sc.items()})
    od.two_sums_default = True                  # the device's summation order (oracle/rls_oracle.c, ggx_light_loop)
    for lobe, nm in ((R.RLS_RAY_DIFFUSE, "diffuse"), (R.RLS_RAY_GLOSSY, "glossy")):
        d.setSampleType(lobe)
        tally(f"disney {nm}", [hostf(t) for t in d.sampleEvalPdf(xi[0], xi[1])], od.sample_eval_pdf(lobe, hxi[0], hxi[1]))
    os.environ["RLS_INTEGRATE_GROUP"] = group
    try:                                              # BASELINE config 3's loop: spp_n^2 samples per lobe, both lobes
        keys = ("diffuse_sum", "diffuse_count", "specular_sum", "specular_count")
        gi, ri = d.integrate(spp_n, seed), od.integrate(spp_n, seed)
        tally("disney integrate", [hostf(gi[k]) for k in keys], [ri[k] for k in keys])
        # the light loop of rlDisney under two lights (MIS; light samples only)
        lts = [O.make_light(center=(2.0, 2.0, 3.0), radius=1.25, radiance=(3.0, 2.0, 1.0)),
               O.make_light(center=(-3.0, 1.0, 2.5), radius=0.5, radiance=(0.5, 4.0, 2.0), mis_mode=1)]
        lgs = [R._capi.SphereLight.from_buffer_copy(bytes(l)) for l in lts]
        tally("disney direct lighting", [hostf(t) for t in d.directLighting(P, lgs, spp_n, seed)],
              od.direct_lighting(hostf(P), lts, spp_n, seed))
        keys = ("direct_diffuse", "direct_specular", "indirect_diffuse", "indirect_specular", "out")
        gs, rs = d.shade(P, lgs, spp_n, seed, env=(1.0, 0.9, 0.8)), od.shade(hostf(P), lts, spp_n, seed, env=(1.0, 0.9, 0.8))
        tally("disney shader_evaluate", [hostf(gs[k]) for k in keys], [rs[k] for k in keys])
    finally:
        del os.environ["RLS_INTEGRATE_GROUP"]
    # --- rlSss probe
    dist = torch.stack([u(32 + j, 0.1, 2.1) for j in range(3)])
    s = R.SssSampler(ctx, N, T, Ks, dist)
    gp = s.getProbeRay(xi[0], xi[1])
    rp = O.Sss(n, hostf(dist), c["KsColor"], N=c["N"], T=c["T"], nthreads=th).probe(hxi[0], hxi[1])
    keys = ("r", "origin", "dir", "maxdist", "pdf", "profile")
    tally("sss probe", [hostf(gp[k]) for k in keys], [rp[k] for k in keys])
    # --- rlSkin
    p = dict(sss_color=Ks, sss_weight=u(35), sss_dist_multiplier=u(36, 0.5, 1.5), sss_scatter_dist=dist,
             specular_color=torch.stack([u(46 + j) for j in range(3)]), specular_weight=u(49),
             specular_roughness=u(50, 0.05, 1.0), specular_ior=u(51, 1.05, 2.55),
             sheen_color=torch.stack([u(52 + j) for j in range(3)]), sheen_weight=u(55),
             sheen_roughness=u(56, 0.05, 1.0), sheen_ior=u(57, 1.05, 2.55))
    sk = R.SkinShader(ctx, wo, N, T, **p)
    gout = sk.sampleEvalPdf(torch.stack(xi))
    rout = O.skin(c["wo"], c["N"], c["T"], {k: hostf(v) for k, v in p.items()}, np.stack(hxi), nthreads=th)
    names = list(O.SKIN_VEC) + list(O.SKIN_SCALAR)
    tally("skin", [hostf(gout[k]) for k in names], [rout[k] for k in names])
    # rlSkin's shader_evaluate over spp_n^2 samples per layer, one light in the two light loops
    os.environ["RLS_INTEGRATE_GROUP"] = group
    try:
        pk = dict(p, sss_scatter_dist=dsmall)
        ski = R.SkinShader(ctx, wo, N, T, **pk)
        one = [O.make_light(center=(0.5, 0.5, 4.0), radius=1.0, radiance=(2.0, 1.5, 1.0))]
        gi = ski.integrate(Psph, sg, spp_n, seed, env=(1.0, 0.9, 0.8), lights=[R._capi.SphereLight.from_buffer_copy(bytes(one[0]))])
        ri = O.skin_integrate(c["wo"], c["N"], c["T"], {k: hostf(v) for k, v in pk.items()}, hostf(Psph), so, spp_n, seed,
                              env=(1.0, 0.9, 0.8), nthreads=th, lights=one)
        keys = ("sheen", "specular", "sss", "out", "sheenFresnel", "specularFresnel", "sssWeight")
        tally("skin shader_evaluate", [hostf(gi[k]) for k in keys], [ri[k] for k in keys])
    finally:
        del os.environ["RLS_INTEGRATE_GROUP"]
    return report
