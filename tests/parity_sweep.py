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


def _draw_uniform_parameters(rng) -> dict:
    """One node's worth of parameter VALUES (every parameter one number for the batch), drawn so that the borders matter:
    weights below / on / above the 1e-4 layer gates, scatter distances across the reciprocal window, ior below 1 and at the
    1e-4 clamp, roughness at both ends."""
    pick = lambda *v: float(v[rng.integers(len(v))])
    f = lambda lo, hi: float(np.float32(rng.uniform(lo, hi)))
    logu = lambda lo, hi: float(np.float32(10.0 ** rng.uniform(lo, hi)))
    rough = lambda: pick(f(0.0, 1.0), f(0.0, 1.0), f(0.0, 1.0), 0.0, 1.0, 1e-3)
    ior = lambda: pick(f(1.05, 2.55), f(1.05, 2.55), f(0.3, 1.0), 1.0, 0.0, 5e-5)
    weight = lambda: pick(f(0.0, 1.0), f(0.0, 1.0), f(0.0, 1.0), 0.0, 1e-4, float(np.nextafter(np.float32(1e-4), np.float32(1))), 1.0)
    dist = lambda: pick(logu(-1.5, 0.5), logu(-1.5, 0.5), logu(-1.5, 0.5), logu(-6, 6), 2.0 ** -13, 2.0 ** 14, 1e-5, 0.0)
    col = lambda: (f(0, 1), f(0, 1), f(0, 1))
    return dict(
        ggx=dict(KsColor=col(), roughness=rough(), ior=ior(), anisotropic=pick(f(0, 1), f(0, 1), 0.0, 1.0)),
        disney=dict(base_color=(0.0, 0.0, 0.0) if rng.integers(4) == 0 else col(), subsurface=f(0, 1), metallic=f(0, 1), specular=f(0, 1),
                    specular_tint=f(0, 1), roughness=rough(), anisotropic=pick(f(0, 1), 0.0, 1.0), sheen=f(0, 1), sheen_tint=f(0, 1),
                    clearcoat=pick(f(0, 1), f(0, 1), 0.0, 1.0), clearcoat_gloss=pick(f(0, 1), f(0, 1), 0.0, 1.0)),
        skin=dict(sss_color=col(), sss_weight=weight(), sss_dist_multiplier=pick(1.0, f(0.5, 1.5), f(0.0, 3.0)),
                  sss_scatter_dist=(dist(), dist(), dist()), specular_color=col(), specular_weight=weight(),
                  specular_roughness=rough(), specular_ior=ior(), sheen_color=col(), sheen_weight=weight(),
                  sheen_roughness=rough(), sheen_ior=ior()))


def sweep_uniform(ctx, n: int, seed: int, draws: int = 8, verbose: bool = True) -> dict:
    """The UNIFORM_ALL kernels (parameter-only arithmetic once per thread): `draws` random parameter sets, each one value per
    parameter for a batch of n device-generated shading points, every one-sample verb of the four units against the oracle,
    counted in output words that differ at all."""
    th = O.hardware_threads()
    hostf = lambda t: t.contiguous().cpu().numpy()
    rng = np.random.default_rng(seed)
    report = {}

    def tally(name, got, ref):
        r = _words(got, ref)
        t = report.setdefault(name, dict(words_differing=0, words=0, max_rel_err=0.0, beyond_1e5=0))
        for k in ("words_differing", "words", "beyond_1e5"):
            t[k] += r[k]
        t["max_rel_err"] = max(t["max_rel_err"], r["max_rel_err"])

    wo, N, T = R.gen_frame(ctx, seed, 0, n)
    hwo, hN, hT = hostf(wo), hostf(N), hostf(T)
    xi = [R.gen_uniform(ctx, seed, 0, n, 11 + j) for j in range(6)]
    hxi = [hostf(t) for t in xi]
    exiting = (torch.arange(n, device=wo.device) % 5 == 0).to(torch.uint8)
    for k in range(draws):
        p = _draw_uniform_parameters(rng)
        g = p["ggx"]
        ex = exiting if k % 2 else None
        s = R.GgxSampler(ctx, wo, N, T, specColor=g["KsColor"], ior=g["ior"], roughness=g["roughness"],
                         anisotropic=g["anisotropic"], exiting=ex)
        og = O.Ggx(hwo, hN, hT, KsColor=g["KsColor"], ior=g["ior"], roughness=g["roughness"], anisotropic=g["anisotropic"],
                   exiting=None if ex is None else hostf(ex), nthreads=th)
        tally("ggx reflect+refract, uniform", [hostf(t) for t in s.reflectRefract(*xi[:4])], og.reflect_refract(*hxi[:4]))
        wi = og.sample(hxi[0], hxi[1])[0]
        dwi = torch.from_numpy(wi).cuda()
        tally("ggx evalBrdf / evalPdf, uniform", [hostf(s.evalBrdf(dwi)), hostf(s.evalPdf(dwi))], [og.eval(wi), og.pdf(wi)])
        d = R.DisneySampler(ctx, wo, N, T, **p["disney"])
        od = O.Disney(hwo, hN, hT, nthreads=th, **p["disney"])
        for lobe, nm in ((R.RLS_RAY_DIFFUSE, "diffuse"), (R.RLS_RAY_GLOSSY, "glossy")):
            d.setSampleType(lobe)
            tally(f"disney {nm}, uniform", [hostf(t) for t in d.sampleEvalPdf(xi[0], xi[1])],
                  od.sample_eval_pdf(lobe, hxi[0], hxi[1]))
        sk = p["skin"]
        ss = R.SssSampler(ctx, N, T, sk["sss_color"], sk["sss_scatter_dist"], multiplier=sk["sss_dist_multiplier"])
        gp = ss.getProbeRay(xi[0], xi[1])
        rp = O.Sss(n, sk["sss_scatter_dist"], sk["sss_color"], multiplier=sk["sss_dist_multiplier"], N=hN, T=hT,
                   nthreads=th).probe(hxi[0], hxi[1])
        keys = ("r", "origin", "dir", "maxdist", "pdf", "profile")
        tally("sss probe, uniform", [hostf(gp[q]) for q in keys], [rp[q] for q in keys])
        gout = R.SkinShader(ctx, wo, N, T, **sk).sampleEvalPdf(torch.stack(xi))
        rout = O.skin(hwo, hN, hT, sk, np.stack(hxi), nthreads=th)
        names = list(O.SKIN_VEC) + list(O.SKIN_SCALAR)
        tally("skin, uniform", [hostf(gout[q]) for q in names], [rout[q] for q in names])
        if verbose:
            bad = {q: r["words_differing"] for q, r in report.items() if r["words_differing"]}
            print(f"uniform sweep seed {seed} draw {k}: differing so far {bad or 0}", flush=True)
            if bad:
                print("   parameters of this draw:", p, flush=True)
    return report


def sweep_by_reference(ctx, n: int, seed: int, m: int, spp_n: int = 2, verbose: bool = True) -> dict:
    """Parameters by reference (rls_material_index) at scale: m node instances' parameters as columns and a material id per
    point against the same values expanded into per-point planes -- every one-sample verb, the integrators, the light loops and
    the three whole-node kernels, GPU against GPU (the planes kernels are what sweep() holds against the oracle), counted in
    output words that differ at all.  A tenth of the ids lie beyond the table (clamped to its last entry)."""
    dev = wo_dev = None
    hostf = lambda t: t.contiguous().cpu().numpy()
    report = {}

    def tally(name, got, ref):
        got = [hostf(t) for t in (got.values() if isinstance(got, dict) else got)]
        ref = [hostf(t) for t in (ref.values() if isinstance(ref, dict) else ref)]
        r = _words(got, ref)
        t = report.setdefault(name, dict(words_differing=0, words=0, max_rel_err=0.0, beyond_1e5=0))
        for k in ("words_differing", "words", "beyond_1e5"):
            t[k] += r[k]
        t["max_rel_err"] = max(t["max_rel_err"], r["max_rel_err"])
        if verbose:
            print(f"by reference seed {seed} m {m} {name}: {r['words_differing']} of {r['words']} words differ", flush=True)

    wo, N, T = R.gen_frame(ctx, seed, 0, n)
    u = lambda stream, lo=0.0, hi=1.0: R.gen_uniform(ctx, seed, 0, n, stream, lo, hi)
    c = lambda stream, lo=0.0, hi=1.0: R.gen_uniform(ctx, seed + 1, 0, m, stream, lo, hi)          # a column: one entry per instance
    c3 = lambda stream, lo=0.0, hi=1.0: torch.stack([c(stream + j, lo, hi) for j in range(3)])
    xi = [u(11 + j) for j in range(6)]
    raw = (u(70) * (m * 1.1)).to(torch.int32)                       # ~9 % of the ids beyond the table
    ids = raw.clamp(0, m - 1)                                       # what the library makes of them
    ex = lambda col: col[..., ids.long()].contiguous()              # the same values as per-point planes
    mat = (raw, m)
    P = torch.stack([u(40, 0, 4), u(41, 0, 4), u(42, 0, 1)])
    lights = [R.make_light(center=(2.0, 2.0, 3.0), radius=1.25, radiance=(3.0, 2.0, 1.0)),
              R.make_light(center=(-3.0, 1.0, 2.5), radius=0.5, radiance=(0.5, 4.0, 2.0), mis_mode=2)]
    os.environ["RLS_INTEGRATE_GROUP"] = "1"
    try:
        # rlGgx
        g = dict(specColor=c3(8), ior=c(6, 0.4, 2.55), roughness=c(5, 0.0, 1.0), anisotropic=c(7, 0.0, 1.0))
        sh = dict(KdColor=c3(20), Kd=c(23), diffuseRoughness=c(24), Ks=c(25), KtColor=c3(26), Kt=c(29))
        a = R.GgxSampler(ctx, wo, N, T, materials=mat, **g)
        b = R.GgxSampler(ctx, wo, N, T, **{k: ex(v) for k, v in g.items()})
        tally("ggx reflect+refract", a.reflectRefract(*xi[:4]), b.reflectRefract(*xi[:4]))
        wi = b.sampleEvalPdf(xi[0], xi[1])[0]
        tally("ggx evalBrdf / evalPdf", [a.evalBrdf(wi), a.evalPdf(wi)], [b.evalBrdf(wi), b.evalPdf(wi)])
        tally("ggx integrate / integrateRefract", list(a.integrate(spp_n, seed)) + list(a.integrateRefract(spp_n, seed)),
              list(b.integrate(spp_n, seed)) + list(b.integrateRefract(spp_n, seed)))
        tally("ggx shader_evaluate", a.shade(P, lights, spp_n, seed, env=(1.0, 0.9, 0.8), **sh),
              b.shade(P, lights, spp_n, seed, env=(1.0, 0.9, 0.8), **{k: ex(v) for k, v in sh.items()}))
        # rlDisney
        d = dict(base_color=c3(8), **{k: c(32 + j) for j, k in enumerate(R._capi.DISNEY_SCALARS)})
        a = R.DisneySampler(ctx, wo, N, T, materials=mat, **d)
        b = R.DisneySampler(ctx, wo, N, T, **{k: ex(v) for k, v in d.items()})
        for lobe in (R.RLS_RAY_DIFFUSE, R.RLS_RAY_GLOSSY):
            a.setSampleType(lobe); b.setSampleType(lobe)
            tally("disney triple", a.sampleEvalPdf(xi[0], xi[1]), b.sampleEvalPdf(xi[0], xi[1]))
        tally("disney integrate", a.integrate(spp_n, seed), b.integrate(spp_n, seed))
        tally("disney shader_evaluate", a.shade(P, lights, spp_n, seed, env=(1.0, 0.9, 0.8)), b.shade(P, lights, spp_n, seed, env=(1.0, 0.9, 0.8)))
        # rlSss / rlSkin
        k = dict(sss_color=c3(8), sss_weight=c(35), sss_dist_multiplier=c(36, 0.5, 1.5), sss_scatter_dist=c3(50, 0.02, 2.1),
                 specular_color=c3(46), specular_weight=c(49), specular_roughness=c(53, 0.05, 1.0), specular_ior=c(54, 1.05, 2.55),
                 sheen_color=c3(55), sheen_weight=c(58), sheen_roughness=c(59, 0.05, 1.0), sheen_ior=c(60, 1.05, 2.55))
        a = R.SssSampler(ctx, N, T, k["sss_color"], k["sss_scatter_dist"], multiplier=k["sss_dist_multiplier"], materials=mat)
        b = R.SssSampler(ctx, N, T, ex(k["sss_color"]), ex(k["sss_scatter_dist"]), multiplier=ex(k["sss_dist_multiplier"]))
        tally("sss probe", a.getProbeRay(xi[0], xi[1]), b.getProbeRay(xi[0], xi[1]))
        scene = R.make_scene("sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
        tally("sss integrateScatter", [a.integrateScatter(N, scene, spp_n, seed)], [b.integrateScatter(N, scene, spp_n, seed)])
        a = R.SkinShader(ctx, wo, N, T, materials=mat, **k)
        b = R.SkinShader(ctx, wo, N, T, **{q: ex(v) for q, v in k.items()})
        tally("skin", a.sampleEvalPdf(torch.stack(xi)), b.sampleEvalPdf(torch.stack(xi)))
        tally("skin shader_evaluate", a.integrate(N, scene, spp_n, seed, env=(1.0, 0.9, 0.8), lights=lights[:1]),
              b.integrate(N, scene, spp_n, seed, env=(1.0, 0.9, 0.8), lights=lights[:1]))
    finally:
        del os.environ["RLS_INTEGRATE_GROUP"]
    return report
