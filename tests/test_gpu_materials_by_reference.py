"""Parameters by reference (rls_material_index, include/rlshaders_amd.h): a batch that mixes the hits of many node instances
carries a per-point material id and the parameters as per-material columns.  The results must be those of the same values
expanded into per-point planes -- bit for bit, for every verb of the four units and for the n^2-spp loops and whole-node
kernels -- and therefore the oracle's."""
import numpy as np
import pytest
import torch

import cases
import rlshaders_amd as R
from gpu_util import dev, host, ggx_oracle, disney_oracle

pytestmark = pytest.mark.gpu

N = 1 << 14
M = 37                       # node instances in the batch


def _same(a, b, what):
    a, b = host(a) if torch.is_tensor(a) else np.asarray(a), host(b) if torch.is_tensor(b) else np.asarray(b)
    ok = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b)) if a.dtype == np.float32 else (a == b)
    assert ok.all(), (what, int((~ok).sum()))


def _ids(n, m, seed=5):
    rng = np.random.default_rng(seed)
    ids = rng.integers(0, m, n).astype(np.int32)
    ids[:m] = np.arange(m)                  # every material occurs
    return ids


def _expand(col, ids):
    return np.ascontiguousarray(col[..., ids])


def test_ggx_every_verb(gpu, oracle):
    wo, Nn, T = cases.frame(cases.SEED_PARITY, N)
    x = cases.xi(cases.SEED_PARITY, N, 4)
    ids = _ids(N, M)
    t = cases.ggx_mixed(cases.SEED_EDGE, M)                       # M parameter sets
    cols = dict(KsColor=t["KsColor"], ior=t["ior"], roughness=t["roughness"], anisotropic=t["anisotropic"])
    cols["ior"][::5] = 0.7                                          # some instances below 1
    ex = (np.arange(N) % 3 == 0).astype(np.uint8)
    exd = torch.from_numpy(ex).cuda()
    mat = (dev(ids), M)
    sr = R.GgxSampler(gpu, dev(wo), dev(Nn), dev(T), specColor=dev(cols["KsColor"]), ior=dev(cols["ior"]),
                      roughness=dev(cols["roughness"]), anisotropic=dev(cols["anisotropic"]), exiting=exd, materials=mat)
    pl = {k: _expand(v, ids) for k, v in cols.items()}
    sp = R.GgxSampler(gpu, dev(wo), dev(Nn), dev(T), specColor=dev(pl["KsColor"]), ior=dev(pl["ior"]),
                      roughness=dev(pl["roughness"]), anisotropic=dev(pl["anisotropic"]), exiting=exd)
    dx = [dev(v) for v in x]
    for k, (a, b) in enumerate(zip(sr.reflectRefract(*dx), sp.reflectRefract(*dx))):
        _same(a, b, ("reflectRefract", k))
    wi = sp.sampleEvalPdf(dx[0], dx[1])[0]
    for k, (a, b) in enumerate(zip(sr.sampleEvalPdf(dx[0], dx[1]), sp.sampleEvalPdf(dx[0], dx[1]))):
        _same(a, b, ("sampleEvalPdf", k))
    _same(sr.evalBrdf(wi), sp.evalBrdf(wi), "evalBrdf")
    _same(sr.evalPdf(wi), sp.evalPdf(wi), "evalPdf")
    _same(sr.ndfPdf(wi), sp.ndfPdf(wi), "ndfPdf")
    for k, (a, b) in enumerate(zip(sr.refractSample(dx[2], dx[3]), sp.refractSample(dx[2], dx[3]))):
        _same(a, b, ("refractSample", k))
    _same(sr.microfacet(dx[0], dx[1]), sp.microfacet(dx[0], dx[1]), "microfacet")
    # ... and the oracle on the expanded values
    og = ggx_oracle(oracle, dict(wo=wo, N=Nn, T=T, **pl), exiting=ex)
    for k, (a, c) in enumerate(zip(sr.reflectRefract(*dx), og.reflect_refract(*x))):
        if cases.strict_parity():
            _same(a, c, ("reflectRefract vs oracle", k))
        else:
            cases.assert_tight(cases.summarize(cases.rel_err(host(a), c)), ("reflectRefract", k))
    # the loops and the whole node: the shader parameters (Kd, Kt ...) are columns too
    u = lambda j, m=M: oracle.gen_uniform(9, 0, m, oracle.S_PARAM0 + j)
    shc = dict(KdColor=np.stack([u(0), u(1), u(2)]), Kd=u(3), diffuseRoughness=u(4), Ks=u(5), KtColor=np.stack([u(6), u(7), u(8)]),
               Kt=u(9))
    P = dev(np.stack([oracle.gen_uniform(9, 0, N, 40 + j, 0.0, 4.0 if j < 2 else 1.0) for j in range(3)]))
    lights = [R.make_light(center=(2.0, 2.0, 3.0), radius=1.25, radiance=(3.0, 2.0, 1.0)),
              R.make_light(center=(-3.0, 1.0, 2.5), radius=0.5, radiance=(0.5, 4.0, 2.0))]
    ga = sr.shade(P, lights, 2, 77, env=(1.0, 0.9, 0.8), **{k: dev(v) for k, v in shc.items()})
    gb = sp.shade(P, lights, 2, 77, env=(1.0, 0.9, 0.8), **{k: dev(_expand(v, ids)) for k, v in shc.items()})
    for k in ga:
        _same(ga[k], gb[k], ("shade", k))
    for k, (a, b) in enumerate(zip(sr.integrate(2, 77), sp.integrate(2, 77))):
        _same(a, b, ("integrate", k))
    for k, (a, b) in enumerate(zip(sr.integrateRefract(2, 77), sp.integrateRefract(2, 77))):
        _same(a, b, ("integrateRefract", k))
    kw = dict(KdColor="KdColor", Kd="Kd", diffuseRoughness="diffuseRoughness", Ks="Ks")
    for k, (a, b) in enumerate(zip(sr.directLighting(P, lights, 2, 77, **{q: dev(shc[q]) for q in kw}),
                                   sp.directLighting(P, lights, 2, 77, **{q: dev(_expand(shc[q], ids)) for q in kw}))):
        _same(a, b, ("directLighting", k))


def test_disney_every_verb(gpu, oracle):
    wo, Nn, T = cases.frame(cases.SEED_PARITY, N)
    x = cases.xi(cases.SEED_PARITY, N, 2)
    dx = [dev(v) for v in x]
    ids = _ids(N, M, 6)
    t = cases.disney_mixed(cases.SEED_EDGE, M)
    cols = {k: t[k] for k in ("base_color",) + tuple(oracle.DISNEY_SCALARS)}
    cols["base_color"][:, ::9] = 0.0
    dr = R.DisneySampler(gpu, dev(wo), dev(Nn), dev(T), materials=(dev(ids), M), **{k: dev(v) for k, v in cols.items()})
    pl = {k: _expand(v, ids) for k, v in cols.items()}
    dp = R.DisneySampler(gpu, dev(wo), dev(Nn), dev(T), **{k: dev(v) for k, v in pl.items()})
    od = disney_oracle(oracle, dict(wo=wo, N=Nn, T=T, **pl))
    for lobe in (R.RLS_RAY_DIFFUSE, R.RLS_RAY_GLOSSY):
        dr.setSampleType(lobe); dp.setSampleType(lobe)
        ga, gb, ref = dr.sampleEvalPdf(*dx), dp.sampleEvalPdf(*dx), od.sample_eval_pdf(lobe, x[0], x[1])
        for k, (a, b, c) in enumerate(zip(ga, gb, ref)):
            _same(a, b, (lobe, "triple", k))
            if cases.strict_parity():
                _same(a, c, (lobe, "triple vs oracle", k))
        _same(dr.evalSample(*dx), dp.evalSample(*dx), (lobe, "evalSample"))
        _same(dr.evalBrdf(ga[0]), dp.evalBrdf(ga[0]), (lobe, "evalBrdf"))
        _same(dr.evalPdf(ga[0]), dp.evalPdf(ga[0]), (lobe, "evalPdf"))
    # the alternates the reference compiles but never selects read the closure through the same index
    for kind in (0, 1):
        _same(dr.altSample(kind, *dx), dp.altSample(kind, *dx), ("altSample", kind))
    ia, ib = dr.integrate(2, 77), dp.integrate(2, 77)
    for k in ia:
        _same(ia[k], ib[k], ("integrate", k))
    # streamed mode in chunks (rls_disney_integrate_chunked): by reference a chunk advances the material IDS, not the
    # per-material columns (M = 37 floats: a column advanced by a chunk's first point would be read far out of bounds).  Sums =
    # the unchunked call's, every chunk's samples = the planes form's, on ragged chunks (N is not a multiple of 3000)
    seen = {"ref": [], "planes": []}
    grab = lambda key: (lambda first, count, ch: seen[key].append((first, count, [host(ch[k]).copy() for k in ("wi", "f", "pdf")])))
    ca, _ = dr.integrateChunked(2, 77, 3000, consume=grab("ref"))
    cb, _ = dp.integrateChunked(2, 77, 3000, consume=grab("planes"))
    for k in ia:
        _same(ca[k], ia[k], ("integrateChunked by reference vs integrate", k))
        _same(cb[k], ia[k], ("integrateChunked planes vs integrate", k))
    assert [c[:2] for c in seen["ref"]] == [(p0, min(3000, N - p0)) for p0 in range(0, N, 3000)] == [c[:2] for c in seen["planes"]]
    for (p0, count, a), (_, _, b) in zip(seen["ref"], seen["planes"]):
        m = 2 * 4 * count
        for k, (u, v) in enumerate(zip(a, b)):
            assert np.array_equal(u[..., :m].view(np.uint32), v[..., :m].view(np.uint32)), ("chunk samples", p0, k)
    P = dev(np.stack([oracle.gen_uniform(9, 0, N, 40 + j, 0.0, 4.0 if j < 2 else 1.0) for j in range(3)]))
    lights = [R.make_light(center=(2.0, 2.0, 3.0), radius=1.25, radiance=(3.0, 2.0, 1.0))]
    sa, sb = dr.shade(P, lights, 2, 77, env=(1.0, 0.9, 0.8)), dp.shade(P, lights, 2, 77, env=(1.0, 0.9, 0.8))
    for k in sa:
        _same(sa[k], sb[k], ("shade", k))
    for k, (a, b) in enumerate(zip(dr.directLighting(P, lights, 2, 77), dp.directLighting(P, lights, 2, 77))):
        _same(a, b, ("directLighting", k))


def test_sss_and_skin(gpu, oracle):
    wo, Nn, T = cases.frame(cases.SEED_PARITY, N)
    xi = cases.xi(cases.SEED_PARITY, N, 6)
    ids = _ids(N, M, 7)
    t = cases.skin_mixed(cases.SEED_EDGE, M)["params"]
    t["sss_scatter_dist"][0, ::6] = 1e-5                    # outside the reciprocal window
    t["sheen_weight"][::4] = 0.0                            # layers off in some instances
    t["specular_weight"][1::4] = 1e-4
    mat = (dev(ids), M)
    pl = {k: _expand(v, ids) for k, v in t.items()}
    # NDProfile / SssSampler
    pr = R.NDProfile(gpu, N, dev(t["sss_scatter_dist"]), dev(t["sss_color"]), multiplier=dev(t["sss_dist_multiplier"]), materials=mat)
    pp = R.NDProfile(gpu, N, dev(pl["sss_scatter_dist"]), dev(pl["sss_color"]), multiplier=dev(pl["sss_dist_multiplier"]))
    rx = dev(xi[0])
    for k, (a, b) in enumerate(zip(pr.sample(rx), pp.sample(rx))):
        _same(a, b, ("nd sample", k))
    r = pp.sample(rx)[0] + 0.01
    _same(pr.getPdf(r), pp.getPdf(r), "getPdf")
    _same(pr.evalProfile(r), pp.evalProfile(r), "evalProfile")
    sr = R.SssSampler(gpu, dev(Nn), dev(T), dev(t["sss_color"]), dev(t["sss_scatter_dist"]), multiplier=dev(t["sss_dist_multiplier"]),
                      materials=mat)
    sp = R.SssSampler(gpu, dev(Nn), dev(T), dev(pl["sss_color"]), dev(pl["sss_scatter_dist"]), multiplier=dev(pl["sss_dist_multiplier"]))
    ga, gb = sr.getProbeRay(dev(xi[0]), dev(xi[1])), sp.getProbeRay(dev(xi[0]), dev(xi[1]))
    for k in ga:
        _same(ga[k], gb[k], ("probe", k))
    scene = R.make_scene("sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
    _same(sr.integrateScatter(dev(Nn), scene, 2, 77), sp.integrateScatter(dev(Nn), scene, 2, 77), "integrateScatter")
    # rlSkin: the one-sample composite and shader_evaluate
    kr = R.SkinShader(gpu, dev(wo), dev(Nn), dev(T), materials=mat, **{k: dev(v) for k, v in t.items()})
    kp = R.SkinShader(gpu, dev(wo), dev(Nn), dev(T), **{k: dev(v) for k, v in pl.items()})
    ga, gb = kr.sampleEvalPdf(dev(xi)), kp.sampleEvalPdf(dev(xi))
    ref = oracle.skin(wo, Nn, T, pl, xi, nthreads=4)
    for k in ga:
        _same(ga[k], gb[k], ("skin", k))
        if cases.strict_parity():
            _same(ga[k], ref[k], ("skin vs oracle", k))
    ia = kr.integrate(dev(Nn), scene, 2, 77, env=(1.0, 0.9, 0.8))
    ib = kp.integrate(dev(Nn), scene, 2, 77, env=(1.0, 0.9, 0.8))
    for k in ia:
        _same(ia[k], ib[k], ("skin integrate", k))


def test_ids_are_clamped_and_arguments_checked(gpu):
    """an id beyond the table reads the LAST entry (documented), never past it; a table without entries is refused"""
    n, m = 4096, 5
    wo, Nn, T = cases.frame(3, n)
    x = cases.xi(3, n, 2)
    ids = np.full(n, 4, np.int32)
    wild = ids.copy()
    wild[::3] = 1000
    wild[1::3] = -1                                          # 0xffffffff as the library sees it
    col = lambda lo, hi: np.linspace(lo, hi, m).astype(np.float32)
    kw = dict(specColor=(1.0, 1.0, 1.0), ior=dev(col(1.1, 2.0)), roughness=dev(col(0.1, 0.9)), anisotropic=dev(col(0.0, 0.8)))
    a = R.GgxSampler(gpu, dev(wo), dev(Nn), dev(T), materials=(dev(ids), m), **kw).sampleEvalPdf(dev(x[0]), dev(x[1]))
    b = R.GgxSampler(gpu, dev(wo), dev(Nn), dev(T), materials=(dev(wild), m), **kw).sampleEvalPdf(dev(x[0]), dev(x[1]))
    for k, (p, q) in enumerate(zip(a, b)):
        _same(p, q, ("clamped", k))
    with pytest.raises(ValueError):
        R.GgxSampler(gpu, dev(wo), dev(Nn), dev(T), materials=(dev(ids), 0), **kw)
    with pytest.raises(TypeError):
        R.GgxSampler(gpu, dev(wo), dev(Nn), dev(T), materials=(dev(ids.astype(np.int64)), m), **kw)
    s = R.GgxSampler(gpu, dev(wo), dev(Nn), dev(T), materials=(dev(ids), m), **kw)
    s.c.materials.count = 0                                  # behind the mirror's back: the C ABI refuses it
    with pytest.raises(R.RlsError):
        s.sampleEvalPdf(dev(x[0]), dev(x[1]))
