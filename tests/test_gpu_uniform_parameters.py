"""The UNIFORM_* specialisations of the pointwise kernels (ggx.hip, disney.hip, skin.hip; rlSss: test_gpu_sss_skin.py): with
the node parameters that enter parameter-only arithmetic one value for the batch -- an Arnold parameter is a constant unless a
texture is linked to it -- that arithmetic runs once per thread ahead of the tile loop and stays in scalar registers.  Colours
(rlGgx's specColor, rlDisney's base_color) may be per-point planes beside it: the colour-map case.  The results must be the
per-point evaluation's: every verb is compared bit for bit with the STREAMED kernel given the same values as per-point planes,
and with the oracle under the gates of the mixed-parameter tests."""
import numpy as np
import pytest
import torch

import cases
import rlshaders_amd as R
from gpu_util import dev, host, ggx_oracle

pytestmark = pytest.mark.gpu

N = 1 << 14

GGX_UNIFORM = {
    "plastic": dict(KsColor=(0.9, 0.8, 0.7), roughness=0.35, ior=1.5, anisotropic=0.25),
    "brushed": dict(KsColor=(1.0, 1.0, 1.0), roughness=0.3, ior=0.47, anisotropic=1.0),
    "mirror_like": dict(KsColor=(1.0, 1.0, 1.0), roughness=0.0, ior=1.0, anisotropic=0.0),
    "rough_dense": dict(KsColor=(0.2, 0.5, 1.0), roughness=1.0, ior=2.55, anisotropic=0.9),
    "degenerate_ior": dict(KsColor=(1.0, 1.0, 1.0), roughness=0.5, ior=0.0, anisotropic=0.5),
}


def _same(a, b, what):
    a, b = host(a) if torch.is_tensor(a) else np.asarray(a), host(b) if torch.is_tensor(b) else np.asarray(b)
    ok = (a.view(np.uint32) == b.view(np.uint32)) if a.dtype == np.float32 else (a == b)
    if a.dtype == np.float32:
        ok = ok | (np.isnan(a) & np.isnan(b))
    assert ok.all(), (what, int((~ok).sum()))


def _full3(v, n):
    return v if isinstance(v, np.ndarray) and v.ndim == 2 else _planes(v, n)


def _planes(v, n):
    return np.repeat(np.asarray(v, np.float32)[:, None], n, axis=1) if np.ndim(v) else np.full(n, v, np.float32)


@pytest.mark.parametrize("colour_map", [False, True])
@pytest.mark.parametrize("exiting", [False, True])
@pytest.mark.parametrize("name", sorted(GGX_UNIFORM))
def test_ggx_uniform_equals_streamed_and_oracle(gpu, oracle, name, exiting, colour_map):
    p = dict(GGX_UNIFORM[name])
    wo, Nn, T = cases.frame(cases.SEED_PARITY, N)
    x = cases.xi(cases.SEED_PARITY, N, 4)
    if colour_map:                       # specColor textured, the rest of the node plain
        p["KsColor"] = cases.xi(cases.SEED_EDGE, N, 3)
    ex = (np.arange(N) % 3 == 0).astype(np.uint8) if exiting else None
    exd = None if ex is None else torch.from_numpy(ex).cuda()
    su = R.GgxSampler(gpu, dev(wo), dev(Nn), dev(T), specColor=dev(p["KsColor"]), ior=p["ior"], roughness=p["roughness"],
                      anisotropic=p["anisotropic"], exiting=exd)
    ss = R.GgxSampler(gpu, dev(wo), dev(Nn), dev(T), specColor=dev(_full3(p["KsColor"], N)), ior=dev(_planes(p["ior"], N)),
                      roughness=dev(_planes(p["roughness"], N)), anisotropic=dev(_planes(p["anisotropic"], N)), exiting=exd)
    og = ggx_oracle(oracle, dict(wo=wo, N=Nn, T=T, **p), exiting=ex)
    dx = [dev(t) for t in x]
    # reflect + refract in one pass, and each verb on its own
    for k, (a, b, c) in enumerate(zip(su.reflectRefract(*dx), ss.reflectRefract(*dx), og.reflect_refract(*x))):
        _same(a, b, (name, "reflectRefract vs streamed", k))
        if cases.strict_parity():
            _same(a, c, (name, "reflectRefract vs oracle", k))
        else:
            cases.assert_tight(cases.summarize(cases.rel_err(host(a), c)), (name, "reflectRefract", k))
    wi = su.sampleEvalPdf(dx[0], dx[1])[0]
    for k, (a, b) in enumerate(zip(su.sampleEvalPdf(dx[0], dx[1]), ss.sampleEvalPdf(dx[0], dx[1]))):
        _same(a, b, (name, "sampleEvalPdf", k))
    for k, (a, b) in enumerate(zip(su.evalSample(dx[0], dx[1]), ss.evalSample(dx[0], dx[1]))):
        _same(a, b, (name, "evalSample", k))
    _same(su.evalBrdf(wi), ss.evalBrdf(wi), (name, "evalBrdf"))
    _same(su.evalPdf(wi), ss.evalPdf(wi), (name, "evalPdf"))
    _same(su.ndfPdf(wi), ss.ndfPdf(wi), (name, "ndfPdf"))
    for k, (a, b) in enumerate(zip(su.refractSample(dx[2], dx[3]), ss.refractSample(dx[2], dx[3]))):
        _same(a, b, (name, "refractSample", k))
    for kern in (R.RLS_KERNEL_VNDF, R.RLS_KERNEL_NDF):
        _same(su.microfacet(dx[0], dx[1], kern), ss.microfacet(dx[0], dx[1], kern), (name, "microfacet", kern))
    # evalBrdf / evalPdf against the oracle on the sampled directions
    hwi = host(wi)
    for got, ref, what in ((su.evalBrdf(wi), og.eval(hwi), "evalBrdf"), (su.evalPdf(wi), og.pdf(hwi), "evalPdf")):
        if cases.strict_parity():
            _same(got, ref, (name, what, "oracle"))
        else:
            cases.assert_tight(cases.summarize(cases.rel_err(host(got), ref)), (name, what))


DISNEY_UNIFORM = dict(cases.DISNEY_PRESETS)
DISNEY_UNIFORM.update({
    "every_lobe": dict(base_color=(0.85, 0.7047, 0.2057), subsurface=0.2, metallic=0.3, specular=0.5, specular_tint=0.25,
                       roughness=0.4, anisotropic=0.4, sheen=0.5, sheen_tint=0.5, clearcoat=0.6, clearcoat_gloss=0.7),
    "black_base": dict(base_color=(0.0, 0.0, 0.0), specular=1.0, specular_tint=1.0, sheen=1.0, sheen_tint=1.0, roughness=0.0,
                       clearcoat=1.0, clearcoat_gloss=1.0),
    "extremes": dict(base_color=(1.0, 0.0, 1.0), subsurface=1.0, metallic=1.0, specular=0.0, roughness=1.0, anisotropic=1.0,
                     clearcoat=1.0, clearcoat_gloss=0.0),
})
DISNEY_DEFAULTS = dict(subsurface=0.0, metallic=0.0, specular=0.5, specular_tint=0.0, roughness=0.5, anisotropic=0.0, sheen=0.0,
                       sheen_tint=0.5, clearcoat=0.0, clearcoat_gloss=1.0)


@pytest.mark.parametrize("colour_map", [False, True])
@pytest.mark.parametrize("name", sorted(DISNEY_UNIFORM))
def test_disney_uniform_equals_streamed_and_oracle(gpu, oracle, name, colour_map):
    p = dict(DISNEY_DEFAULTS, **DISNEY_UNIFORM[name])
    wo, Nn, T = cases.frame(cases.SEED_PARITY, N)
    x = cases.xi(cases.SEED_PARITY, N, 2)
    dx = [dev(t) for t in x]
    if colour_map:                       # base_color textured (with black texels), the ten scalars plain: UNIFORM_SCALARS
        p["base_color"] = cases.xi(cases.SEED_EDGE, N, 3)
        p["base_color"][:, ::7] = 0.0
    du = R.DisneySampler(gpu, dev(wo), dev(Nn), dev(T), **{k: dev(v) for k, v in p.items()})
    ds = R.DisneySampler(gpu, dev(wo), dev(Nn), dev(T), **{k: dev(_full3(v, N)) for k, v in p.items()})
    od = oracle.Disney(wo, Nn, T, nthreads=4, **p)
    for lobe in (R.RLS_RAY_DIFFUSE, R.RLS_RAY_GLOSSY):
        du.setSampleType(lobe); ds.setSampleType(lobe)
        gu, gs, ref = du.sampleEvalPdf(*dx), ds.sampleEvalPdf(*dx), od.sample_eval_pdf(lobe, x[0], x[1])
        for k, (a, b, c) in enumerate(zip(gu, gs, ref)):
            _same(a, b, (name, lobe, "triple vs streamed", k))
            if cases.strict_parity():
                _same(a, c, (name, lobe, "triple vs oracle", k))
            else:
                cases.assert_tight(cases.summarize(cases.rel_err(host(a), c)), (name, lobe, "triple", k))
        wi = gu[0]
        _same(du.evalSample(*dx), ds.evalSample(*dx), (name, lobe, "evalSample"))
        _same(du.evalBrdf(wi), ds.evalBrdf(wi), (name, lobe, "evalBrdf"))
        _same(du.evalPdf(wi), ds.evalPdf(wi), (name, lobe, "evalPdf"))


def test_random_uniform_parameter_sets(gpu, oracle):
    """tests/parity_sweep.py, sweep_uniform: random parameter VALUES with the borders over-sampled (layer weights around 1e-4,
    scatter distances across the reciprocal window, ior below 1 and at its clamp), every one-sample verb of the four units."""
    import parity_sweep
    rep = parity_sweep.sweep_uniform(gpu, 1 << 16, 20261003, draws=6, verbose=False)
    assert len(rep) == 6
    for name, r in rep.items():
        assert r["beyond_1e5"] == 0, (name, r)
        if cases.strict_parity():
            assert r["words_differing"] == 0, (name, r)
        else:
            assert r["words_differing"] <= max(4, r["words"] // 100000), (name, r)
