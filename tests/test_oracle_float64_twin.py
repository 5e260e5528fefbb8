"""CPU: the oracle against an INDEPENDENT float64 restatement of the same reference lines (tests/twin64.py, numpy) -- a
second anchor beside the survey's known answers and the closed forms: the two were written separately, in two languages and
two precisions, from src/rlGgx.cpp:14-99, src/rlGgx.h:72-357 and src/rlSss.cpp:20-106, so a transcription error in either
shows up as a disagreement far above round-off.  What they may differ by is the reference's own fp32 rounding: medians at
1e-7; tails where its formulas are ill conditioned (D(h) at low roughness, the slope equations, log(1 - t w)); a few points
in 2^18 on the other side of a branch threshold.  The gates are ~3 x what was measured (the numbers in the comments)."""
import numpy as np

import cases
import oracle_lib as O
import twin64
from gpu_util_cpu import ggx_oracle

N = 1 << 18


def _stats(got, ref):
    e = cases.rel_err(np.asarray(got, np.float32), ref)
    e = np.where(np.isfinite(e), e, np.inf)
    return dict(median=float(np.median(e)), p99=float(np.quantile(e, 0.99)), gt5=float((e > 1e-5).mean()),
                gt3=float((e > 1e-3).mean()), max=float(e.max()))


def test_ggx_closure_against_the_float64_twin():
    c = cases.ggx_mixed(cases.SEED_PARITY, N)
    x = cases.xi(cases.SEED_PARITY, N, 4)
    og = ggx_oracle(O, c, nthreads=4)
    tw = twin64.Ggx64(c)
    m_ref = og.microfacet(x[0], x[1])
    st = _stats(tw.microfacet(x[0], x[1]), m_ref)            # median 8e-8, > 1e-5: 6.8e-4, > 1e-3: 3.8e-6 (threshold flips)
    print("microfacet", st)
    assert st["median"] <= 3e-7 and st["gt5"] <= 2e-3 and st["gt3"] <= 2e-5, st
    wi, f, pdf, fres = og.sample_eval_pdf(x[0], x[1])
    w64, m64 = wi.astype(np.float64), m_ref.astype(np.float64)
    for name, got, ref in (("evalBrdf on the oracle's wi", tw.eval(w64), f), ("evalPdf on the oracle's wi", tw.pdf(w64), pdf)):
        st = _stats(got, ref)                               # median 2.7e-7, p99 8.9e-6, max 6e-4 (D(h) at low roughness)
        print(name, st)
        assert st["median"] <= 1e-6 and st["p99"] <= 3e-5 and st["gt3"] == 0.0 and st["max"] <= 2e-3, (name, st)
    st = _stats(tw.fresnel(w64, m64), fres)                  # max 3.5e-6
    assert st["max"] <= 1e-5, ("fresnel", st)
    st = _stats(tw.reflect(m64), wi)                         # max 2.7e-7
    assert st["max"] <= 1e-6, ("reflectDirection", st)
    r = og.reflect_refract(x[0], x[1], x[2], x[3])
    m2 = og.microfacet(x[2], x[3]).astype(np.float64)
    st = _stats(tw.sample_weight(r[4].astype(np.float64), m2), r[5])       # median 9e-8, > 1e-5: 1.7e-4, max 2.7e-4
    print("getSampleWeight", st)
    assert st["median"] <= 3e-7 and st["gt5"] <= 1e-3 and st["max"] <= 1e-3, st


def test_ggx_presets_against_the_float64_twin():
    """the reference's own regression presets (testsuite/mtoa/0001..0003): ior < 1 (gold), full anisotropy"""
    wo, Nn, T = cases.frame(cases.SEED_EDGE, 1 << 14)
    x = cases.xi(cases.SEED_EDGE, 1 << 14, 2)
    for name, p in cases.GGX_PRESETS.items():
        c = dict(wo=wo, N=Nn, T=T, **p)
        og = ggx_oracle(O, c, nthreads=4)
        tw = twin64.Ggx64(c)
        st = _stats(tw.microfacet(x[0], x[1]), og.microfacet(x[0], x[1]))
        assert st["median"] <= 3e-7 and st["gt5"] <= 5e-3 and st["gt3"] <= 2e-4, (name, st)
        wi, f, pdf, fres = og.sample_eval_pdf(x[0], x[1])
        w64 = wi.astype(np.float64)
        for what, got, ref in (("eval", tw.eval(w64), f), ("pdf", tw.pdf(w64), pdf)):
            st = _stats(got, ref)
            assert st["median"] <= 1e-6 and st["p99"] <= 1e-4 and st["max"] <= 5e-3, (name, what, st)


def test_nd_profile_against_the_float64_twin():
    s = cases.sss_mixed(cases.SEED_PARITY, N)
    x = cases.xi(cases.SEED_PARITY, N, 1)
    r, pdf, prof = O.Sss(N, s["dist"], s["albedo"], nthreads=4).nd_sample(x[0])
    nd = twin64.NdProfile64(s["dist"])
    st = _stats(nd.radius(x[0]), r)                          # median 1e-7, > 1e-5: 2.5e-3 (log(1 - t w) near t w = 0), > 1e-3: 1.9e-5
    print("getRadius", st)
    assert st["median"] <= 3e-7 and st["gt5"] <= 8e-3 and st["gt3"] <= 1e-4, st
    r64 = r.astype(np.float64)
    assert _stats(nd.pdf(r64), pdf)["max"] <= 2e-6           # max 3.5e-7
    assert _stats(nd.profile(r64), prof)["max"] <= 2e-6      # max 2.5e-7


def test_disney_closure_against_the_float64_twin():
    """evalBrdf / evalPdf of both lobes on the oracle's own sampled directions (src/rlDisney.cpp:199-236, 318-357, 515-577)"""
    from gpu_util_cpu import disney_oracle
    c = cases.disney_mixed(cases.SEED_PARITY, N)
    x = cases.xi(cases.SEED_PARITY, N, 2)
    od = disney_oracle(O, c, nthreads=4)
    tw = twin64.Disney64(c)
    wi, f, pdf = od.sample_eval_pdf(0x08, x[0], x[1])               # AI_RAY_DIFFUSE
    w = wi.astype(np.float64)
    for what, got, ref in (("diffuse eval", tw.eval_diffuse(w), f), ("diffuse pdf", tw.pdf_diffuse(w), pdf)):
        st = _stats(got, ref)                                       # max 6e-6
        assert st["max"] <= 2e-5, (what, st)
    wi, f, pdf = od.sample_eval_pdf(0x10, x[0], x[1])               # AI_RAY_GLOSSY
    ok = ~(wi == 0).all(axis=0)
    w = wi.astype(np.float64)
    # D_GTR2Aniso at alpha = 0.01 and D_GTR1 at alpha = 0.001 amplify the half vector's fp32 rounding
    for what, got, ref in (("glossy eval", tw.eval_specular(w)[:, ok], f[:, ok]), ("glossy pdf", tw.pdf_specular(w)[ok], pdf[ok])):
        st = _stats(got, ref)                                       # median 2.4e-7 / 2.8e-7, p99 9e-6 / 2e-5, > 1e-3: 6e-5 / 1.5e-4, max 8e-3 / 1e-2
        print(what, st)
        assert st["median"] <= 1e-6 and st["p99"] <= 6e-5 and st["gt3"] <= 5e-4 and st["max"] <= 3e-2, (what, st)
