"""GPU parity of the rlDisney closure kernels against the CPU oracle, through the C ABI.
EXACT mode restates the host libm's powf / logf (and the angle functions), so decoupled eval/pdf,
sampled directions, the fused chain and the n^2-spp sums all reproduce the oracle (cases.assert_tight)."""
import numpy as np
import pytest

import cases
import rlshaders_amd as R
from gpu_util import dev, disney_oracle, disney_sampler, host

pytestmark = pytest.mark.gpu

N = 1 << 16
TOL = 1e-5
LOBES = ((R.RLS_RAY_DIFFUSE, "diffuse"), (R.RLS_RAY_GLOSSY, "glossy"))


@pytest.fixture(scope="module")
def mixed(oracle):
    return cases.disney_mixed(cases.SEED_PARITY, N), cases.xi(cases.SEED_PARITY, N, 2)


@pytest.mark.parametrize("lobe,name", LOBES)
def test_eval_pdf_decoupled(gpu, oracle, mixed, lobe, name):
    c, x = mixed
    od = disney_oracle(oracle, c)
    wi = od.sample(lobe, x[0], x[1])
    s = disney_sampler(gpu, c)
    s.setSampleType(lobe)
    ef = cases.summarize(cases.rel_err(host(s.evalBrdf(dev(wi))), od.eval(lobe, wi)))
    ep = cases.summarize(cases.rel_err(host(s.evalPdf(dev(wi))), od.pdf(lobe, wi)))
    print("disney", name, "eval decoupled", ef)
    print("disney", name, "pdf  decoupled", ep)
    assert ef["nonfinite"] == 0 and ep["nonfinite"] == 0
    assert ef["max"] <= TOL and ep["max"] <= TOL


@pytest.mark.parametrize("lobe,name", LOBES)
def test_sample_and_fused(gpu, oracle, mixed, lobe, name):
    c, x = mixed
    od = disney_oracle(oracle, c)
    ref = od.sample_eval_pdf(lobe, x[0], x[1])
    s = disney_sampler(gpu, c)
    s.setSampleType(lobe)
    got = [host(t) for t in s.sampleEvalPdf(dev(x[0]), dev(x[1]))]
    wi = host(s.evalSample(dev(x[0]), dev(x[1])))
    assert np.array_equal(wi.view(np.uint32), got[0].view(np.uint32))
    # invalid samples (zero vector, src/rlDisney.cpp:385-387) must be flagged identically
    z_ref = (ref[0] == 0).all(axis=0)
    z_got = (got[0] == 0).all(axis=0)
    assert (z_ref != z_got).sum() <= cases.flag_slack()          # none under strict parity
    both = ~z_ref & ~z_got
    for k, nm in enumerate(("wi", "f", "pdf")):
        st = cases.summarize(cases.rel_err(got[k][..., both], ref[k][..., both]))
        print("disney", name, "fused", nm, st)
        cases.assert_tight(st, (name, nm))
    # zero sample -> black, pdf 0
    assert np.all(got[1][:, z_got] == 0) and np.all(got[2][z_got] == 0)


@pytest.mark.parametrize("preset", sorted(cases.DISNEY_PRESETS))
def test_testsuite_presets(gpu, oracle, preset):
    n = 1 << 14
    wo, N, T = cases.frame(cases.SEED_PARITY, n)
    c = dict(wo=wo, N=N, T=T, **cases.DISNEY_PRESETS[preset])
    x = cases.xi(cases.SEED_PARITY, n, 2)
    od = disney_oracle(oracle, c)
    s = disney_sampler(gpu, c)
    for lobe, name in LOBES:
        s.setSampleType(lobe)
        ref = od.sample_eval_pdf(lobe, x[0], x[1])
        got = [host(t) for t in s.sampleEvalPdf(dev(x[0]), dev(x[1]))]
        for k, nm in enumerate(("wi", "f", "pdf")):
            st = cases.summarize(cases.rel_err(got[k], ref[k]))
            print(preset, name, nm, st)
            cases.assert_tight(st, (preset, name, nm))
        f = host(s.evalBrdf(dev(ref[0])))
        assert cases.summarize(cases.rel_err(f, ref[1]))["max"] <= TOL


def test_zero_indir_and_grazing(gpu, oracle, mixed):
    c, x = mixed
    s = disney_sampler(gpu, c)
    z = dev(np.zeros((3, N), np.float32))
    for lobe, _ in LOBES:
        s.setSampleType(lobe)
        assert np.all(host(s.evalBrdf(z)) == 0)
        assert np.all(host(s.evalPdf(z)) == 0)
    # directions below the horizon: black (src/rlDisney.cpp:204,323), diffuse pdf floored at 1e-4
    wi = (-c["N"]).astype(np.float32)
    od = disney_oracle(oracle, c)
    for lobe, _ in LOBES:
        s.setSampleType(lobe)
        assert np.all(host(s.evalBrdf(dev(wi))) == 0)
        assert cases.summarize(cases.rel_err(host(s.evalPdf(dev(wi))), od.pdf(lobe, wi)))["max"] <= TOL


def test_integrate_reduced_and_streamed(gpu, oracle):
    """spp_n^2 in-kernel samples per lobe: per-sample stream against the oracle, sums against the
    oracle's sums, every lane-group width against G = 1"""
    import os
    n, spp_n = 1 << 11, 4
    c = cases.disney_mixed(cases.SEED_PARITY, n)
    od = disney_oracle(oracle, c)
    ref = od.integrate(spp_n, 4321, streamed=True)
    s = disney_sampler(gpu, c)
    os.environ["RLS_INTEGRATE_GROUP"] = "1"      # one lane per point: sums in the reference's order
    got = {k: host(v) for k, v in s.integrate(spp_n, 4321, streamed=True).items()}
    st = cases.summarize(cases.rel_err(got["wi"], ref["wi"]))
    print("disney integrate streamed wi", st)
    cases.assert_tight(st, "streamed wi")
    for k in ("diffuse_count", "specular_count"):
        assert (got[k] != ref[k]).sum() <= cases.flag_slack(), k
    for k in ("diffuse_sum", "specular_sum"):
        st = cases.summarize(cases.rel_err(got[k], ref[k]))
        print("disney integrate", k, st)
        cases.assert_tight(st, k)
    base = {k: host(v) for k, v in s.integrate(spp_n, 4321).items()}
    del os.environ["RLS_INTEGRATE_GROUP"]
    for k in ("diffuse_sum", "specular_sum", "diffuse_count", "specular_count"):
        assert np.array_equal(base[k].view(np.uint32), got[k].view(np.uint32)), k   # reduced == streamed sums
    for g in ("4", "16"):
        os.environ["RLS_INTEGRATE_GROUP"] = g
        try:
            alt = {k: host(v) for k, v in s.integrate(spp_n, 4321).items()}
        finally:
            del os.environ["RLS_INTEGRATE_GROUP"]
        for k in ("diffuse_count", "specular_count"):
            assert np.array_equal(alt[k], base[k]), (g, k)
        for k in ("diffuse_sum", "specular_sum"):
            cases.assert_same_bits(alt[k], base[k], (g, k))    # sums in sample order whatever the group width


def test_unselected_alternates(gpu, oracle, mixed):
    """the plain-NDF samplers, the non-VNDF pdf branch and D_GTR2 (src/rlDisney.cpp:406-414,504-512,541-542,
    553-559): compiled by the reference, never selected (191); kept batched for completeness"""
    c, x = mixed
    od = disney_oracle(oracle, c)
    s = disney_sampler(gpu, c)
    for kind in (0, 1):
        st = cases.summarize(cases.rel_err(host(s.altSample(kind, dev(x[0]), dev(x[1]))), od.alt(kind, x[0], x[1])))
        print("disney alt sampler", kind, st)
        cases.assert_tight(st, ("alt sampler", kind))
    wi = od.sample(R.RLS_RAY_GLOSSY, x[0], x[1])
    st = cases.summarize(cases.rel_err(host(s.altPdf(dev(wi))), od.alt(2, v=wi)))
    print("disney alt pdf", st)
    cases.assert_tight(st, "alt pdf")
    st = cases.summarize(cases.rel_err(host(s.dGtr2(dev(wi))), od.alt(3, v=wi)))
    cases.assert_tight(st, "D_GTR2")
