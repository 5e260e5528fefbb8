"""BASELINE.json configs[0]: rlGgx reflect eval+sample on 2^20 (wo, N, alpha = 0.3) tuples.
CPU leg: the oracle alone (plumbing: batch drivers, threading, generator) with physical sanity checks.
GPU leg: the same 2^20 tuples through the C ABI, every output within 1e-5 of the oracle on every point."""
import time

import numpy as np
import pytest

import cases

N = 1 << 20


def _case():
    c = cases.ggx_alpha03(cases.SEED_THROUGHPUT, N)
    x = cases.xi(cases.SEED_THROUGHPUT, N, 2)
    return c, x


def test_config1_cpu_closure_path(oracle):
    c, x = _case()
    g = oracle.Ggx(c["wo"], c["N"], c["T"], KsColor=c["KsColor"], ior=c["ior"], roughness=c["roughness"],
                   anisotropic=c["anisotropic"], nthreads=oracle.hardware_threads())
    t0 = time.perf_counter()
    wi, f, pdf, F = g.sample_eval_pdf(x[0], x[1])
    dt = time.perf_counter() - t0
    print(f"config 1: {N / dt / 1e6:.2f} Msamples/s on {oracle.hardware_threads()} threads")
    assert np.isfinite(wi).all() and np.isfinite(f).all() and np.isfinite(pdf).all()
    assert (pdf >= 1e-4).all() and ((F >= 0) & (F <= 1)).all()
    # alpha = roughness^2 = 0.3 exactly as configured (src/rlGgx.h:149)
    assert abs(float(np.float32(c["roughness"]) ** 2) - 0.3) < 1e-7
    # VNDF importance sampling: the weight f/pdf is Fresnel x G1(L) <= 1 (Heitz & d'Eon), and its mean --
    # the directional albedo of a white dielectric GGX lobe averaged over the view distribution -- is
    # a few percent for ior 1.5
    w = f[0] / pdf
    up = (wi * c["N"]).sum(axis=0) > 0
    assert (w[up] <= 1.0 + 1e-4).all() and (w[up] >= 0).all()
    assert 0.02 < float(w.mean()) < 0.25
    # splitting the batch across threads does not change a bit
    g1 = oracle.Ggx(c["wo"], c["N"], c["T"], KsColor=c["KsColor"], ior=c["ior"], roughness=c["roughness"],
                    anisotropic=c["anisotropic"], nthreads=1)
    m = 1 << 16
    wi1, f1, pdf1, F1 = oracle.Ggx(c["wo"][:, :m].copy(), c["N"][:, :m].copy(), c["T"][:, :m].copy(),
                                   KsColor=c["KsColor"], ior=c["ior"], roughness=c["roughness"],
                                   anisotropic=c["anisotropic"], nthreads=1).sample_eval_pdf(x[0][:m], x[1][:m])
    assert np.array_equal(wi1, wi[:, :m]) and np.array_equal(f1, f[:, :m]) and np.array_equal(pdf1, pdf[:m])


@pytest.mark.gpu
def test_config1_gpu_matches_cpu(gpu, oracle):
    from gpu_util import dev, ggx_oracle, ggx_sampler, host
    c, x = _case()
    ref = ggx_oracle(oracle, c, nthreads=oracle.hardware_threads()).sample_eval_pdf(x[0], x[1])
    s = ggx_sampler(gpu, c)
    got = [host(t) for t in s.sampleEvalPdf(dev(x[0]), dev(x[1]))]
    words = 0
    for nm, a, b in zip(("wi", "f", "pdf", "fresnel"), got, ref):
        st = cases.summarize(cases.rel_err(a, b))
        words += int((a.view(np.uint32) != b.view(np.uint32)).sum())
        print("config 1", nm, st)
        cases.assert_tight(st, ("config 1", nm))
    print("config 1: words differing from the CPU closures:", words)
    # separate verbs: eval and pdf of the CPU's own directions
    assert np.array_equal(host(s.evalBrdf(dev(ref[0]))).view(np.uint32), ref[1].view(np.uint32))
    assert np.array_equal(host(s.evalPdf(dev(ref[0]))).view(np.uint32), ref[2].view(np.uint32))
