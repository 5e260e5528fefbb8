"""GPU: BASELINE configs 4 and 5 at the size ONE GPU of the 8-GPU job holds -- rlSss probe sampling on 2^25 points
(2^28 / 8) and rlSkin on 2^27 points (2^30 / 8, 31.7 GB of planes) -- through size-independent properties and oracle
windows on the very same device-generated inputs (the shard starts at the index rank 5 of 8 would own)."""
import numpy as np
import pytest
import torch

import cases
import rlshaders_amd as R

pytestmark = pytest.mark.gpu


def _windows(n, k=6, width=192):
    return [slice(p0, p0 + width) for p0 in (j * (n // k) + 4099 * j for j in range(k))]


def test_config4_shard_sss_probe(gpu, oracle):
    n, seed = 1 << 25, 1234
    first = 5 * n                                             # rank 5 of 8: global indices [5n, 6n)
    _, N, T = R.gen_frame(gpu, seed, first, n)
    u = lambda stream, lo=0.0, hi=1.0: R.gen_uniform(gpu, seed, first, n, stream, lo, hi)
    dist = torch.stack([u(32 + j, 0.1, 2.1) for j in range(3)])
    albedo = torch.stack([u(8 + j) for j in range(3)])
    xi = [u(11 + j) for j in range(2)]
    s = R.SssSampler(gpu, N, T, albedo, dist)
    out = s.getProbeRay(xi[0], xi[1])
    maxR = 3.0 * dist.max(dim=0).values
    assert (out["r"] >= 0).all() and (out["r"] <= maxR * (1 + 1e-5)).all()           # inside maxR = 3 max(d)
    pos = out["r"] > 1e-4                                     # r = 0 (xi = 0: two of 2^25 points) has pdf = x / 0, as the reference
    assert torch.isfinite(out["pdf"][pos]).all() and (out["pdf"][pos] > 0).all()
    assert (out["profile"] >= 0).all() and torch.isfinite(out["profile"]).all()
    assert torch.allclose(torch.linalg.vector_norm(out["dir"], dim=0), torch.ones(n, device="cuda"), atol=1e-5)
    # the ray starts on the sphere of radius maxR around the shading point and is 2 offset.y long (src/rlSss.h:511-517)
    # (a radius that rounds past maxR makes sqrt(maxR^2 - r^2) a NaN, in the reference too: a handful of 2^25 points)
    nr = torch.linalg.vector_norm(out["origin"], dim=0)
    ok = torch.isfinite(nr)
    assert (~ok).float().mean().item() < 1e-5
    assert ((nr[ok] - maxR[ok]).abs() <= 2e-5 * maxR[ok]).all()
    assert (out["maxdist"][ok] >= 0).all() and (out["maxdist"][ok] <= 2 * maxR[ok] * (1 + 1e-5)).all()
    ck = {k: R.checksum(gpu, v) for k, v in out.items()}
    again = s.getProbeRay(xi[0], xi[1])
    assert ck == {k: R.checksum(gpu, v) for k, v in again.items()}
    for sl in _windows(n):
        sub = lambda t: t[..., sl].contiguous().cpu().numpy()
        # the oracle regenerates the window's inputs from the global indices: generator and kernels agree on them
        o = oracle.Sss(sl.stop - sl.start, sub(dist), sub(albedo), N=sub(N), T=sub(T), nthreads=4)
        ref = o.probe(sub(xi[0]), sub(xi[1]))
        assert np.array_equal(oracle.gen_uniform(seed, first + sl.start, sl.stop - sl.start, 11), sub(xi[0]))
        for k in ("r", "origin", "dir", "maxdist", "pdf", "profile"):
            cases.assert_tight(cases.summarize(cases.rel_err(sub(out[k]), ref[k])), ("config 4 window", sl.start, k))


def test_config5_shard_skin(gpu, oracle):
    n, seed = 1 << 27, 1234
    first = 5 * n
    free, _ = torch.cuda.mem_get_info()
    assert free > 40 << 30, "the config-5 shard needs ~32 GB of planes"
    wo, N, T = R.gen_frame(gpu, seed, first, n)
    u = lambda k, lo=0.0, hi=1.0: R.gen_uniform(gpu, seed, first, n, 32 + k, lo, hi)
    u3 = lambda k, lo=0.0, hi=1.0: torch.stack([u(k + j, lo, hi) for j in range(3)])
    p = dict(sss_color=u3(0), sss_weight=u(3), sss_dist_multiplier=u(4, 0.5, 1.5), sss_scatter_dist=u3(5, 0.1, 2.1),
             specular_color=u3(8), specular_weight=u(11), specular_roughness=u(12, 0.05, 1.0), specular_ior=u(13, 1.05, 2.55),
             sheen_color=u3(14), sheen_weight=u(17), sheen_roughness=u(18, 0.05, 1.0), sheen_ior=u(19, 1.05, 2.55))
    xi = torch.stack([R.gen_uniform(gpu, seed, first, n, 11 + j) for j in range(6)])
    sk = R.SkinShader(gpu, wo, N, T, **p)
    out = sk.sampleEvalPdf(xi)
    for k in ("sheenFresnel", "specularFresnel", "sssWeight"):
        assert torch.isfinite(out[k]).all() and (out[k] >= 0).all() and (out[k] <= 1 + 1e-6).all(), k
    # layer arithmetic holds point by point (src/rlSkin.cpp:238)
    want = p["sss_weight"] * (1.0 - out["specularFresnel"] * (1.0 - out["sheenFresnel"]))
    assert torch.equal(want, out["sssWeight"])
    for k in ("sheen_pdf", "spec_pdf"):
        on = p["sheen_weight" if k == "sheen_pdf" else "specular_weight"] > 1e-4
        assert (out[k][on] >= 1e-4).all()                                     # pdf floor, src/rlGgx.h:79
    assert torch.isfinite(out["profile"]).all() and (out["r"] >= 0).all()
    ck = {k: R.checksum(gpu, v) for k, v in out.items()}
    again = sk.sampleEvalPdf(xi)
    assert ck == {k: R.checksum(gpu, v) for k, v in again.items()}
    del again
    for sl in _windows(n, k=4, width=128):
        sub = lambda t: t[..., sl].contiguous().cpu().numpy()
        ref = oracle.skin(sub(wo), sub(N), sub(T), {k: sub(v) for k, v in p.items()}, sub(xi), nthreads=4)
        for k in ref:
            cases.assert_tight(cases.summarize(cases.rel_err(sub(out[k]), ref[k])), ("config 5 window", sl.start, k))
