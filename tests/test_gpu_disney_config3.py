"""GPU: BASELINE config 3 as it is stated -- rlDisney, both lobes, 8 x 8 = 64 stratified samples per point and
lobe (the sample loops of src/rlDisney.cpp:240-315) -- against the oracle at a size the oracle finishes in
seconds, through size-independent properties at the full 2^26 points, and in streamed mode walked in chunks."""
import os

import numpy as np
import pytest
import torch

import cases
import rlshaders_amd as R
from gpu_util import dev, disney_oracle, disney_sampler, host

pytestmark = pytest.mark.gpu

SUMS = ("diffuse_sum", "specular_sum")
COUNTS = ("diffuse_count", "specular_count")


def _with_group(g, fn):
    os.environ["RLS_INTEGRATE_GROUP"] = str(g)
    try:
        return fn()
    finally:
        del os.environ["RLS_INTEGRATE_GROUP"]


def test_integrate_8x8_against_oracle(gpu, oracle):
    """64 spp per lobe (LDS table of 2 x 64 words, the G-lane split at spp 64): G = 1 adds in the reference's order
    and must match the oracle bit for bit; G = 4 / 16 / 64 differ by the summation order only."""
    n, spp_n, seed = 1 << 11, 8, 4321
    c = cases.disney_mixed(cases.SEED_PARITY, n)
    ref = disney_oracle(oracle, c).integrate(spp_n, seed)
    s = disney_sampler(gpu, c)
    base = _with_group(1, lambda: {k: host(v) for k, v in s.integrate(spp_n, seed).items()})
    for k in COUNTS:
        assert (base[k] != ref[k]).sum() <= cases.flag_slack(), k
        assert base[k].min() >= 0 and base[k].max() <= spp_n * spp_n
    for k in SUMS:
        st = cases.summarize(cases.rel_err(base[k], ref[k]))
        print("disney 8x8", k, st, "words differing", int((base[k].view(np.uint32) != ref[k].view(np.uint32)).sum()))
        cases.assert_tight(st, k)
    for g in (4, 16, 64):
        alt = _with_group(g, lambda: {k: host(v) for k, v in s.integrate(spp_n, seed).items()})
        for k in COUNTS:
            assert np.array_equal(alt[k], base[k]), (g, k)
        for k in SUMS:
            cases.assert_same_bits(alt[k], base[k], (g, k))    # sums in sample order whatever the group width
    # the automatic width (small batch -> several lanes per point)
    auto = {k: host(v) for k, v in s.integrate(spp_n, seed).items()}
    for k in COUNTS + SUMS:
        cases.assert_same_bits(auto[k], base[k], ("auto", k))


def test_streamed_chunked_equals_unchunked(gpu, oracle):
    """rls_disney_integrate_chunked: ragged last chunk, every chunk handed to the consumer; the samples and the
    sums are those of the unchunked streamed call (and of the oracle)."""
    n, spp_n, seed, first = 1000, 8, 99, (1 << 33) + 17
    spp = spp_n * spp_n
    c = cases.disney_mixed(cases.SEED_PARITY, n)
    s = disney_sampler(gpu, c)
    whole = _with_group(1, lambda: {k: host(v) for k, v in s.integrate(spp_n, seed, streamed=True, first_index=first).items()})
    ref = disney_oracle(oracle, c).integrate(spp_n, seed, streamed=True, first_index=first)
    for k in ("wi", "f", "pdf"):
        cases.assert_tight(cases.summarize(cases.rel_err(whole[k], ref[k])), ("streamed", k))
    got = {k: np.zeros_like(whole[k]) for k in ("wi", "f", "pdf")}
    seen = []

    def consume(p0, count, chunk):
        torch.cuda.synchronize()
        seen.append((p0, count))
        for k in ("wi", "f", "pdf"):
            a = host(chunk[k])[..., : 2 * spp * count]
            a = a.reshape(a.shape[:-1] + (2 * spp, count))
            dst = got[k].reshape(got[k].shape[:-1] + (2 * spp, n))
            dst[..., p0:p0 + count] = a

    sums, _ = _with_group(1, lambda: s.integrateChunked(spp_n, seed, 300, consume=consume, first_index=first))
    assert seen == [(0, 300), (300, 300), (600, 300), (900, 100)]
    for k in ("wi", "f", "pdf"):
        assert np.array_equal(got[k].view(np.uint32), whole[k].view(np.uint32)), k
    for k in SUMS + COUNTS:
        assert np.array_equal(host(sums[k]).view(np.uint32), whole[k].view(np.uint32)), k
    # a consumer that fails stops the walk with an error, not a crash
    def bad(p0, count, chunk):
        raise RuntimeError("consumer failed")
    with pytest.raises(RuntimeError, match="consumer failed"):
        s.integrateChunked(spp_n, seed, 300, consume=bad)
    # ... and the C ABI says so with its own status: RLS_ERR_ABORTED (6), not "invalid argument"
    import ctypes as C
    from rlshaders_amd import _capi as capi
    from rlshaders_amd.closures import plane, rgb, vec3
    spp = spp_n * spp_n
    m = 2 * spp * 300
    ch = dict(wi=gpu.empty(3, m), f=gpu.empty(3, m), pdf=gpu.empty(m))
    so = capi.DisneyStreamOut(vec3(ch["wi"], m, "wi"), rgb(ch["f"], m, "f"), plane(ch["pdf"], m, "pdf"))
    o = {k: gpu.empty(3, n) for k in SUMS}
    o.update({k: gpu.empty(n) for k in COUNTS})
    stop = capi.DisneyChunkFn(lambda user, p0, count, chunk: 7)

    def chunked(ctx, cb):
        return ctx.lib.rls_disney_integrate_chunked(
            ctx.handle, n, C.byref(s.c), spp_n, seed, first, rgb(o["diffuse_sum"], n, "d"), plane(o["diffuse_count"], n, "dc"),
            rgb(o["specular_sum"], n, "s"), plane(o["specular_count"], n, "sc"), 300, C.byref(so), cb, None)
    assert chunked(gpu, stop) == 6 and b"consumer returned 7" in gpu.lib.rls_last_error()
    # a consumer cannot be recorded into a launch graph (a replay would overwrite the chunk buffers without calling it):
    # refused while capturing (RLS_ERR_UNSUPPORTED = 5); without a consumer the chunk launches record fine
    own = R.Context(0, use_torch_stream=False)
    try:
        torch.cuda.synchronize()
        assert own.lib.rls_graph_begin_capture(own.handle) == 0
        assert chunked(own, stop) == 5 and b"launch graph" in own.lib.rls_last_error()
        assert chunked(own, capi.DisneyChunkFn()) == 0
        h = C.c_void_p()
        assert own.lib.rls_graph_end_capture(own.handle, C.byref(h)) == 0
        assert own.lib.rls_graph_launch(own.handle, h) == 0
        own.synchronize()
        own.lib.rls_graph_destroy(h)
        for k in SUMS + COUNTS:                          # the replayed chunk launches produced the batch's sums
            assert np.array_equal(host(o[k]).view(np.uint32), whole[k].view(np.uint32)), k
    finally:
        own.close()
    # no consumer: samples discarded, sums unchanged
    sums2, _ = _with_group(1, lambda: s.integrateChunked(spp_n, seed, 256, first_index=first))
    for k in SUMS + COUNTS:
        assert np.array_equal(host(sums2[k]).view(np.uint32), whole[k].view(np.uint32)), k


def test_full_size_config3(gpu, oracle):
    """2^26 points x 64 spp x 2 lobes in reduced mode: counts, finiteness, lane-width independence, a second launch
    gives the same bits, and oracle spot checks on windows of the very same device-generated inputs."""
    n, spp_n, seed = 1 << 26, 8, 1234
    ctx = gpu
    wo, N, T = R.gen_frame(ctx, seed, 0, n)
    u = lambda stream, lo=0.0, hi=1.0: R.gen_uniform(ctx, seed, 0, n, stream, lo, hi)
    base = torch.stack([u(8 + j) for j in range(3)])
    sc = {k: u(32 + j) for j, k in enumerate(R._capi.DISNEY_SCALARS)}
    d = R.DisneySampler(ctx, wo, N, T, base_color=base, **sc)
    out = _with_group(1, lambda: d.integrate(spp_n, seed))
    spp = spp_n * spp_n
    for k in COUNTS:
        cnt = out[k]
        assert (cnt >= 0).all() and (cnt <= spp).all() and (cnt == cnt.round()).all(), k
    # the cosine lobe is valid wherever the direction has cos > pi * 1e-4: nearly every sample
    assert out["diffuse_count"].mean().item() > 0.99 * spp
    assert out["specular_count"].mean().item() > 0.5 * spp
    for k in SUMS:
        fin = torch.isfinite(out[k]).all(dim=0)
        assert fin.float().mean().item() > 0.99999, k
        assert (out[k][:, fin] >= 0).all(), k
    ck = {k: R.checksum(ctx, out[k]) for k in SUMS + COUNTS}
    again = _with_group(1, lambda: d.integrate(spp_n, seed))
    assert ck == {k: R.checksum(ctx, again[k]) for k in SUMS + COUNTS}
    del again
    alt = _with_group(4, lambda: d.integrate(spp_n, seed))
    for k in COUNTS:
        assert torch.equal(alt[k], out[k]), k
    # four lanes per point: the sums grow in sample order all the same (fold, rls_loops.hpp) -- every word of 2^26 points
    assert ck == {k: R.checksum(ctx, alt[k]) for k in SUMS + COUNTS}
    del alt
    # oracle on eight windows of 256 points; first_index aligns the oracle's scrambles with the batch's
    for w in range(8):
        p0 = w * (n // 8) + 4099 * w
        sl = slice(p0, p0 + 256)
        sub = lambda t: t[..., sl].contiguous().cpu().numpy()
        c = dict(wo=sub(wo), N=sub(N), T=sub(T), base_color=sub(base), **{k: sub(v) for k, v in sc.items()})
        ref = disney_oracle(oracle, c).integrate(spp_n, seed, first_index=p0)
        for k in COUNTS:
            assert (sub(out[k]) != ref[k]).sum() <= cases.flag_slack(), (w, k)
        for k in SUMS:
            cases.assert_tight(cases.summarize(cases.rel_err(sub(out[k]), ref[k])), ("window", w, k))


def test_sharded_integrators_draw_the_batch_numbers(gpu, oracle):
    """first_index: two shards of a batch reproduce the unsplit call bit for bit, for every in-kernel-sampling
    entry point (rls_ggx_integrate, rls_ggx_direct_lighting, rls_disney_integrate, rls_disney_direct_lighting,
    rls_ggx_shade, rls_disney_shade, rls_sss_integrate_scatter)."""
    n, h, spp_n, seed = 4096, 1500, 4, 31
    cut = lambda a, sl: a[..., sl] if isinstance(a, np.ndarray) else a
    parts = (slice(0, h), slice(h, n))

    def run(fn_whole, fn_part):
        whole = _with_group(1, fn_whole)
        pieces = [_with_group(1, lambda sl=sl: fn_part(sl)) for sl in parts]
        for j, w in enumerate(whole):
            joined = np.concatenate([p[j] for p in pieces], axis=-1)
            assert np.array_equal(joined.view(np.uint32), w.view(np.uint32)), j

    # rlGgx integrate + direct lighting
    c = cases.ggx_mixed(cases.SEED_PARITY, n)
    from gpu_util import ggx_sampler
    P = np.stack([oracle.gen_uniform(seed, 0, n, 40 + j, 0.0, 4.0) for j in range(3)])
    lt = R.make_light(center=(2.0, 2.0, 6.0), radius=1.25, radiance=(3.0, 2.0, 1.0))
    mk = lambda sl: ggx_sampler(gpu, {k: cut(v, sl) for k, v in c.items()})
    run(lambda: [host(t) for t in mk(slice(0, n)).integrate(spp_n, seed)],
        lambda sl: [host(t) for t in mk(sl).integrate(spp_n, seed, first_index=sl.start)])
    run(lambda: [host(t) for t in mk(slice(0, n)).directLighting(dev(P), lt, spp_n, seed)],
        lambda sl: [host(t) for t in mk(sl).directLighting(dev(np.ascontiguousarray(P[:, sl])), lt, spp_n, seed,
                                                           first_index=sl.start)])
    # rlDisney
    cd = cases.disney_mixed(cases.SEED_PARITY, n)
    keys = SUMS + COUNTS
    mkd = lambda sl: disney_sampler(gpu, {k: cut(v, sl) for k, v in cd.items()})
    run(lambda: [host(mkd(slice(0, n)).integrate(spp_n, seed)[k]) for k in keys],
        lambda sl: [host(mkd(sl).integrate(spp_n, seed, first_index=sl.start)[k]) for k in keys])
    lts = [lt, R.make_light(center=(-1.0, 5.0, 7.0), radius=0.6, radiance=(0.5, 0.5, 4.0), mis_mode=2)]
    run(lambda: [host(t) for t in mkd(slice(0, n)).directLighting(dev(P), lts, spp_n, seed)],
        lambda sl: [host(t) for t in mkd(sl).directLighting(dev(np.ascontiguousarray(P[:, sl])), lts, spp_n, seed,
                                                            first_index=sl.start)])
    # the whole-node entry points
    keysd = ("direct_diffuse", "direct_specular", "indirect_diffuse", "indirect_specular", "out")
    run(lambda: [host(mkd(slice(0, n)).shade(dev(P), lts, spp_n, seed)[k]) for k in keysd],
        lambda sl: [host(mkd(sl).shade(dev(np.ascontiguousarray(P[:, sl])), lts, spp_n, seed, first_index=sl.start)[k])
                    for k in keysd])
    keysg = ("direct_diffuse", "direct_specular", "refraction", "indirect_diffuse", "indirect_specular", "out")
    run(lambda: [host(mk(slice(0, n)).shade(dev(P), lts, spp_n, seed, Kt=0.5)[k]) for k in keysg],
        lambda sl: [host(mk(sl).shade(dev(np.ascontiguousarray(P[:, sl])), lts, spp_n, seed, Kt=0.5, first_index=sl.start)[k])
                    for k in keysg])
    # rlSss integrateScatter on the unit sphere
    dist = np.stack([oracle.gen_uniform(seed, 0, n, 32 + j, 0.02, 0.3) for j in range(3)])
    scene = R.make_scene("sphere", sphere_radius=1.0, light_dir=(0.0, 0.6, 0.8), use_cavity_fade=True)
    Nn, Tt = c["N"], c["T"]
    mks = lambda sl: R.SssSampler(gpu, dev(np.ascontiguousarray(Nn[:, sl])), dev(np.ascontiguousarray(Tt[:, sl])),
                                  (0.8, 0.5, 0.3), dev(np.ascontiguousarray(dist[:, sl])))
    run(lambda: [host(mks(slice(0, n)).integrateScatter(dev(Nn), scene, spp_n, seed))],
        lambda sl: [host(mks(sl).integrateScatter(dev(np.ascontiguousarray(Nn[:, sl])), scene, spp_n, seed,
                                                  first_index=sl.start))])


def test_packed_rare_branches_with_every_lane_asking(gpu, oracle):
    """the n^2-spp loops queue the rare sampler branches of four samples per wavefront in LDS and evaluate them 64 at a
    time (rls_loops.hpp, SlowLds).  Here every lane asks at every sample -- views along the normal make every visible-
    normal sample take the uniform-slope fallback, and clearcoat = 1 sends a fifth of the samples to the clearcoat lobe
    -- so the queue holds 256 requests per pass of four samples and is worked off in four rounds; also a sample count (9)
    that is not a multiple of four, and G > 1 (the lanes of a group hold different samples)."""
    n, seed = 1 << 11, 77
    wo, N, T = cases.frame(cases.SEED_PARITY, n)
    wo = N.copy()                                                      # theta = 0: nearNormal in every lane
    c = cases.disney_mixed(cases.SEED_PARITY, n)
    c = dict(c, wo=wo, N=N, T=T, clearcoat=np.ones(n, np.float32))
    od, d = disney_oracle(oracle, c), disney_sampler(gpu, c)
    for spp_n in (3, 4):
        ref = od.integrate(spp_n, seed, streamed=True)
        got = _with_group(1, lambda: {k: host(v) for k, v in d.integrate(spp_n, seed, streamed=True).items()})
        for k in ref:
            cases.assert_tight(cases.summarize(cases.rel_err(got[k], ref[k])), ("every lane asks", spp_n, k))
        for g in (4, 16):
            alt = _with_group(g, lambda: {k: host(v) for k, v in d.integrate(spp_n, seed, streamed=True).items()})
            for k in ("wi", "f", "pdf"):                               # the samples themselves do not depend on G
                assert np.array_equal(alt[k].view(np.uint32), got[k].view(np.uint32)), (g, k)
    # the rlGgx loops: the same view, low roughness
    from gpu_util import ggx_oracle, ggx_sampler
    g = cases.ggx_mixed(cases.SEED_PARITY, n)
    g = dict(g, wo=wo, N=N, T=T)
    ref = ggx_oracle(oracle, g).integrate(3, seed)
    got = _with_group(1, lambda: [host(t) for t in ggx_sampler(gpu, g).integrate(3, seed)])
    for nm, a, b in zip(("sum", "avg"), got, ref):
        cases.assert_tight(cases.summarize(cases.rel_err(a, b)), ("ggx every lane asks", nm))


def _border_check(a, b, is_count, what):
    """bit for bit (NaN where the oracle has NaN) under strict parity; on another libm flavour still: the same non-finite
    pattern -- a wrong window guard of the loop reciprocals shows as NaN / inf / garbage, not as a last-bit difference --
    sums within the tight tolerance, counts within flag_slack()"""
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    if cases.strict_parity():
        assert same.all(), (what, int((~same).sum()))
        return
    assert np.array_equal(np.isfinite(a), np.isfinite(b)) and np.array_equal(np.isnan(a), np.isnan(b)), what
    fin = np.isfinite(b)
    if is_count:
        assert int((a[fin] != b[fin]).sum()) <= cases.flag_slack() * 64, (what, int((a[fin] != b[fin]).sum()))
    else:
        cases.assert_tight(cases.summarize(cases.rel_err(a[..., fin] if a.ndim == 1 else a[:, fin.all(axis=0)],
                                                         b[..., fin] if b.ndim == 1 else b[:, fin.all(axis=0)])), what)


def test_loop_reciprocals_at_their_window_borders(gpu, oracle):
    """The sample loops divide by alpha_x, alpha_y, the lobe weight 1 / (clearcoat + 1), its complement and G1 through
    per-point reciprocals when those lie inside rlm::div32_y's window, and the IEEE way for the whole wavefront when a lane's
    do not (the per-point reciprocals of round 4: D_GTR2Aniso's, the lobe weight's, G1's).  Parameter sets on either side of every border, in one batch (so that
    wavefronts mix both kinds) and as uniform values: bit for bit the oracle's sums and counts."""
    n, spp_n, seed = 1 << 12, 4, 31
    base = cases.disney_mixed(cases.SEED_PARITY, n)
    k = np.arange(n) % 8
    c = dict(base)
    # clearcoat: 0 (complement of the weight is 0), 1e-6 and 2e-4 (complement below / near 2^-14), ordinary
    c["clearcoat"] = np.choose(k % 4, [np.zeros(n, np.float32), np.full(n, 1e-6, np.float32), np.full(n, 2.4e-4, np.float32),
                                       base["clearcoat"]]).astype(np.float32)
    # roughness: 0 and 1e-3 (alpha floored at 1e-2), ordinary, 130 (alpha = 16900 > 2^14: no reciprocal), 1e20 (alpha infinite)
    c["roughness"] = np.choose(k, [np.zeros(n, np.float32), np.full(n, 1e-3, np.float32), base["roughness"], base["roughness"],
                                   np.full(n, 130.0, np.float32), base["roughness"], np.full(n, 1e20, np.float32),
                                   base["roughness"]]).astype(np.float32)
    # grazing views: the stretched view's G1 falls below 2^-14 for alpha tan(theta) > 2^15
    wo = base["wo"].copy()
    graze = (k == 5)
    t = (base["T"] + 1e-6 * base["N"]).astype(np.float32)
    wo[:, graze] = (t / np.linalg.norm(t, axis=0, keepdims=True))[:, graze].astype(np.float32)
    c["wo"] = wo
    ref = disney_oracle(oracle, c).integrate(spp_n, seed)
    got = _with_group(1, lambda: {kk: host(v) for kk, v in disney_sampler(gpu, c).integrate(spp_n, seed).items()})
    for kk in SUMS + COUNTS:
        _border_check(got[kk], ref[kk], kk in COUNTS, (kk,))
    # the same borders as UNIFORM values (the hoisted path keeps the reciprocals in scalar registers)
    for cc, rr in ((0.0, 0.4), (1e-6, 0.4), (2.4e-4, 0.0), (0.7, 130.0), (0.3, 1e-3)):
        cu = dict(base, clearcoat=np.float32(cc), roughness=np.float32(rr))
        cu_o = dict(base, clearcoat=np.full(n, cc, np.float32), roughness=np.full(n, rr, np.float32))
        ref = disney_oracle(oracle, cu_o).integrate(spp_n, seed)
        got = _with_group(1, lambda: {kk: host(v) for kk, v in disney_sampler(gpu, cu).integrate(spp_n, seed).items()})
        for kk in SUMS + COUNTS:
            _border_check(got[kk], ref[kk], kk in COUNTS, (cc, rr, kk))
