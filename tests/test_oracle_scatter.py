"""Oracle checks for SssSampler::integrateScatter over an analytic scene (src/rlSss.h:167-280,
293-356, 361-424, 439-454).  The closed renderer services are stand-ins (parity unpinned), so the
anchors are closed forms: the normalized-diffusion profile integrates to one over the plane, the
probe estimator with its three-axis MIS pdf is unbiased, and a light/shadow edge through the
shading point halves the result."""
import math

import numpy as np
import pytest

import oracle_lib as O


def test_scene_trace_plane_and_sphere():
    pl = O.make_scene("plane", plane_point=(0, 0, 1), plane_normal=(0, 0, 1))
    k, t, hp, hn = O.scene_trace(pl, (0.5, 0.25, 3.0), (0, 0, -1), 10.0)
    assert k == 1 and t[0] == 2.0 and hp[0] == (0.5, 0.25, 1.0) and hn[0] == (0.0, 0.0, 1.0)
    assert O.scene_trace(pl, (0, 0, 3.0), (0, 0, -1), 1.5)[0] == 0           # beyond maxdist
    assert O.scene_trace(pl, (0, 0, 3.0), (1, 0, 0), 10.0)[0] == 0            # parallel
    assert O.scene_trace(pl, (0, 0, 3.0), (0, 0, 1), 10.0)[0] == 0            # behind the origin
    sp = O.make_scene("sphere", sphere_center=(0, 0, 0), sphere_radius=2.0)
    k, t, hp, hn = O.scene_trace(sp, (0, 0, 5.0), (0, 0, -1), 10.0)
    assert k == 2 and t == [3.0, 7.0] and hn[0] == (0.0, 0.0, 1.0) and hn[1] == (0.0, 0.0, -1.0)
    k, t, _, _ = O.scene_trace(sp, (0, 0, 5.0), (0, 0, -1), 4.0)
    assert k == 1 and t == [3.0]                                               # second root beyond maxdist
    k, t, _, _ = O.scene_trace(sp, (0, 0, 0), (1, 0, 0), 10.0)                 # from inside: one root ahead
    assert k == 1 and t == [2.0]
    assert O.scene_trace(sp, (0, 3.0, 5.0), (0, 0, -1), 10.0)[0] == 0         # misses


def _frames(n, rng):
    """points on the plane z = 0 with random tangents"""
    ang = rng.uniform(0, 2 * math.pi, n)
    N = np.zeros((3, n), np.float32); N[2] = 1
    T = np.stack([np.cos(ang), np.sin(ang), np.zeros(n)]).astype(np.float32)
    P = np.zeros((3, n), np.float32)
    P[0], P[1] = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    return N, T, P


def _plane_expectation(dist, albedo, light):
    rmax = 3 * max(dist)
    return [a * e / math.pi * (1 - (math.exp(-rmax / d) + 3 * math.exp(-rmax / (3 * d))) / 4)
            for d, a, e in zip(dist, albedo, light)]


@pytest.mark.parametrize("has_dPdu", [True, False])
def test_plane_uniform_light_matches_the_profile_integral(has_dPdu):
    rng = np.random.default_rng(5)
    n = 4096
    N, T, P = _frames(n, rng)
    dist, albedo, light = (0.05, 0.1, 0.2), (0.8, 0.5, 0.3), (2.0, 1.0, 0.5)
    s = O.Sss(n, dist, albedo=albedo, N=N, T=T, has_dPdu=has_dPdu, nthreads=O.hardware_threads())
    sc = O.make_scene("plane", light_dir=(0, 0, 1), light_color=light)
    res, depth = O.integrate_scatter(s, P, sc, 4, 77)
    want = _plane_expectation(dist, albedo, light)
    got = res.astype(np.float64).mean(axis=1)
    np.testing.assert_allclose(got, want, rtol=0.02)
    # only the probes along the normal (half of them) can hit a plane, once each
    assert abs(depth.mean() - 0.5) < 0.01 and depth.max() <= 1.0
    assert np.isfinite(res).all() and (res >= 0).all()


def test_light_edge_halves_the_result_and_decays_into_the_shadow():
    rng = np.random.default_rng(6)
    n = 8192
    N, T, P = _frames(n, rng)
    dist, albedo = (0.1, 0.1, 0.1), (1.0, 1.0, 1.0)
    s = O.Sss(n, dist, albedo=albedo, N=N, T=T, nthreads=O.hardware_threads())
    full = _plane_expectation(dist, albedo, (1, 1, 1))[0]
    means = []
    for x in (-0.6, -0.1, 0.0, 0.1, 0.6):          # signed distance of the shading points from the edge
        Px = P.copy(); Px[0] = x
        sc = O.make_scene("plane", gate_point=(0, 0, 0), gate_normal=(1, 0, 0))
        means.append(O.integrate_scatter(s, Px, sc, 4, 3)[0].astype(np.float64).mean())
    assert means[0] == 0.0                            # 0.6 > maxRadius = 0.3: nothing reaches the point
    assert means[0] < means[1] < means[2] < means[3] < means[4]
    assert abs(means[2] / full - 0.5) < 0.03
    assert abs(means[4] / full - 1.0) < 0.03
    assert abs((means[1] + means[3]) / full - 1.0) < 0.03      # the lit and the shadowed side are complementary


def test_sphere_probe_depth_cavity_fade_and_literal_matrix():
    rng = np.random.default_rng(7)
    n = 2048
    # points on a sphere of radius 1 about the origin, light from +z
    v = rng.normal(size=(3, n)); v /= np.linalg.norm(v, axis=0)
    N = v.astype(np.float32); P = N.copy()
    t = np.cross(N.T, rng.normal(size=(n, 3))).T; t /= np.linalg.norm(t, axis=0)
    T = t.astype(np.float32)
    s = O.Sss(n, (0.1, 0.2, 0.3), albedo=(0.9, 0.6, 0.4), N=N, T=T, nthreads=O.hardware_threads())
    base = O.make_scene("sphere", sphere_radius=1.0, light_dir=(0, 0, 1))
    res, depth = O.integrate_scatter(s, P, base, 4, 11)
    assert np.isfinite(res).all() and (res >= 0).all()
    assert depth.max() <= 2.0 and depth.mean() > 0.5           # tangential probes hit a sphere too, some twice
    lit = N[2] > 0.5
    assert res[:, lit].mean() > 10 * res[:, N[2] < -0.95].mean()
    fade = O.make_scene("sphere", sphere_radius=1.0, light_dir=(0, 0, 1), use_cavity_fade=True)
    rf, _ = O.integrate_scatter(s, P, fade, 4, 11)
    assert (rf <= res * (1 + 1e-6)).all() and rf.sum() < res.sum()   # fade <= 1 on a convex body
    lit_m = O.make_scene("sphere", sphere_radius=1.0, light_dir=(0, 0, 1), literal_matrix=True)
    rl, _ = O.integrate_scatter(s, P, lit_m, 4, 11)
    assert not np.array_equal(rl, res)                                # the two readings of AiM4VectorByMatrixMult differ
