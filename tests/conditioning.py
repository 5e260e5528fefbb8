"""SURVEY.md 8(c) protocol (3): how far the ORACLE's own outputs move when its inputs move by one ulp.

The oracle is a step function of its inputs (a dot product of three rounded products need not change at all under one
component's ulp), so the 1-ulp neighbourhood of an input point is sampled two ways: the 38 axis nudges (each of the 19 input
scalars of config 2 -- wo3 N3 T3 KsColor3 roughness ior anisotropic xi4 -- by +1 and by -1 ulp, the others fixed) and
`corners` random corners of the box (every scalar moved by -1, 0 or +1 ulp at once).  sens[k] = the largest relative
movement of output k over all of them: a lower bound of the reference's local variation, tighter the more corners are drawn.
Used by tests/test_gpu_fast_mode.py and tools/fast_conditioning.py."""
import numpy as np

import cases

NAMES = ("wi", "f", "pdf", "fresnel", "wt", "weight")
KEYS = [("wo", j) for j in range(3)] + [("N", j) for j in range(3)] + [("T", j) for j in range(3)] + \
       [("KsColor", j) for j in range(3)] + [("roughness", None), ("ior", None), ("anisotropic", None)] + \
       [("xi", j) for j in range(4)]
KEY_NAMES = [f"{a}{'' if j is None else j}" for a, j in KEYS]
_ONE = np.nextafter(np.float32(1), np.float32(0))


def _nudge(v, step):
    up = np.nextafter(v, np.float32(np.inf)).astype(np.float32)
    dn = np.nextafter(v, np.float32(-np.inf)).astype(np.float32)
    return np.where(step > 0, up, np.where(step < 0, dn, v)).astype(np.float32)


def _moved(c, x, steps):
    """case dict + xi with every input scalar k moved by steps[k] (an int array per point, or a scalar) ulps of -1 / 0 / +1"""
    c2 = dict(c)
    x2 = x.copy()
    for (name, j), st in zip(KEYS, steps):
        if st is None:
            continue
        if name == "xi":
            x2[j] = np.clip(_nudge(x[j], st), 0, _ONE).astype(np.float32)
        elif j is None:
            c2[name] = _nudge(np.asarray(c[name], np.float32), st)
        else:
            if c2[name] is c[name]:
                c2[name] = np.array(c[name], np.float32, copy=True)
            c2[name][j] = _nudge(c[name][j], st)
    return c2, x2


def subset(c, x, idx):
    cs = {k: (np.ascontiguousarray(v[..., idx]) if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    for k in ("KsColor", "roughness", "ior", "anisotropic"):            # uniform parameters -> planes, so that they can move
        if not isinstance(cs[k], np.ndarray):
            v = np.asarray(cs[k], np.float32)
            cs[k] = np.ascontiguousarray(np.repeat(v[:, None], idx.size, axis=1) if v.ndim else np.full(idx.size, v, np.float32))
    return cs, np.ascontiguousarray(x[:, idx])


def ggx_chain_sensitivity(make_oracle, c, x, base, corners=96, seed=0, axis=True, sens=None, which=None):
    """c, x: the case / random numbers of the points in question (already a subset); base: the oracle's outputs there
    (reflect_refract's six).  -> (sens, which): per output, the largest relative movement over the nudges and the index
    (into KEYS) of the axis nudge that caused it (-1: a corner, or none)."""
    m = x.shape[1]
    if sens is None:
        sens = [np.zeros(m, np.float64) for _ in NAMES]
        which = [np.full(m, -1, np.int32) for _ in NAMES]

    def take(c2, x2, ki):
        p = make_oracle(c2).reflect_refract(x2[0], x2[1], x2[2], x2[3])
        for k in range(len(NAMES)):
            e = cases.rel_err(p[k], base[k]).astype(np.float64)
            same = (p[k].view(np.uint32) == base[k].view(np.uint32)) | (np.isnan(p[k]) & np.isnan(base[k]))
            e = np.where(same.all(axis=0) if same.ndim == 2 else same, 0.0, e)
            e = np.where(np.isfinite(e), e, np.inf)
            upd = e > sens[k]
            sens[k] = np.where(upd, e, sens[k])
            which[k] = np.where(upd, ki, which[k])

    if axis:
        for ki in range(len(KEYS)):
            for d in (1, -1):
                steps = [None] * len(KEYS)
                steps[ki] = d
                c2, x2 = _moved(c, x, steps)
                take(c2, x2, ki)
    rng = np.random.default_rng(seed)
    for _ in range(corners):
        c2, x2 = _moved(c, x, [rng.integers(-1, 2, m) for _ in KEYS])
        take(c2, x2, -1)
    return sens, which


def chain_errors(got, ref):
    """relative error per output, 0 where the bits agree (an infinity the reference produces too) or both are NaN"""
    err = []
    for g, r in zip(got, ref):
        e = cases.rel_err(g, r)
        same = (g.view(np.uint32) == r.view(np.uint32)) | (np.isnan(g) & np.isnan(r))
        e = np.where(same.all(axis=0) if same.ndim == 2 else same, 0.0, e)
        err.append(np.where(np.isfinite(e), e, np.inf))
    return err
