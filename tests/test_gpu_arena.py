"""GPU: plane arenas (rls_arena_*, rls_probe_block, R.Arena): planes are disjoint, aligned and usable by the
kernels; probing keeps the fastest candidate; results do not depend on where the planes live."""
import ctypes as C

import numpy as np
import pytest
import torch

import cases
import rlshaders_amd as R
from gpu_util import dev, ggx_sampler, host

pytestmark = pytest.mark.gpu


def test_c_arena_planes_and_info(gpu):
    lib, n, planes = gpu.lib, 100003, 7            # ragged n: planes are padded to 256-byte boundaries
    h = C.c_void_p()
    R._capi.check(lib.rls_arena_create(gpu.handle, n, planes, 3, C.byref(h)))
    ptrs = [lib.rls_arena_plane(h, k) for k in range(planes)]
    assert all(p and p % 256 == 0 for p in ptrs)
    assert all(b - a >= n * 4 for a, b in zip(ptrs, ptrs[1:]))
    assert lib.rls_arena_plane(h, planes) is None and lib.rls_arena_plane(h, -1) is None
    size, cand = C.c_size_t(), C.c_int()
    g, lo, hi = C.c_float(), C.c_float(), C.c_float()
    R._capi.check(lib.rls_arena_info(h, C.byref(size), C.byref(cand), C.byref(g), C.byref(lo), C.byref(hi)))
    assert size.value >= planes * n * 4 and cand.value == 3
    assert lo.value <= g.value == hi.value and g.value > 0          # the kept block is the fastest one probed
    lib.rls_arena_destroy(h)
    assert lib.rls_arena_create(gpu.handle, 0, 3, 1, C.byref(h)) == 1
    assert lib.rls_arena_create(gpu.handle, 10, 3, 17, C.byref(h)) == 1


def test_results_do_not_depend_on_placement(gpu):
    n = 50021
    c = cases.ggx_mixed(cases.SEED_PARITY, n)
    x = cases.xi(cases.SEED_PARITY, n, 4)
    ref = [host(t) for t in ggx_sampler(gpu, c).reflectRefract(*(dev(x[k]) for k in range(4)))]
    A = R.Arena(gpu, n, 31, candidates=3)
    info = A.info()
    assert info["arena"] and info["candidates_probed"] == 3 and info["chosen_gb_per_s"] == max(A.probe_gbs)

    def put(a):
        t = A.planes(a.shape[0]) if a.ndim == 2 else A.plane()
        t.copy_(torch.from_numpy(a))
        return t
    g = R.GgxSampler(gpu, put(c["wo"]), put(c["N"]), put(c["T"]), specColor=put(c["KsColor"]), ior=put(c["ior"]),
                     roughness=put(c["roughness"]), anisotropic=put(c["anisotropic"]))
    xi = [put(x[k]) for k in range(4)]
    out = (A.planes(3), A.planes(3), A.plane(), A.plane(), A.planes(3), A.plane())
    got = [host(t) for t in g.reflectRefract(*xi, out=out)]
    for a, b in zip(got, ref):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    with pytest.raises(RuntimeError):
        A.plane()                                                     # all 31 planes handed out
