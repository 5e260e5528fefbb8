"""helpers for the -m gpu parity tests: numpy <-> device planes, closure construction from case dicts"""
import numpy as np
import torch

import rlshaders_amd as R
import gpu_util_cpu


def ggx_oracle(O, case, exiting=None, nthreads=4):
    """the oracle closure the device is compared with: its light loops in the SECOND form of the estimator -- one sum per
    strategy, the order the kernels' separate passes produce (oracle/rls_oracle.c, ggx_light_loop).  The canonical form
    (one running sum) is what the CPU-only tests check; tests/test_oracle_light_loops.py ties the two together."""
    g = gpu_util_cpu.ggx_oracle(O, case, exiting=exiting, nthreads=nthreads)
    g.two_sums_default = True
    return g


def disney_oracle(O, case, nthreads=4):
    d = gpu_util_cpu.disney_oracle(O, case, nthreads=nthreads)
    d.two_sums_default = True
    return d


def dev(a):
    """numpy array / python scalar / tuple -> device tensor (arrays) or pass-through (uniform params)"""
    if isinstance(a, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return a


def host(t):
    return t.detach().cpu().numpy()


def ggx_sampler(ctx, case, exiting=None):
    return R.GgxSampler(ctx, dev(case["wo"]), dev(case["N"]), dev(case["T"]), specColor=dev(case["KsColor"]),
                        ior=dev(case["ior"]), roughness=dev(case["roughness"]), anisotropic=dev(case["anisotropic"]),
                        exiting=None if exiting is None else torch.from_numpy(exiting).cuda())


def disney_sampler(ctx, case):
    sc = {k: dev(case[k]) for k in R._capi.DISNEY_SCALARS if k in case}
    return R.DisneySampler(ctx, dev(case["wo"]), dev(case["N"]), dev(case["T"]),
                           base_color=dev(case.get("base_color", (1.0, 1.0, 1.0))), **sc)
