"""helpers for the -m gpu parity tests: numpy <-> device planes, closure construction from case dicts"""
import numpy as np
import torch

import rlshaders_amd as R


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


def ggx_oracle(O, case, exiting=None, nthreads=4):
    return O.Ggx(case["wo"], case["N"], case["T"], KsColor=case["KsColor"], ior=case["ior"],
                 roughness=case["roughness"], anisotropic=case["anisotropic"], exiting=exiting, nthreads=nthreads)


def disney_sampler(ctx, case):
    sc = {k: dev(case[k]) for k in R._capi.DISNEY_SCALARS if k in case}
    return R.DisneySampler(ctx, dev(case["wo"]), dev(case["N"]), dev(case["T"]),
                           base_color=dev(case.get("base_color", (1.0, 1.0, 1.0))), **sc)


def disney_oracle(O, case, nthreads=4):
    sc = {k: case[k] for k in O.DISNEY_SCALARS if k in case}
    return O.Disney(case["wo"], case["N"], case["T"], base_color=case.get("base_color", (1.0, 1.0, 1.0)),
                    nthreads=nthreads, **sc)
