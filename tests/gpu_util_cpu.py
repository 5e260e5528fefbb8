"""oracle-side closure construction from case dicts (no torch: usable by the CPU-only tests)"""


def ggx_oracle(O, case, exiting=None, nthreads=4):
    return O.Ggx(case["wo"], case["N"], case["T"], KsColor=case["KsColor"], ior=case["ior"],
                 roughness=case["roughness"], anisotropic=case["anisotropic"], exiting=exiting, nthreads=nthreads)


def disney_oracle(O, case, nthreads=4):
    sc = {k: case[k] for k in O.DISNEY_SCALARS if k in case}
    return O.Disney(case["wo"], case["N"], case["T"], base_color=case.get("base_color", (1.0, 1.0, 1.0)),
                    nthreads=nthreads, **sc)
