"""rlshaders_amd -- MI355X-native batched BSDF evaluator / importance sampler.

A drop-in for the closure layer (sample / eval / pdf) of shihchinw/rlShaders' rlGgx, rlDisney and
rlSss/rlSkin shaders: hand-written HIP kernels for gfx950 behind the C ABI in
``include/rlshaders_amd.h``.  This package is the host-side mirror of the reference's closure
classes; it fails loudly when the HIP library is missing -- there is no CPU path.
"""
from ._capi import (RLS_KERNEL_NDF, RLS_KERNEL_VNDF, RLS_RAY_DIFFUSE, RLS_RAY_GLOSSY, RlsError, load)
from .closures import (Arena, Context, DisneySampler, GaussianProfile, GgxSampler, NDProfile, Pipeline, SkinShader, SssSampler,
                       checksum, host_libm_mismatches, libm_flavour,
                       gen_aniso, gen_frame, gen_uniform, make_light, make_scene, util_directions, util_reflect_luminance)

__all__ = [
    "Context", "Arena", "Pipeline", "GgxSampler", "DisneySampler", "NDProfile", "GaussianProfile", "SssSampler", "SkinShader",
    "RLS_RAY_DIFFUSE", "RLS_RAY_GLOSSY", "RLS_KERNEL_VNDF", "RLS_KERNEL_NDF",
    "RlsError", "load", "gen_frame", "gen_uniform", "gen_aniso", "checksum", "libm_flavour", "host_libm_mismatches", "util_directions", "util_reflect_luminance", "make_scene", "make_light",
]
