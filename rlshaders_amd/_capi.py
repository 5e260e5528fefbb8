"""ctypes binding of the C ABI declared in ``include/rlshaders_amd.h``.

The shared library is the product: there is no Python / CPU fallback.  If it is missing or fails
to load this module raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "librlshaders_amd.so"

RLS_OK = 0
RLS_RAY_DIFFUSE = 0x08
RLS_RAY_GLOSSY = 0x10
RLS_KERNEL_VNDF = 0
RLS_KERNEL_NDF = 1

c_float_p = C.POINTER(C.c_float)
c_u8_p = C.POINTER(C.c_uint8)


class CVec3(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p)]


class Vec3(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p)]


class CRgb(C.Structure):
    _fields_ = [("r", C.c_void_p), ("g", C.c_void_p), ("b", C.c_void_p)]


class Rgb(C.Structure):
    _fields_ = [("r", C.c_void_p), ("g", C.c_void_p), ("b", C.c_void_p)]


class Param(C.Structure):
    _fields_ = [("v", C.c_void_p), ("u", C.c_float)]


class ParamRgb(C.Structure):
    _fields_ = [("r", C.c_void_p), ("g", C.c_void_p), ("b", C.c_void_p),
                ("ur", C.c_float), ("ug", C.c_float), ("ub", C.c_float)]


class MaterialIndex(C.Structure):
    """rls_material_index: per-point material id (uint32 [n], device) + the length of the parameter columns"""
    _fields_ = [("id", C.c_void_p), ("count", C.c_uint32)]


class GgxClosure(C.Structure):
    _fields_ = [("wo", CVec3), ("N", CVec3), ("T", CVec3),
                ("exiting", C.c_void_p),
                ("KsColor", ParamRgb),
                ("specularRoughness", Param), ("ior", Param), ("anisotropic", Param),
                ("materials", MaterialIndex)]


class DisneyClosure(C.Structure):
    _fields_ = [("wo", CVec3), ("N", CVec3), ("T", CVec3),
                ("base_color", ParamRgb),
                ("subsurface", Param), ("metallic", Param), ("specular", Param),
                ("specular_tint", Param), ("roughness", Param), ("anisotropic", Param),
                ("sheen", Param), ("sheen_tint", Param), ("clearcoat", Param),
                ("clearcoat_gloss", Param),
                ("materials", MaterialIndex)]


DISNEY_SCALARS = ("subsurface", "metallic", "specular", "specular_tint", "roughness", "anisotropic",
                  "sheen", "sheen_tint", "clearcoat", "clearcoat_gloss")


class DisneyStreamOut(C.Structure):
    _fields_ = [("wi", Vec3), ("f", Rgb), ("pdf", C.c_void_p)]


# int consume(void *user, int64_t first_point, int64_t count, const rls_disney_stream_out *chunk)
DisneyChunkFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(DisneyStreamOut))


class SssClosure(C.Structure):
    _fields_ = [("sss_color", ParamRgb),
                ("sss_dist_multiplier", Param),
                ("sss_scatter_dist", Param * 3),
                ("N", CVec3), ("T", CVec3),
                ("has_dPdu", C.c_int),
                ("materials", MaterialIndex)]


class SphereLight(C.Structure):
    """rls_sphere_light"""
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("radiance", C.c_float * 3), ("mis_mode", C.c_int)]


class GgxShader(C.Structure):
    """rls_ggx_shader: KdColor, Kd, diffuseRoughness, Ks, KtColor, Kt (src/rlGgx.cpp:170-179)"""
    _fields_ = [("KdColor", ParamRgb), ("Kd", Param), ("diffuseRoughness", Param), ("Ks", Param),
                ("KtColor", ParamRgb), ("Kt", Param)]


RLS_MIS_BOTH, RLS_MIS_LIGHT_ONLY, RLS_MIS_BSDF_ONLY = 0, 1, 2


class SssScene(C.Structure):
    """rls_sss_scene."""
    _fields_ = [("geometry", C.c_int),
                ("plane_point", C.c_float * 3), ("plane_normal", C.c_float * 3),
                ("sphere_center", C.c_float * 3), ("sphere_radius", C.c_float),
                ("light_dir", C.c_float * 3), ("light_color", C.c_float * 3),
                ("has_gate", C.c_int),
                ("gate_point", C.c_float * 3), ("gate_normal", C.c_float * 3),
                ("use_cavity_fade", C.c_int), ("literal_matrix", C.c_int)]


RLS_SCENE_PLANE, RLS_SCENE_SPHERE = 0, 1


class SkinClosure(C.Structure):
    _fields_ = [("wo", CVec3), ("N", CVec3), ("T", CVec3),
                ("sss_color", ParamRgb),
                ("sss_weight", Param), ("sss_dist_multiplier", Param),
                ("sss_scatter_dist", Param * 3),
                ("specular_color", ParamRgb),
                ("specular_weight", Param), ("specular_roughness", Param), ("specular_ior", Param),
                ("sheen_color", ParamRgb),
                ("sheen_weight", Param), ("sheen_roughness", Param), ("sheen_ior", Param),
                ("materials", MaterialIndex)]


class SkinOut(C.Structure):
    _fields_ = [("sheen_wi", Vec3), ("sheen_f", Rgb), ("sheen_pdf", C.c_void_p), ("sheen_fresnel", C.c_void_p),
                ("spec_wi", Vec3), ("spec_f", Rgb), ("spec_pdf", C.c_void_p), ("spec_fresnel", C.c_void_p),
                ("r", C.c_void_p), ("r_pdf", C.c_void_p), ("profile", Rgb),
                ("sheenFresnel", C.c_void_p), ("specularFresnel", C.c_void_p), ("sssWeight", C.c_void_p)]


class GgxShadeOut(C.Structure):
    """rls_ggx_shade_out."""
    _fields_ = [("direct_diffuse", Rgb), ("direct_specular", Rgb), ("refraction", Rgb), ("indirect_diffuse", Rgb),
                ("indirect_specular", Rgb), ("out", Rgb)]


class DisneyShadeOut(C.Structure):
    """rls_disney_shade_out."""
    _fields_ = [("direct_diffuse", Rgb), ("direct_specular", Rgb), ("indirect_diffuse", Rgb), ("indirect_specular", Rgb),
                ("out", Rgb)]


class SkinIntegrateOut(C.Structure):
    """rls_skin_integrate_out."""
    _fields_ = [("sheen", Rgb), ("specular", Rgb), ("sss", Rgb), ("out", Rgb),
                ("sheenFresnel", C.c_void_p), ("specularFresnel", C.c_void_p), ("sssWeight", C.c_void_p)]


_ctx = C.c_void_p
_i64 = C.c_int64
_vp = C.c_void_p

# name -> (restype, argtypes).  Every symbol include/rlshaders_amd.h declares is listed here;
# tests/test_capi_symbols.py checks the two stay in step.
PROTOTYPES = {
    "rls_device_count": (C.c_int, []),
    "rls_shard_range": (C.c_int, [_i64, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "rls_context_create": (C.c_int, [C.c_int, C.POINTER(_ctx)]),
    "rls_context_destroy": (None, [_ctx]),
    "rls_context_set_stream": (C.c_int, [_ctx, _vp]),
    "rls_context_use_own_stream": (C.c_int, [_ctx]),
    "rls_context_set_math_mode": (C.c_int, [_ctx, C.c_int]),
    "rls_context_get_math_mode": (C.c_int, [_ctx]),
    "rls_context_get_stream": (_vp, [_ctx]),
    "rls_context_synchronize": (C.c_int, [_ctx]),
    "rls_context_device": (C.c_int, [_ctx]),
    "rls_last_error": (C.c_char_p, []),
    "rls_status_string": (C.c_char_p, [C.c_int]),
    "rls_version": (C.c_int, []),
    "rls_libm_flavour": (C.c_char_p, []),
    "rls_host_libm_matches": (C.c_int, [C.POINTER(C.c_int)]),
    "rls_device_info": (C.c_int, [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                  C.c_char_p, C.c_size_t]),
    "rls_device_alloc": (C.c_int, [_ctx, C.c_size_t, C.POINTER(_vp)]),
    "rls_device_free": (C.c_int, [_ctx, _vp]),
    "rls_copy_to_device": (C.c_int, [_ctx, _vp, _vp, C.c_size_t]),
    "rls_copy_to_host": (C.c_int, [_ctx, _vp, _vp, C.c_size_t]),
    "rls_timer_start": (C.c_int, [_ctx]),
    "rls_timer_stop": (C.c_int, [_ctx]),
    "rls_timer_elapsed_ms": (C.c_int, [_ctx, C.POINTER(C.c_float)]),
    "rls_diag_clock_stamps_begin": (C.c_int, [_ctx]),
    "rls_diag_clock_stamps_read": (C.c_int, [_ctx, _i64, _vp, C.POINTER(_i64)]),
    "rls_diag_clock_stamps_end": (C.c_int, [_ctx]),
    "rls_arena_create": (C.c_int, [_ctx, _i64, C.c_int, C.c_int, C.POINTER(_vp)]),
    "rls_arena_plane": (_vp, [_vp, C.c_int]),
    "rls_arena_info": (C.c_int, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_float),
                                 C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "rls_arena_destroy": (None, [_vp]),
    "rls_probe_block": (C.c_int, [_ctx, _vp, C.c_size_t, C.POINTER(C.c_float)]),
    "rls_libm_eval": (C.c_int, [_ctx, C.c_int, _i64, _vp, _vp, _vp]),
    "rls_graph_begin_capture": (C.c_int, [_ctx]),
    "rls_graph_end_capture": (C.c_int, [_ctx, C.POINTER(_vp)]),
    "rls_graph_launch": (C.c_int, [_ctx, _vp]),
    "rls_graph_destroy": (None, [_vp]),
    "rls_host_alloc": (C.c_int, [_ctx, C.c_size_t, C.POINTER(_vp)]),
    "rls_host_free": (C.c_int, [_ctx, _vp]),
    "rls_host_register": (C.c_int, [_ctx, _vp, C.c_size_t]),
    "rls_host_unregister": (C.c_int, [_ctx, _vp]),
    "rls_pipeline_create": (C.c_int, [_ctx, _i64, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "rls_pipeline_run": (C.c_int, [_vp, _i64, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp]),
    "rls_pipeline_destroy": (None, [_vp]),
    "rls_measure_copy_rates": (C.c_int, [_ctx, C.c_size_t, C.POINTER(C.c_float)]),
    # rlGgx
    "rls_ggx_sample": (C.c_int, [_ctx, _i64, C.POINTER(GgxClosure), _vp, _vp, Vec3, _vp]),
    "rls_ggx_eval": (C.c_int, [_ctx, _i64, C.POINTER(GgxClosure), CVec3, Rgb]),
    "rls_ggx_pdf": (C.c_int, [_ctx, _i64, C.POINTER(GgxClosure), CVec3, _vp]),
    "rls_ggx_sample_eval_pdf": (C.c_int, [_ctx, _i64, C.POINTER(GgxClosure), _vp, _vp, Vec3, Rgb, _vp, _vp]),
    "rls_ggx_refract_sample": (C.c_int, [_ctx, _i64, C.POINTER(GgxClosure), _vp, _vp, Vec3, _vp, _vp]),
    "rls_ggx_reflect_refract": (C.c_int, [_ctx, _i64, C.POINTER(GgxClosure), _vp, _vp, _vp, _vp,
                                          Vec3, Rgb, _vp, _vp, Vec3, _vp]),
    "rls_ggx_microfacet": (C.c_int, [_ctx, _i64, C.POINTER(GgxClosure), C.c_int, _vp, _vp, Vec3]),
    "rls_ggx_ndf_pdf": (C.c_int, [_ctx, _i64, C.POINTER(GgxClosure), CVec3, _vp]),
    "rls_ggx_direct_lighting": (C.c_int, [_ctx, _i64, C.POINTER(GgxClosure), C.POINTER(GgxShader), CVec3,
                                          C.POINTER(SphereLight), C.c_int, C.c_int, C.c_uint32, C.c_uint64, Rgb, Rgb]),
    "rls_ggx_shade": (C.c_int, [_ctx, _i64, C.POINTER(GgxClosure), C.POINTER(GgxShader), CVec3, C.POINTER(SphereLight),
                                C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_uint32, C.c_uint64,
                                C.POINTER(GgxShadeOut)]),
    "rls_disney_shade": (C.c_int, [_ctx, _i64, C.POINTER(DisneyClosure), CVec3, C.POINTER(SphereLight), C.c_int,
                                   C.POINTER(C.c_float), C.c_int, C.c_uint32, C.c_uint64, C.POINTER(DisneyShadeOut)]),
    "rls_disney_direct_lighting": (C.c_int, [_ctx, _i64, C.POINTER(DisneyClosure), CVec3, C.POINTER(SphereLight), C.c_int,
                                             C.c_int, C.c_uint32, C.c_uint64, Rgb, Rgb]),
    "rls_ggx_integrate_refract": (C.c_int, [_ctx, _i64, C.POINTER(GgxClosure), C.c_int, C.POINTER(C.c_float), C.c_int,
                                            C.c_uint32, C.c_uint64, Rgb, _vp]),
    "rls_skin_integrate": (C.c_int, [_ctx, _i64, C.POINTER(SkinClosure), CVec3, C.POINTER(SssScene), C.POINTER(C.c_float),
                                     C.POINTER(SphereLight), C.c_int, C.c_int, C.c_uint32, C.c_uint64,
                                     C.POINTER(SkinIntegrateOut)]),
    "rls_ggx_integrate": (C.c_int, [_ctx, _i64, C.POINTER(GgxClosure), C.c_int, C.c_uint32, C.c_uint64, Rgb, _vp]),
    # rlDisney
    "rls_disney_sample": (C.c_int, [_ctx, _i64, C.POINTER(DisneyClosure), C.c_int, _vp, _vp, Vec3]),
    "rls_disney_eval": (C.c_int, [_ctx, _i64, C.POINTER(DisneyClosure), C.c_int, CVec3, Rgb]),
    "rls_disney_pdf": (C.c_int, [_ctx, _i64, C.POINTER(DisneyClosure), C.c_int, CVec3, _vp]),
    "rls_disney_sample_eval_pdf": (C.c_int, [_ctx, _i64, C.POINTER(DisneyClosure), C.c_int, _vp, _vp,
                                             Vec3, Rgb, _vp]),
    "rls_disney_integrate": (C.c_int, [_ctx, _i64, C.POINTER(DisneyClosure), C.c_int, C.c_uint32, C.c_uint64,
                                       Rgb, _vp, Rgb, _vp, C.POINTER(DisneyStreamOut)]),
    "rls_disney_integrate_chunked": (C.c_int, [_ctx, _i64, C.POINTER(DisneyClosure), C.c_int, C.c_uint32, C.c_uint64,
                                               Rgb, _vp, Rgb, _vp, _i64, C.POINTER(DisneyStreamOut),
                                               DisneyChunkFn, C.c_void_p]),
    "rls_disney_alt_sample": (C.c_int, [_ctx, _i64, C.POINTER(DisneyClosure), C.c_int, _vp, _vp, Vec3]),
    "rls_disney_alt_pdf": (C.c_int, [_ctx, _i64, C.POINTER(DisneyClosure), CVec3, _vp]),
    "rls_disney_d_gtr2": (C.c_int, [_ctx, _i64, C.POINTER(DisneyClosure), CVec3, _vp]),
    "rls_gaussian_sample": (C.c_int, [_ctx, _i64, Param, _vp, _vp, _vp, _vp]),
    # rlSss
    "rls_nd_sample": (C.c_int, [_ctx, _i64, C.POINTER(SssClosure), _vp, _vp, _vp, Rgb]),
    "rls_nd_pdf": (C.c_int, [_ctx, _i64, C.POINTER(SssClosure), _vp, _vp]),
    "rls_nd_eval": (C.c_int, [_ctx, _i64, C.POINTER(SssClosure), _vp, Rgb]),
    "rls_sss_probe_ray": (C.c_int, [_ctx, _i64, C.POINTER(SssClosure), _vp, _vp, CVec3,
                                    _vp, Vec3, Vec3, _vp, _vp, Rgb]),
    "rls_sss_mis_pdf": (C.c_int, [_ctx, _i64, C.POINTER(SssClosure), CVec3, CVec3, C.c_int, _vp]),
    "rls_sss_cavity_fade": (C.c_int, [_ctx, _i64, CVec3, CVec3, CVec3, _vp]),
    "rls_sss_sample_diffuse_direction": (C.c_int, [_ctx, _i64, CVec3, CVec3, _vp, _vp, Vec3]),
    "rls_sss_integrate_scatter": (C.c_int, [_ctx, _i64, C.POINTER(SssClosure), CVec3, C.POINTER(SssScene),
                                            C.c_int, C.c_uint32, C.c_uint64, Rgb, _vp]),
    # rlSkin
    "rls_skin_sample_eval_pdf": (C.c_int, [_ctx, _i64, C.POINTER(SkinClosure), C.POINTER(_vp), C.POINTER(SkinOut)]),
    # rlUtil, generator, checksum
    "rls_util_directions": (C.c_int, [_ctx, _i64, _vp, _vp, Vec3, Vec3]),
    "rls_util_reflect_luminance": (C.c_int, [_ctx, _i64, CVec3, CVec3, CVec3, Vec3, _vp]),
    "rls_gen_frame": (C.c_int, [_ctx, C.c_uint32, C.c_uint64, _i64, Vec3, Vec3, Vec3]),
    "rls_gen_uniform": (C.c_int, [_ctx, C.c_uint32, C.c_uint64, _i64, C.c_uint32, C.c_float, C.c_float, _vp]),
    "rls_gen_aniso": (C.c_int, [_ctx, C.c_uint32, C.c_uint64, _i64, _vp]),
    "rls_checksum": (C.c_int, [_ctx, _i64, _vp, C.POINTER(C.c_uint64)]),
}

_lib = None


class RlsError(RuntimeError):
    """A C-ABI call returned a non-zero rls_status."""

    def __init__(self, status: int, message: str):
        super().__init__(f"rlshaders_amd: status {status}: {message}")
        self.status = status


# rls_pipeline_launch_fn: (user, slot context, first_point, count, device_in planes, device_out planes) -> rls_status
PipelineLaunchFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p),
                               C.POINTER(C.c_void_p))


def load() -> C.CDLL:
    """Load (once) and prototype the HIP library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("RLSHADERS_AMD_LIB", str(LIB_PATH)))
    if not path.exists():
        raise RuntimeError(
            f"rlshaders_amd: HIP library not found at {path}. Build it with "
            f"`python -m rlshaders_amd.build` (needs hipcc); there is no CPU fallback.")
    lib = C.CDLL(str(path), mode=C.RTLD_GLOBAL)
    missing = []
    for name, (restype, argtypes) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if not name.startswith("rls_diag_"):       # a library built with RLS_DIAGNOSTICS=0 exports the drop-in surface only
                missing.append(name)
            continue
        fn.restype = restype
        fn.argtypes = argtypes
    if missing:
        raise RuntimeError(f"rlshaders_amd: {path} lacks C-ABI symbols: {missing}")
    _lib = lib
    global _lib_path
    _lib_path = path.resolve()
    return lib


_lib_path = None


def loaded_path() -> Path:
    """the file the C ABI was loaded from (RLSHADERS_AMD_LIB or the in-tree build) -- what rlshaders_amd.codeid identifies"""
    load()
    return _lib_path


def check(status: int) -> None:
    if status != RLS_OK:
        lib = load()
        msg = lib.rls_last_error().decode("utf-8", "replace")
        if not msg:
            msg = lib.rls_status_string(status).decode()
        raise RlsError(status, msg)
