"""GPU: context, memory, stream, timer and error behaviour of the C ABI (no torch in the data path:
buffers come from rls_device_alloc, exactly as a C host would use the library)."""
import ctypes as C

import numpy as np
import pytest

import cases
import rlshaders_amd as R
from rlshaders_amd import _capi as capi

pytestmark = pytest.mark.gpu


def _ctx():
    lib = R.load()
    h = C.c_void_p()
    assert lib.rls_context_create(0, C.byref(h)) == 0
    return lib, h


def test_pure_c_abi_round_trip(oracle):
    """alloc -> upload -> rls_ggx_sample_eval_pdf (uniform parameters) -> download, on the context's own stream"""
    lib, h = _ctx()
    n = 10007
    wo, N, T = cases.frame(cases.SEED_PARITY, n)
    x = cases.xi(cases.SEED_PARITY, n, 2)
    host_in = np.concatenate([wo, N, T, x]).astype(np.float32)          # 11 planes
    planes_in, planes_out = 11, 8
    buf = C.c_void_p()
    assert lib.rls_device_alloc(h, (planes_in + planes_out) * n * 4, C.byref(buf)) == 0
    assert lib.rls_copy_to_device(h, buf, host_in.ctypes.data_as(C.c_void_p), host_in.nbytes) == 0
    p = lambda k: buf.value + 4 * n * k
    c = capi.GgxClosure()
    c.wo, c.N, c.T = capi.CVec3(p(0), p(1), p(2)), capi.CVec3(p(3), p(4), p(5)), capi.CVec3(p(6), p(7), p(8))
    c.KsColor = capi.ParamRgb(None, None, None, 1.0, 0.9, 0.8)
    c.specularRoughness = capi.Param(None, 0.35)
    c.ior = capi.Param(None, 1.35)
    c.anisotropic = capi.Param(None, 0.0)
    o = planes_in
    st = lib.rls_ggx_sample_eval_pdf(h, n, C.byref(c), p(9), p(10), capi.Vec3(p(o), p(o + 1), p(o + 2)),
                                     capi.Rgb(p(o + 3), p(o + 4), p(o + 5)), p(o + 6), p(o + 7))
    assert st == 0, lib.rls_last_error()
    out = np.empty(planes_out * n, np.float32)
    assert lib.rls_copy_to_host(h, out.ctypes.data_as(C.c_void_p), p(o), out.nbytes) == 0
    out = out.reshape(planes_out, n)
    ref = oracle.Ggx(wo, N, T, KsColor=(1.0, 0.9, 0.8), ior=1.35, roughness=0.35).sample_eval_pdf(x[0], x[1])
    assert np.array_equal(out[0:3].view(np.uint32), ref[0].view(np.uint32))
    assert np.array_equal(out[3:6].view(np.uint32), ref[1].view(np.uint32))
    assert np.array_equal(out[6].view(np.uint32), ref[2].view(np.uint32))
    assert np.array_equal(out[7].view(np.uint32), ref[3].view(np.uint32))
    assert lib.rls_device_free(h, buf) == 0
    lib.rls_context_destroy(h)


def test_device_info_and_timer():
    lib, h = _ctx()
    cus, total, free = C.c_int(), C.c_size_t(), C.c_size_t()
    name = C.create_string_buffer(64)
    assert lib.rls_device_info(h, C.byref(cus), C.byref(total), C.byref(free), name, 64) == 0
    assert cus.value == 256 and name.value.decode().startswith("gfx950")
    assert total.value > 200 * (1 << 30) and 0 < free.value <= total.value      # 288 GB HBM3E
    assert lib.rls_context_device(h) == 0 and lib.rls_context_get_math_mode(h) == 0
    buf = C.c_void_p()
    n = 1 << 24
    assert lib.rls_device_alloc(h, 4 * n, C.byref(buf)) == 0
    assert lib.rls_timer_start(h) == 0
    assert lib.rls_gen_uniform(h, 1, 0, n, 3, 0.0, 1.0, buf) == 0
    assert lib.rls_timer_stop(h) == 0
    ms = C.c_float()
    assert lib.rls_timer_elapsed_ms(h, C.byref(ms)) == 0 and ms.value > 0      # positive; how long is the box's business
    ck = C.c_uint64()
    assert lib.rls_checksum(h, n, buf, C.byref(ck)) == 0 and ck.value != 0
    assert lib.rls_context_synchronize(h) == 0
    lib.rls_device_free(h, buf)
    lib.rls_context_destroy(h)


def test_error_paths():
    lib, h = _ctx()
    assert lib.rls_context_create(99, C.byref(C.c_void_p())) == 2 and b"out of range" in lib.rls_last_error()
    assert lib.rls_context_create(0, None) == 1
    big = C.c_void_p()
    st = lib.rls_device_alloc(h, 1 << 50, C.byref(big))
    assert st in (3, 4) and not big.value, st                      # HIP error or out of memory, never a crash
    assert lib.rls_device_alloc(h, 0, C.byref(big)) == 0 and not big.value
    assert lib.rls_device_free(h, None) == 0
    assert lib.rls_copy_to_host(h, None, None, 16) == 1
    c = capi.DisneyClosure()
    v = capi.Vec3(None, None, None)
    assert lib.rls_disney_sample(h, 8, C.byref(c), 3, None, None, v) == 1 and b"lobe" in lib.rls_last_error()
    assert lib.rls_ggx_integrate(h, 8, None, 0, 1, 0, capi.Rgb(None, None, None), None) == 1
    # the light loops and whole-node entry points: n = 0 is a no-op before any pointer is looked at; then NULL
    # arguments, light counts and sample counts are rejected with a message, never dereferenced
    z3, zrgb = capi.CVec3(None, None, None), capi.Rgb(None, None, None)
    env = (C.c_float * 3)(1.0, 1.0, 1.0)
    assert lib.rls_ggx_shade(h, 0, None, None, z3, None, 0, None, 1, 4, 1, 0, None) == 0
    assert lib.rls_disney_shade(h, 0, None, z3, None, 0, None, 4, 1, 0, None) == 0
    assert lib.rls_disney_direct_lighting(h, 0, None, z3, None, 0, 4, 1, 0, zrgb, zrgb) == 0
    assert lib.rls_ggx_shade(h, 8, None, None, z3, None, 0, env, 1, 4, 1, 0, None) == 1 and b"NULL" in lib.rls_last_error()
    assert lib.rls_disney_shade(h, 8, None, z3, None, 0, env, 4, 1, 0, None) == 1 and b"NULL" in lib.rls_last_error()
    d = capi.DisneyClosure()
    assert lib.rls_disney_direct_lighting(h, 8, C.byref(d), z3, None, 1, 4, 1, 0, zrgb, zrgb) == 1      # planes are NULL
    assert lib.rls_disney_direct_lighting(h, 8, C.byref(d), z3, None, 1, 17, 1, 0, zrgb, zrgb) == 1 and b"spp_n" in lib.rls_last_error()
    assert lib.rls_status_string(4) == b"out of device memory"
    assert lib.rls_version() == 1
    lib.rls_context_destroy(h)
    lib.rls_context_destroy(None)                                   # no-op


def test_streams_are_taken_literally():
    """launches on torch's current stream see data produced on that stream (the handle 0 = null
    stream is taken literally, not as "use the context's own stream")"""
    import torch
    ctx = R.Context(0)
    assert ctx.lib.rls_context_get_stream(ctx.handle) in (None, 0)
    side = torch.cuda.Stream()
    n = 1 << 22
    with torch.cuda.stream(side):
        ctx.use_stream(side)
        a = R.gen_uniform(ctx, 5, 0, n, 1)
        b = (a * 2).contiguous()                    # torch op on the same stream, no explicit sync
        ck = R.checksum(ctx, b)
    ctx.use_stream(torch.cuda.current_stream())
    assert ck == R.checksum(ctx, (R.gen_uniform(ctx, 5, 0, n, 1) * 2).contiguous())
    ctx.use_stream(None)                            # the context's private stream
    assert ctx.lib.rls_context_get_stream(ctx.handle) not in (None, 0)
    ctx.close()
