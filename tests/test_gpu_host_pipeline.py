"""GPU: host-resident batches (rls_pipeline_*, rlshaders_amd.Pipeline, rlsb::Pipeline).  The shading points of an
Arnold-side stub start in host memory (the reference evaluates per hit on CPU render threads, src/rlGgx.cpp:248-261);
the pipeline sends them through the GPU in overlapped chunks.  Its results must be those of ONE device-resident call
on the same points, bit for bit -- ragged batches, batches shorter than a chunk, one slot or several, uniform
parameters (host planes that are not streamed), and the oracle on top."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

import cases
import rlshaders_amd as R
from gpu_util import dev, ggx_oracle, host

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _pinned(a: np.ndarray) -> torch.Tensor:
    t = torch.empty(a.shape[0], dtype=torch.float32, pin_memory=True)
    t.copy_(torch.from_numpy(np.ascontiguousarray(a)))
    return t


def _launch(slot, first, count, i, o):
    # everything on the slot's Context: its stream orders the upload, these kernels and the download.  The verbs read and
    # write the chunk's device planes in place ([3, count] views of three consecutive planes for vectors and colours)
    g = R.GgxSampler(slot, i.rows(0), i.rows(3), i.rows(6), specColor=i.rows(9), roughness=i[12], ior=i[13], anisotropic=i[14])
    g.reflectRefract(i[15], i[16], i[17], i[18], out=(o.rows(0), o.rows(3), o[6], o[7], o.rows(8), o[11]))


@pytest.mark.parametrize("n,chunk,depth,one_array", [(100_003, 1 << 14, 3, False), (5000, 1 << 14, 2, False),
                                                    (1 << 16, 1 << 14, 1, True), (70_001, 4096, 4, False),
                                                    (100_003, 1 << 14, 3, True), (70_001, 4096, 2, True)])
def test_pipeline_equals_device_resident(gpu, oracle, n, chunk, depth, one_array):
    """one_array: the host planes are rows of ONE page-locked [planes, n] array (a stub's batch buffers, rlsb::HostPlanes) and
    travel as one strided copy per chunk and direction; otherwise separate allocations, one copy per plane (also when they
    happen to be equally spaced: the runtime refuses a strided copy across allocations and rls_pipeline_run falls back)"""
    c = cases.ggx_mixed(cases.SEED_PARITY, n)
    x = cases.xi(cases.SEED_PARITY, n, 4)
    planes = [c["wo"][k] for k in range(3)] + [c["N"][k] for k in range(3)] + [c["T"][k] for k in range(3)] + \
             [c["KsColor"][k] for k in range(3)] + [c["roughness"], c["ior"], c["anisotropic"]] + [x[k] for k in range(4)]
    if one_array:
        hin_all = torch.empty(19, n, dtype=torch.float32, pin_memory=True)
        hin_all.copy_(torch.from_numpy(np.stack(planes)))
        hout_all = torch.full((12, n), float("nan"), dtype=torch.float32).pin_memory()
        hin, hout = [hin_all[k] for k in range(19)], [hout_all[k] for k in range(12)]
    else:
        hin = [_pinned(p) for p in planes]
        hout = [torch.full((n,), float("nan"), dtype=torch.float32).pin_memory() for _ in range(12)]
    pipe = R.Pipeline(gpu, chunk, 19, 12, depth)
    try:
        pipe.run(n, hin, hout, _launch)
    finally:
        pipe.close()
    # the same points in one device-resident call
    from gpu_util import ggx_sampler
    ref = [host(t) for t in ggx_sampler(gpu, c).reflectRefract(dev(x[0]), dev(x[1]), dev(x[2]), dev(x[3]))]
    flat = [ref[0][0], ref[0][1], ref[0][2], ref[1][0], ref[1][1], ref[1][2], ref[2], ref[3], ref[4][0], ref[4][1], ref[4][2], ref[5]]
    for k, (a, b) in enumerate(zip(hout, flat)):
        assert np.array_equal(a.numpy().view(np.uint32), b.view(np.uint32)), (k, n, chunk, depth, one_array)
    # and the oracle
    orc = ggx_oracle(oracle, c).reflect_refract(x[0], x[1], x[2], x[3])
    got_f = np.stack([hout[3].numpy(), hout[4].numpy(), hout[5].numpy()])
    cases.assert_tight(cases.summarize(cases.rel_err(got_f, orc[1])), "pipeline f")


def test_pipeline_uniform_planes_and_argument_checks(gpu):
    """host planes that are None are not copied: uniform node parameters, outputs nobody wants"""
    n = 20_000
    c = cases.ggx_mixed(cases.SEED_EDGE, n)
    x = cases.xi(cases.SEED_EDGE, n, 2)
    hin = [_pinned(c["wo"][k]) for k in range(3)] + [_pinned(c["N"][k]) for k in range(3)] + \
          [_pinned(c["T"][k]) for k in range(3)] + [None] + [_pinned(x[0]), _pinned(x[1])]
    hout = [torch.empty(n, dtype=torch.float32).pin_memory() for _ in range(3)] + [None]
    seen = []
    scratch = torch.empty(8192, device="cuda")        # the Fresnel side output nobody downloads

    def launch(slot, first, count, i, o):
        assert i[9] is None and o[3] is None and all(i[k].shape == (count,) for k in range(9)) and i.rows(8) is None
        seen.append((first, count))
        g = R.GgxSampler(slot, i.rows(0), i.rows(3), i.rows(6), specColor=(0.9, 0.8, 0.7), roughness=0.35, ior=1.5)
        g.evalSample(i[10], i[11], out_wi=o.rows(0), out_fresnel=scratch[:count])

    pipe = R.Pipeline(gpu, 8192, 12, 4, 2)
    try:
        pipe.run(n, hin, hout, launch)
        assert seen == [(0, 8192), (8192, 8192), (16384, n - 16384)]
        g = R.GgxSampler(gpu, dev(c["wo"]), dev(c["N"]), dev(c["T"]), specColor=(0.9, 0.8, 0.7), roughness=0.35, ior=1.5)
        wi = host(g.evalSample(dev(x[0]), dev(x[1]))[0])
        for k in range(3):
            assert np.array_equal(hout[k].numpy().view(np.uint32), wi[k].view(np.uint32))
        with pytest.raises(TypeError):                      # pageable host memory is refused
            pipe.run(n, [torch.empty(n)] + hin[1:], hout, launch)
        with pytest.raises(ValueError):
            pipe.run(n, hin[:5], hout, launch)

        def bad(slot, first, count, i, o):
            raise RuntimeError("consumer failed")
        with pytest.raises(RuntimeError, match="consumer failed"):
            pipe.run(n, hin, hout, bad)
        pipe.run(n, hin, hout, launch)                      # still usable after a failed run
    finally:
        pipe.close()
    lib = gpu.lib
    import ctypes as C
    h = C.c_void_p()
    assert lib.rls_pipeline_create(gpu.handle, 0, 1, 1, 2, C.byref(h)) == 1        # chunk_points < 1
    assert lib.rls_pipeline_create(gpu.handle, 16, 1, 1, 0, C.byref(h)) == 1       # depth out of range
    rates = R.Pipeline(gpu, 1024, 1, 1, 1)
    try:
        r = rates.copy_rates(1 << 26)
        assert r["h2d"] > 1.0 and r["d2h"] > 1.0 and r["both"] > 1.0
    finally:
        rates.close()


def test_cpp_host_pipeline_example():
    """rlsb::Pipeline + rlsb::HostPlanes (rls_host_alloc): the C++ stub's path, checked against the device-resident call
    inside the example (exit code) and reported with the box's copy rates"""
    from rlshaders_amd import build
    build.build_library()
    build.build_host_examples()
    exe = ROOT / "rlshaders_amd" / "build" / "example_host_pipeline"
    p = subprocess.run([str(exe), "21", "17", "3", "2"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout, p.stderr)
    got = json.loads(p.stdout.strip().splitlines()[-1])
    assert got["bit_identical_to_device_resident"] is True and got["points"] == (1 << 21) - 37
    assert got["gsamples_per_s"] > 0 and got["box_h2d"] > 1.0
    print(got)
    # the whole-node mode: rls_ggx_shade per chunk, parameters by reference, only sg->out.RGB downloaded
    p = subprocess.run([str(exe), "20", "16", "3", "2", "shade"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout, p.stderr)
    got = json.loads(p.stdout.strip().splitlines()[-1])
    assert got["mode"] == "shade" and got["bit_identical_to_device_resident"] is True and got["points"] == (1 << 20) - 37
    print(got)


def test_whole_node_by_reference_through_the_pipeline(gpu, oracle):
    """The whole shader_evaluate of rlGgx on a host-resident batch whose parameters go by reference (a material id per point, the
    node instances' sixteen parameters as device-resident columns): per shading point only the geometry and the id go up and
    sg->out.RGB comes down -- the AOV planes exist in the slot, the kernel writes them, nobody downloads them.  Chunked with
    first_index = the chunk's first point, so every point draws the batch's sample numbers: the result equals ONE
    device-resident call with the parameters expanded into planes, and the oracle's shader_evaluate."""
    n, chunk, m = 30_011, 4096, 13
    c = cases.ggx_mixed(cases.SEED_PARITY, n)
    t = cases.ggx_mixed(cases.SEED_EDGE, m)
    u = lambda j, k=m: oracle.gen_uniform(17, 0, k, oracle.S_PARAM0 + j)
    sh = dict(KdColor=np.stack([u(0), u(1), u(2)]), Kd=u(3), diffuseRoughness=u(4), Ks=u(5), KtColor=np.stack([u(6), u(7), u(8)]), Kt=u(9))
    ids = np.random.default_rng(3).integers(0, m, n).astype(np.int32)
    P = np.stack([oracle.gen_uniform(17, 0, n, 40 + j, 0.0, 4.0 if j < 2 else 1.0) for j in range(3)])
    planes = [c["wo"][k] for k in range(3)] + [c["N"][k] for k in range(3)] + [c["T"][k] for k in range(3)] + [P[k] for k in range(3)] + \
             [ids.view(np.float32)]
    hin_all = torch.empty(13, n, dtype=torch.float32, pin_memory=True)
    hin_all.copy_(torch.from_numpy(np.stack(planes)))
    hout_all = torch.full((3, n), float("nan"), dtype=torch.float32).pin_memory()
    hin, hout = [hin_all[k] for k in range(13)], [None] * 15 + [hout_all[k] for k in range(3)]
    lights = [R.make_light(center=(2.0, 2.0, 3.0), radius=1.25, radiance=(3.0, 2.0, 1.0)),
              R.make_light(center=(-3.0, 1.0, 2.5), radius=0.5, radiance=(0.5, 4.0, 2.0), mis_mode=2)]
    tab = {k: dev(v) for k, v in dict(specColor=t["KsColor"], ior=t["ior"], roughness=t["roughness"], anisotropic=t["anisotropic"]).items()}
    shd = {k: dev(v) for k, v in sh.items()}

    def launch(slot, first, count, i, o):
        g = R.GgxSampler(slot, i.rows(0), i.rows(3), i.rows(6), materials=(i.device_rows(12, dtype=torch.int32), m), **tab)
        # the AOV planes are not streamed to the host: device_rows() hands out the slot's planes all the same
        out = {k: o.device_rows(3 * j, 3) for j, k in enumerate(R.GgxSampler.SHADE_AOVS + ("out",))}
        g.shade(i.rows(9), lights, 2, 99, env=(1.0, 0.9, 0.8), out=out, first_index=first, **shd)

    pipe = R.Pipeline(gpu, chunk, 13, 18, 3)
    try:
        pipe.run(n, hin, hout, launch)
    finally:
        pipe.close()
    expanded = dict(KsColor=t["KsColor"][:, ids], ior=t["ior"][ids], roughness=t["roughness"][ids], anisotropic=t["anisotropic"][ids])
    shx = {k: np.ascontiguousarray(v[..., ids]) for k, v in sh.items()}
    g = R.GgxSampler(gpu, dev(c["wo"]), dev(c["N"]), dev(c["T"]), specColor=dev(np.ascontiguousarray(expanded["KsColor"])),
                     ior=dev(expanded["ior"]), roughness=dev(expanded["roughness"]), anisotropic=dev(expanded["anisotropic"]))
    ref = host(g.shade(dev(P), lights, 2, 99, env=(1.0, 0.9, 0.8), **{k: dev(v) for k, v in shx.items()})["out"])
    got = hout_all.numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    og = ggx_oracle(oracle, dict(wo=c["wo"], N=c["N"], T=c["T"], **{k: np.ascontiguousarray(v) for k, v in expanded.items()}))
    olts = [oracle.make_light(center=(2.0, 2.0, 3.0), radius=1.25, radiance=(3.0, 2.0, 1.0)),
            oracle.make_light(center=(-3.0, 1.0, 2.5), radius=0.5, radiance=(0.5, 4.0, 2.0), mis_mode=2)]
    orc = og.shade(P, olts, 2, 99, Kd_color=shx["KdColor"], Kd=shx["Kd"], Kd_roughness=shx["diffuseRoughness"], Ks=shx["Ks"],
                   Kt_color=shx["KtColor"], Kt=shx["Kt"], env=(1.0, 0.9, 0.8))["out"]
    cases.assert_tight(cases.summarize(cases.rel_err(got, orc)), "whole node through the pipeline vs oracle")
