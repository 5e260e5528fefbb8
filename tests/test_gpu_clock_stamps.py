"""GPU: the diagnostic clock stamps (rls_diag_clock_stamps_*, include/rlshaders_amd_diag.h).  The stamped instantiation of each
BASELINE kernel must produce the product kernel's bits, stamp every workgroup it ran, and report a clock a gfx950 can hold;
outside begin/end nothing is stamped."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import rlshaders_amd as R

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


@pytest.mark.parametrize("name,log2n", [("ggx_reflect_refract", 22), ("sss_probe", 22), ("skin", 21), ("disney_integrate", 18)])
def test_stamped_instantiation_same_bits_and_plausible_clock(gpu, name, log2n):
    from bench_workloads import make_workload
    n = 1 << log2n
    wl = make_workload(R, gpu, name, n, first=0, candidates=1)
    outs = list(wl.outputs.values()) if isinstance(wl.outputs, dict) else list(wl.outputs)
    for _ in range(3):
        wl.launch()
    gpu.timer_start()
    wl.launch()
    gpu.timer_stop()
    plain_ms = gpu.timer_elapsed_ms()
    want = [R.checksum(gpu, t) for t in outs]
    for t in outs:
        t.zero_()
    gpu.clock_stamps_begin()
    try:
        gpu.timer_start()
        wl.launch()
        gpu.timer_stop()
        stamped_ms = gpu.timer_elapsed_ms()
        st = gpu.clock_stamps_read()
    finally:
        gpu.clock_stamps_end()
    assert [R.checksum(gpu, t) for t in outs] == want, "the stamped instantiation changed an output bit"
    clock = R.Context.clock_from_stamps(st)
    # every workgroup of the launch stamped its slot: the grid is min(tiles, cap) rounded up to a multiple of 8
    tiles = n // 256
    assert clock["workgroups"] >= min(tiles, 256 * 64) and clock["workgroups"] <= tiles + 8, clock
    assert (st[:, 1] > st[:, 0]).all() and (st[:, 3] > st[:, 2]).all()
    # MI355X: 2.4 GHz maximum; under vector load the chip holds 1.3-2.3 GHz
    assert 0.8 < clock["effective_clock_ghz"] <= 2.45, clock
    assert clock["p05_ghz"] > 0.5 and clock["p95_ghz"] < 2.6, clock
    # the 100 MHz counter: first entry to last exit is the launch's duration as the HIP events saw it (loosely: a launch of
    # a millisecond or less has tens of microseconds of ramp either side)
    assert 0.5 * stamped_ms < clock["span_ms"] < 1.3 * stamped_ms + 0.05, (clock, stamped_ms)
    # two reads of two counters per workgroup: the stamped launch takes what the product launch takes
    assert stamped_ms < 1.25 * plain_ms + 0.05, (stamped_ms, plain_ms)


def test_nothing_is_stamped_outside_begin_end(gpu):
    from bench_workloads import make_workload
    wl = make_workload(R, gpu, "ggx_reflect_refract", 1 << 18, first=0, candidates=1)
    with pytest.raises(R.RlsError):
        gpu.clock_stamps_read()                     # begin is not in force
    gpu.clock_stamps_begin()
    assert len(gpu.clock_stamps_read()) == 0        # begin clears the buffer
    wl.launch()
    first = gpu.clock_stamps_read()
    assert len(first) == (1 << 18) // 256
    gpu.clock_stamps_end()
    wl.launch()                                     # the product kernel: no stamp
    gpu.clock_stamps_begin()
    assert len(gpu.clock_stamps_read()) == 0
    # a kernel without a stamped instantiation launches as always and leaves the buffer empty
    wo, N, T = R.gen_frame(gpu, 7, 0, 1 << 16)
    g = R.GgxSampler(gpu, wo, N, T, specColor=(0.9, 0.8, 0.7), ior=1.5, roughness=0.3, anisotropic=0.0)
    xi = [R.gen_uniform(gpu, 7, 0, 1 << 16, 11 + j) for j in range(2)]
    g.sampleEvalPdf(xi[0], xi[1])
    assert len(gpu.clock_stamps_read()) == 0
    with pytest.raises(RuntimeError):
        R.Context.clock_from_stamps(gpu.clock_stamps_read())
    gpu.clock_stamps_end()


def test_read_returns_the_last_stamped_launch_only(gpu):
    """two stamped launches with different grids inside ONE begin/end bracket -- the slots the larger grid wrote
    must not come back with the smaller launch (they would skew the median clock and span_ms): every stamped launch clears
    the slots on its stream first"""
    from bench_workloads import make_workload
    big = make_workload(R, gpu, "ggx_reflect_refract", 1 << 20, first=0, candidates=1)
    small = make_workload(R, gpu, "sss_probe", 1 << 16, first=0, candidates=1)
    gpu.clock_stamps_begin()
    try:
        big.launch()
        assert len(gpu.clock_stamps_read()) == (1 << 20) // 256
        small.launch()
        st = gpu.clock_stamps_read()
        assert len(st) == (1 << 16) // 256, len(st)
        # ... and they are the small launch's: all entered after the big launch's read-back
        big.launch()
        later = gpu.clock_stamps_read()
        assert len(later) == (1 << 20) // 256 and later[:, 2].min() > st[:, 3].max()
    finally:
        gpu.clock_stamps_end()


def test_stamps_and_graph_recording_exclude_each_other():
    """a stamped launch recorded into a graph would keep writing stamps on every replay after _end, and _read
    inside a recording would synchronise a capturing stream"""
    from bench_workloads import make_workload
    ctx = R.Context(0, use_torch_stream=False)          # the context's own stream: the NULL stream cannot be captured
    try:
        wl = make_workload(R, ctx, "ggx_reflect_refract", 1 << 16, first=0, candidates=1)
        outs = list(wl.outputs.values()) if isinstance(wl.outputs, dict) else list(wl.outputs)
        ctx.clock_stamps_begin()
        with pytest.raises(R.RlsError, match="clock_stamps"):
            ctx.capture().__enter__()
        wl.launch()                                     # the refused recording left the context usable
        assert len(ctx.clock_stamps_read()) == (1 << 16) // 256
        ctx.clock_stamps_end()
        want = [R.checksum(ctx, t) for t in outs]
        with ctx.capture() as graph:
            with pytest.raises(R.RlsError):
                ctx.clock_stamps_begin()                # not while recording
            wl.launch()
        ctx.synchronize()
        for t in outs:
            t.zero_()
        torch.cuda.synchronize()                        # (zero_ ran on torch's stream, the graph runs on the context's own)
        graph.launch()
        ctx.synchronize()
        assert [R.checksum(ctx, t) for t in outs] == want
        # the recorded launch is the product kernel: replaying it inside a later bracket stamps nothing
        ctx.clock_stamps_begin()
        graph.launch()
        ctx.synchronize()
        assert len(ctx.clock_stamps_read()) == 0
        ctx.clock_stamps_end()
    finally:
        ctx.close()


def test_bad_arguments(gpu):
    import ctypes as C
    lib = R.load()
    count = C.c_int64()
    assert lib.rls_diag_clock_stamps_begin(None) != 0
    assert lib.rls_diag_clock_stamps_end(None) != 0
    assert lib.rls_diag_clock_stamps_read(gpu.handle, 0, None, None) != 0          # count is NULL
    gpu.clock_stamps_begin()
    assert lib.rls_diag_clock_stamps_read(gpu.handle, 4, None, C.byref(count)) != 0      # capacity without a buffer
    assert lib.rls_diag_clock_stamps_read(gpu.handle, -1, None, C.byref(count)) != 0
    assert lib.rls_diag_clock_stamps_read(gpu.handle, 0, None, C.byref(count)) == 0 and count.value > 0
    gpu.clock_stamps_end()
