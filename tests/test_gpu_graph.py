"""GPU: launch graphs (rls_graph_*): a recorded flush -- the GGX callback triple as three launches plus
the refract sample -- replays bit-identically on new data in the same buffers, and host-synchronising
calls are refused while recording."""
import time

import numpy as np
import pytest
import torch

import cases
import rlshaders_amd as R
from gpu_util import dev, host

pytestmark = pytest.mark.gpu


def test_graph_replay_matches_direct_launches(oracle):
    n = 1 << 14
    ctx = R.Context(0, use_torch_stream=False)          # the context's own stream: the NULL stream cannot be captured
    try:
        c = cases.ggx_mixed(cases.SEED_PARITY, n)
        x = cases.xi(cases.SEED_PARITY, n, 4)
        rx, ry, rx2, ry2 = (dev(x[k]) for k in range(4))
        g = R.GgxSampler(ctx, dev(c["wo"]), dev(c["N"]), dev(c["T"]), specColor=dev(c["KsColor"]), ior=dev(c["ior"]),
                         roughness=dev(c["roughness"]), anisotropic=dev(c["anisotropic"]))
        wi, F, f, pdf = ctx.empty(3, n), ctx.empty(n), ctx.empty(3, n), ctx.empty(n)
        wt, w, flag = ctx.empty(3, n), ctx.empty(n), torch.empty(n, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()

        def flush():
            g.evalSample(rx, ry, out_wi=wi, out_fresnel=F)
            g.evalBrdf(wi, out=f)
            g.evalPdf(wi, out=pdf)
            g.refractSample(rx2, ry2, out=(wt, w, flag))

        with ctx.capture() as graph:
            flush()
        # recording runs nothing
        for t in (wi, f, pdf, wt):
            t.zero_()
        torch.cuda.synchronize()
        graph.launch()
        ctx.synchronize()
        got = [host(t).copy() for t in (wi, F, f, pdf, wt, w)]
        flush()
        ctx.synchronize()
        for a, t in zip(got, (wi, F, f, pdf, wt, w)):
            assert np.array_equal(a.view(np.uint32), host(t).view(np.uint32))
        # and against the oracle
        ref = oracle.Ggx(c["wo"], c["N"], c["T"], KsColor=c["KsColor"], ior=c["ior"], roughness=c["roughness"],
                         anisotropic=c["anisotropic"], nthreads=4).sample_eval_pdf(x[0], x[1])
        cases.assert_tight(cases.summarize(cases.rel_err(got[2], ref[1])), "graph f")

        # new random numbers in the same buffers: the replay picks them up
        x2 = cases.xi(cases.SEED_EDGE, n, 4)
        for t, k in ((rx, 0), (ry, 1), (rx2, 2), (ry2, 3)):
            t.copy_(dev(x2[k]))
        torch.cuda.synchronize()
        graph.launch()
        ctx.synchronize()
        replay = host(wi).copy()
        flush()
        ctx.synchronize()
        assert np.array_equal(replay.view(np.uint32), host(wi).view(np.uint32))
        assert not np.array_equal(replay, got[0])

        # launch-bound regime: one graph launch against four kernel launches
        def timed(fn, reps=200):
            fn(); ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            ctx.synchronize()
            return (time.perf_counter() - t0) / reps * 1e6
        t_direct, t_graph = timed(flush), timed(graph.launch)
        print(f"flush of 4 launches on {n} points: direct {t_direct:.1f} us, graph replay {t_graph:.1f} us")
        # timings are printed, not asserted: wall-clock on a shared box says nothing about correctness
        graph.close()
    finally:
        ctx.close()


def test_capture_rules():
    ctx = R.Context(0)                                  # torch's current stream = the NULL stream
    try:
        with pytest.raises(R.RlsError):
            with ctx.capture():
                pass
        assert ctx.lib.rls_graph_begin_capture(ctx.handle) == 1      # RLS_ERR_INVALID_ARGUMENT
        ctx.use_stream(None)
        assert ctx.lib.rls_graph_begin_capture(ctx.handle) == 0
        assert ctx.lib.rls_graph_begin_capture(ctx.handle) == 1      # already recording
        assert ctx.lib.rls_context_synchronize(ctx.handle) == 1      # host sync refused while recording
        import ctypes as C
        h = C.c_void_p()
        assert ctx.lib.rls_graph_end_capture(ctx.handle, C.byref(h)) == 0
        assert ctx.lib.rls_graph_launch(ctx.handle, h) == 0          # an empty graph is launchable
        ctx.synchronize()
        ctx.lib.rls_graph_destroy(h)
        assert ctx.lib.rls_graph_end_capture(ctx.handle, C.byref(h)) == 1   # nothing in progress
    finally:
        ctx.close()
