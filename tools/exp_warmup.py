import sys, time
sys.path.insert(0, "/root/repo")
import torch, rlshaders_amd as R
import bench
ctx = R.Context(0)
n = 1 << 26
wl = bench.make_workload(R, ctx, "ggx_reflect_refract", n, first=0, candidates=int(sys.argv[1]))
torch.cuda.synchronize()
def one():
    ctx.timer_start(); wl.launch(); ctx.timer_stop(); return ctx.timer_elapsed_ms()
if len(sys.argv) > 2 and sys.argv[2] == "prefast":
    ctx.set_math_mode(True)
    t = [one() for _ in range(6)]
    print("fast first:", " ".join(f"{x:.2f}" for x in t))
    ctx.set_math_mode(False)
if len(sys.argv) > 2 and sys.argv[2].startswith("probe"):
    import ctypes as C
    ms = float(sys.argv[2][5:] or 60)
    scratch = torch.empty(31 * (1 << 23), dtype=torch.float32, device="cuda")
    g = C.c_float()
    t0 = time.perf_counter()
    k = 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        R._capi.check(ctx.lib.rls_probe_block(ctx.handle, C.c_void_p(scratch.data_ptr()), scratch.numel() * 4, C.byref(g)))
        k += 1
    print(f"preheat: {k} probe calls in {(time.perf_counter() - t0) * 1e3:.0f} ms")
print("exact launches:", " ".join(f"{one():.2f}" for _ in range(14)))
