"""Does the driver's background clearing of freed VRAM slow a kernel launched right after a big free?
(companion of tools/diag_copy_rates.py)"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import rlshaders_amd as R
import bench
ctx = R.Context(0)
def timed(wl, reps=20):
    for _ in range(10): wl.launch()
    torch.cuda.synchronize(); ctx.timer_start()
    for _ in range(reps): wl.launch()
    ctx.timer_stop(); return ctx.timer_elapsed_ms() / reps
for name in ("ggx_pdf", "ggx_reflect_refract"):
    wl = bench.make_workload(R, ctx, name, 1 << 26, 0, 1); print(name, "baseline", round(timed(wl), 4), flush=True)
    for trial in range(2):
        big = torch.empty(1 << 33, dtype=torch.float32, device="cuda"); big.zero_(); torch.cuda.synchronize()
        del big; torch.cuda.empty_cache()
        t0 = time.time()
        for k in range(4):
            print(name, f"after freeing 32 GB, t={time.time()-t0:.2f}s", round(timed(wl, 10), 4), flush=True)
    del wl; torch.cuda.empty_cache()
