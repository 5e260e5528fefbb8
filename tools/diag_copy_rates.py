import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import rlshaders_amd as R
import bench
ctx = R.Context(0)
def rates(tag):
    p = R.Pipeline(ctx, 1024, 1, 1, 1)
    r = p.copy_rates(1 << 28); p.close()
    print(tag, {k: round(v, 1) for k, v in r.items()}, flush=True)
rates("fresh")
t = torch.empty(1 << 28, device="cuda"); torch.cuda.synchronize(); rates("after 1 GB torch alloc")
del t; torch.cuda.empty_cache(); rates("after empty_cache")
wl = bench.make_workload(R, ctx, "ggx_reflect_refract", 1 << 26, 0, 1); torch.cuda.synchronize(); rates("after arena 8 GB + gen")
for _ in range(5): wl.launch()
torch.cuda.synchronize(); rates("after launches")
del wl; torch.cuda.empty_cache(); rates("after free")
wl = bench.make_workload(R, ctx, "ggx_reflect_refract", 1 << 26, 0, 2); torch.cuda.synchronize(); rates("after arena with 2 candidates (probe)")
del wl; torch.cuda.empty_cache(); rates("after free 2")
import oracle_lib as O
bench.cpu_baseline("ggx_reflect", 1.0); rates("after cpu baseline")
wl = bench.make_workload(R, ctx, "disney_stream", 1 << 26, 0, 1); wl.launch(); torch.cuda.synchronize(); rates("after disney_stream"); del wl; torch.cuda.empty_cache()
wl = bench.make_workload(R, ctx, "skin", 1 << 27, 0, 1); wl.launch(); torch.cuda.synchronize(); rates("after skin 2^27"); del wl; torch.cuda.empty_cache(); rates("after free 3")
