#!/bin/bash
# The round's profile pass on one GPU box: every workload's un-profiled bench line (roofline + CPU baseline), rocprofv3
# kernel stats, PMC traffic and instruction mix (tools/profile_workload.sh), then bench-only lines of the smaller ones.
# usage: tools/profile_all.sh <tag>      -> gpurun_out/prof_<tag>_*; tools/summarize_workload.py <tag> <workload> afterwards
TAG=${1:-r03}
PROFILE_STEPS=200 bash tools/profile_workload.sh $TAG ggx_reflect_refract --math exact
for w in skin sss_probe disney_integrate disney_stream ggx_reflect_refract_uniform ggx_reflect ggx_eval ggx_pdf disney_triple_diffuse disney_triple_glossy nd_sample \
         disney_triple_glossy_uniform disney_triple_glossy_colour_map sss_probe_uniform skin_uniform \
         skin_integrate ggx_shade disney_shade disney_direct ggx_direct sss_scatter; do
  bash tools/profile_workload.sh $TAG $w --math exact
done
mkdir -p gpurun_out/bench_only
python3 bench.py --math fast --steps 40 --warmup 10 > gpurun_out/bench_only/${TAG}_bench_fast.json 2>/dev/null
