#!/bin/bash
# round 4, GPU session 6: the round's profile pass (configs 2-5 + NDProfile) and the whole GPU test suite
mkdir -p gpurun_out
PROFILE_STEPS=200 bash tools/profile_workload.sh r04 ggx_reflect_refract --math exact
for w in sss_probe skin disney_integrate nd_sample; do bash tools/profile_workload.sh r04 $w --math exact; done
for w in ggx_reflect_refract sss_probe skin disney_integrate nd_sample; do python3 tools/summarize_workload.py r04 $w; done
mkdir -p gpurun_out/profiles_r04; cp profiles/r04_*_bench*.json profiles/r04_*_kernel_stats.csv profiles/r04_*_traffic.json profiles/r04_*_flops.json gpurun_out/profiles_r04/ 2>/dev/null
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r04_gputest.log 2>&1; tail -6 gpurun_out/r04_gputest.log
