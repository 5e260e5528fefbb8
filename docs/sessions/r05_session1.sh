#!/bin/bash
# round 5, GPU session 1: the clock stamps (new), parity after the kernel-body refactor, the default bench command timed, then
# the round's profile pass of BASELINE configs 2-5 with the clock passes and the by-name stall counters
mkdir -p gpurun_out
( time python -m pytest tests/test_gpu_clock_stamps.py -x -q ) > gpurun_out/r05_clock_test.log 2>&1; tail -4 gpurun_out/r05_clock_test.log
( time python bench.py ) > gpurun_out/r05_bench_default.out 2> gpurun_out/r05_bench_default.err; tail -3 gpurun_out/r05_bench_default.err; tail -c 1900 gpurun_out/r05_bench_default.out
PROFILE_STEPS=200 bash tools/profile_workload.sh r05 ggx_reflect_refract --math exact
bash tools/profile_workload.sh r05 sss_probe --math exact --log2-points 25
bash tools/profile_workload.sh r05 skin --math exact --log2-points 27
CLOCK_ARGS=" " bash tools/profile_workload.sh r05 disney_integrate --math exact
bash tools/pmc_stalls.sh r05 ggx_reflect_refract
bash tools/pmc_stalls.sh r05 sss_probe --log2-points 25
bash tools/pmc_stalls.sh r05 skin --log2-points 27
bash tools/pmc_stalls.sh r05 disney_integrate
for w in ggx_reflect_refract sss_probe skin disney_integrate; do python3 tools/summarize_workload.py r05 $w; python3 tools/summarize_stalls.py r05 $w; done > gpurun_out/r05_summaries.log 2>&1
mkdir -p gpurun_out/profiles_r05; cp profiles/r05_* gpurun_out/profiles_r05/ 2>/dev/null
tail -60 gpurun_out/r05_summaries.log
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r05_gputest.log 2>&1; tail -6 gpurun_out/r05_gputest.log
