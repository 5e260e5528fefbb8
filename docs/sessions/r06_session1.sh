#!/bin/bash
# round 6, GPU session 1: the new fail-fast / stamp tests, the default bench command timed and traced, then the round's profile
# pass of BASELINE configs 2-5 -- the first whose summaries carry the device-code ids natively (bench.py computes them on
# this box for the library it has loaded; rlshaders_amd/codeid.py) -- and the whole GPU suite.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m rlshaders_amd.codeid > gpurun_out/r06_library_id.json; head -4 gpurun_out/r06_library_id.json
( time python -m pytest tests/test_gpu_clock_stamps.py tests/test_gpu_bench_multirank.py tests/test_bench_headline.py tests/test_profile_binding.py -q ) > gpurun_out/r06_new_tests.log 2>&1; tail -6 gpurun_out/r06_new_tests.log
( time python bench.py ) > gpurun_out/r06_bench_default.out 2> gpurun_out/r06_bench_default.err; tail -3 gpurun_out/r06_bench_default.err; tail -c 1900 gpurun_out/r06_bench_default.out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_default_trace -- python3 bench.py --no-cpu-baseline > gpurun_out/r06_bench_default_traced.out 2> /dev/null
PROFILE_STEPS=200 bash tools/profile_workload.sh r06 ggx_reflect_refract --math exact
bash tools/profile_workload.sh r06 sss_probe --math exact --log2-points 25
bash tools/profile_workload.sh r06 skin --math exact --log2-points 27
CLOCK_ARGS=" " bash tools/profile_workload.sh r06 disney_integrate --math exact
STALL_STEPS=20 STALL_WARMUP=10 bash tools/pmc_stalls.sh r06 ggx_reflect_refract
STALL_STEPS=20 STALL_WARMUP=10 bash tools/pmc_stalls.sh r06 sss_probe --log2-points 25
STALL_STEPS=20 STALL_WARMUP=10 bash tools/pmc_stalls.sh r06 skin --log2-points 27
STALL_STEPS=20 STALL_WARMUP=10 bash tools/pmc_stalls.sh r06 disney_integrate
for w in ggx_reflect_refract sss_probe skin disney_integrate; do python3 tools/summarize_workload.py r06 $w; python3 tools/summarize_stalls.py r06 $w; done > gpurun_out/r06_summaries.log 2>&1
mkdir -p gpurun_out/profiles_r06; cp profiles/r06_* gpurun_out/profiles_r06/ 2>/dev/null
find gpurun_out/r06_default_trace -name '*kernel_stats.csv' -exec cp {} gpurun_out/profiles_r06/r06_bench_default_kernel_stats.csv \;
tail -40 gpurun_out/r06_summaries.log
# the raw counter CSVs are large: keep the summaries, drop the per-dispatch files before gpurun merges gpurun_out/ back
find gpurun_out -name '*counter_collection.csv' -size +2M -delete; find gpurun_out -name '*kernel_trace.csv' -size +2M -delete
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r06_gputest.log 2>&1; tail -6 gpurun_out/r06_gputest.log
du -sh gpurun_out
