#!/bin/bash
# round 4, first GPU session: the new bench line's tests, the default bench command timed, and the rlSss latency experiments
# (occupancy sweep + two tiles per iteration) -> gpurun_out/r04_*
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_materials_by_reference.py -x -q -m gpu > gpurun_out/r04_t1.log 2>&1
tail -5 gpurun_out/r04_t1.log
( time python bench.py ) > gpurun_out/r04_bench_default.out 2> gpurun_out/r04_bench_default.err
tail -c 1900 gpurun_out/r04_bench_default.out; tail -4 gpurun_out/r04_bench_default.err
for W in sss_probe nd_sample; do
  echo "== $W" >> gpurun_out/r04_sss_two_points.txt
  bash tools/ab.sh $W sssw2 sssw3 sssw4 sssw6 ssstwo8 ssstwo6 ssstwo5 ssstwo4 >> gpurun_out/r04_sss_two_points.txt 2>&1
done
cat gpurun_out/r04_sss_two_points.txt
