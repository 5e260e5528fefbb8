#!/bin/bash
# round 6, GPU session 2: the multi-rank failure-shape tests after "every refusing rank prints its line", ONE short soak of the
# round's library (host-side changes only in csrc/; the device code is round 5's by hash -- this is the record, not a search),
# and the default bench command with the round's stamped counter files in place
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time python -m pytest tests/test_gpu_bench_multirank.py tests/test_bench_headline.py tests/test_profile_binding.py -q ) > gpurun_out/r06_s2_tests.log 2>&1; tail -5 gpurun_out/r06_s2_tests.log
( time python tools/parity_soak.py --log2-points 24 --seeds 606,607,608,609,610,611 --out gpurun_out/r06_parity_soak.json ) > gpurun_out/r06_parity_soak.log 2>&1; tail -4 gpurun_out/r06_parity_soak.log
( time python bench.py ) > gpurun_out/r06_bench_final.out 2> gpurun_out/r06_bench_final.err; tail -2 gpurun_out/r06_bench_final.err; tail -c 1900 gpurun_out/r06_bench_final.out
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_bench_driver_cmd.out 2> gpurun_out/r06_bench_driver_cmd.err; tail -c 700 gpurun_out/r06_bench_driver_cmd.out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8
