#!/bin/bash
# round 6, GPU session 3: the round's final tree (code-path switches removed, fixed -cuid per unit: device code unchanged by
# hash) -- the whole GPU suite, smoke, the default bench command and the driver's command
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m rlshaders_amd.codeid | head -4
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r06_gputest.log 2>&1; tail -5 gpurun_out/r06_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time python bench.py ) > gpurun_out/r06_bench_final.out 2> gpurun_out/r06_bench_final.err; tail -c 1700 gpurun_out/r06_bench_final.out
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_bench_driver_cmd.out 2> gpurun_out/r06_bench_driver_cmd.err; tail -c 500 gpurun_out/r06_bench_driver_cmd.out
