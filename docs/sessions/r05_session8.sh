#!/bin/bash
# round 5, GPU session 8: the tree as committed at the end of the round -- what the driver runs (pytest -m gpu -x, smoke, bench)
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r05_last_gputest.log 2>&1; tail -5 gpurun_out/r05_last_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py ) > gpurun_out/r05_last_bench.out 2> gpurun_out/r05_last_bench.err; tail -3 gpurun_out/r05_last_bench.err; tail -c 1600 gpurun_out/r05_last_bench.out
