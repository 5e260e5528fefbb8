#!/bin/bash
# round 5, GPU session 4: the round's final library (csrc/integrate.hip split into four units) -- what the driver runs at round
# end (pytest -m gpu, smoke, the default bench command), a parity soak on it, and the round's profile of the headline command
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r05_final_gputest.log 2>&1; tail -5 gpurun_out/r05_final_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time python bench.py ) > gpurun_out/r05_final_bench.out 2> gpurun_out/r05_final_bench.err; tail -3 gpurun_out/r05_final_bench.err; tail -c 1700 gpurun_out/r05_final_bench.out
python3 tools/parity_soak.py --out gpurun_out/r05_parity_soak_final.json --seeds 6001,6002 > gpurun_out/r05_parity_soak_final.log 2>&1; tail -4 gpurun_out/r05_parity_soak_final.log
python3 tools/parity_soak.py --out gpurun_out/r05_parity_soak_final_groups.json --seeds 6003 --groups 4,16,64 --spp-n 3 --log2-points 20 > gpurun_out/r05_parity_soak_final_groups.log 2>&1; tail -4 gpurun_out/r05_parity_soak_final_groups.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_final_trace -- python3 bench.py --no-cpu-baseline > gpurun_out/r05_final_bench_traced.out 2>/dev/null; f=$(find gpurun_out/r05_final_trace -name '*kernel_stats.csv' | head -1); head -8 $f | cut -c1-200
