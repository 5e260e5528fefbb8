#!/bin/bash
# round 5, GPU session 3: the tests added after session 2 (multi-rank line with cpu_baseline, the border test's loose branch), a
# parity soak of the round's library (the kernel bodies were split out of their __global__ functions: 0 differing words demanded),
# and one attempt at rocprofv3's stochastic PC sampling on config 2 (stall reasons per sampled wave; beta -- under a timeout)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
( time python -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_clock_stamps.py -q ) > gpurun_out/r05_s3_tests.log 2>&1; tail -5 gpurun_out/r05_s3_tests.log
( RLS_TEST_LOOSE=1 python -m pytest tests/test_gpu_disney_config3.py -q -k borders ) > gpurun_out/r05_s3_loose.log 2>&1; tail -3 gpurun_out/r05_s3_loose.log
python3 tools/parity_soak.py --out gpurun_out/r05_parity_soak.json --seeds 5001,5002,5003,5004 > gpurun_out/r05_parity_soak.log 2>&1; tail -4 gpurun_out/r05_parity_soak.log
python3 tools/parity_soak.py --out gpurun_out/r05_parity_soak_uniform.json --uniform-draws 16 --seeds 5005,5006 > gpurun_out/r05_parity_soak_uniform.log 2>&1; tail -4 gpurun_out/r05_parity_soak_uniform.log
rm -rf gpurun_out/pcsamp; ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 180 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 --kernel-trace --output-format csv -d gpurun_out/pcsamp -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-mode --no-clock --workloads none --arena-candidates 1 > gpurun_out/r05_pcsamp.out 2> gpurun_out/r05_pcsamp.err; echo "pc sampling rc=$?"; tail -5 gpurun_out/r05_pcsamp.err; find gpurun_out/pcsamp -type f | head; 
if [ -z "$(find gpurun_out/pcsamp -name '*pc_sampling*' 2>/dev/null | head -1)" ]; then rm -rf gpurun_out/pcsamp; ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 180 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 100 --kernel-trace --output-format csv -d gpurun_out/pcsamp -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-mode --no-clock --workloads none --arena-candidates 1 > gpurun_out/r05_pcsamp2.out 2> gpurun_out/r05_pcsamp2.err; echo "host_trap rc=$?"; tail -5 gpurun_out/r05_pcsamp2.err; find gpurun_out/pcsamp -type f | head; fi
for f in $(find gpurun_out/pcsamp -name '*pc_sampling*csv' 2>/dev/null); do echo $f; head -3 $f; wc -l $f; done
