#!/bin/bash
# round 5, GPU session 2: the instruction-class costs in TRUE cycles (valurate with in-kernel clock stamps) and what each class
# does to the SQ_ACTIVE_INST_VALU / VALU2 counters; the by-name stall counters again on warm, steady launches; rlSss probe
# against the workgroup count per CU at the 8-GPU shard size; the whole GPU suite; the default bench command timed
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
( cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o valurate valurate.hip && ./valurate ) > gpurun_out/r05_valurate.txt 2>&1; tail -32 gpurun_out/r05_valurate.txt
rm -rf gpurun_out/valurate_pmc; rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_CYCLES --output-format csv -d gpurun_out/valurate_pmc -- tools/micro/valurate 8 > /dev/null 2>&1
python3 tools/summarize_valurate_pmc.py gpurun_out/valurate_pmc > gpurun_out/r05_valurate_pmc.txt 2>&1; cat gpurun_out/r05_valurate_pmc.txt
export STALL_STEPS=20 STALL_WARMUP=10
bash tools/pmc_stalls.sh r05 ggx_reflect_refract
bash tools/pmc_stalls.sh r05 sss_probe --log2-points 25
bash tools/pmc_stalls.sh r05 skin --log2-points 27
bash tools/pmc_stalls.sh r05 disney_integrate
for w in ggx_reflect_refract sss_probe skin disney_integrate; do python3 tools/summarize_stalls.py r05 $w; done > gpurun_out/r05_stalls.log 2>&1
mkdir -p gpurun_out/profiles_r05; cp profiles/r05_*_stalls.json gpurun_out/profiles_r05/
grep -h "valu_port\|\"workload\"\|grbm_clock_ghz\|kernel_ms_profiled" gpurun_out/r05_stalls.log
for b in 8 16 32 64 128 256; do echo "RLS_BLOCKS_PER_CU=$b"; for rep in 1 2; do RLS_BLOCKS_PER_CU=$b python3 bench.py --config 4 --steps 100 --warmup 20 --no-cpu-baseline --no-other-mode --arena-candidates 1 2>/dev/null | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('  ', r['roofline']['kernel_ms'], r['roofline'].get('effective_clock_ghz'))"; done; done > gpurun_out/r05_sss_blocks_per_cu.txt 2>&1; cat gpurun_out/r05_sss_blocks_per_cu.txt
( time python -m pytest tests -m gpu -q ) > gpurun_out/r05_gputest.log 2>&1; tail -6 gpurun_out/r05_gputest.log
( time python bench.py ) > gpurun_out/r05_bench_default.out 2> gpurun_out/r05_bench_default.err; tail -3 gpurun_out/r05_bench_default.err; tail -c 2100 gpurun_out/r05_bench_default.out
