#!/bin/bash
# Where the waves of a workload's dominant kernel spend their cycles: busy / wait / issue counters of the SQ per launch, BY
# EXACT KERNEL NAME (the name bench.py prints as roofline.kernel -- `ggx_kernel<5, 0, 1>` is the EXACT kernel, `<5, 1, 1>`
# the FAST one, `ggx_kernel_stamped<...>` the diagnostic instantiation: neither of the latter two is counted), with the
# profiled command launching nothing but the measured kernel (--no-other-mode --no-clock).
# usage: tools/pmc_stalls.sh <tag> <workload> [bench args ...]   -> gpurun_out/stalls_<tag>_<workload>/ ;
#        tools/summarize_stalls.py <tag> <workload> turns that into profiles/<tag>_<workload>_stalls.json
set -u
TAG=$1; W=$2; shift 2
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
d=$R/gpurun_out/stalls_${TAG}_$W; rm -rf $d; mkdir -p $d
# STALL_STEPS / STALL_WARMUP: 3 / 1 for a quick look; 20 / 10 gives the counters of warm launches at the clock the kernel settles at
SHORT="python3 bench.py --workload $W --steps ${STALL_STEPS:-3} --warmup ${STALL_WARMUP:-1} --no-cpu-baseline --no-other-mode --no-clock --arena-candidates 1 --workloads none $*"
$SHORT > $d/bench.json 2> $d/bench.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $d/a -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $d/b -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --output-format csv -d $d/c -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $d/e -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $d/f -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $d/g -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM SQ_IFETCH --output-format csv -d $d/h -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $d/i -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_SMEM SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL --output-format csv -d $d/j -- $SHORT > /dev/null 2>&1
echo "$W: $(find $d -name '*counter_collection.csv' | wc -l) counter files; $(tail -c 200 $d/bench.err)"
