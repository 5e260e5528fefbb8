#!/bin/bash
# Profile one bench workload on the GPU box: un-profiled line, rocprofv3 kernel trace + stats, HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, FETCH_SIZE calibrated on a known-size kernel) and the dynamic
# instruction mix.  usage: tools/profile_workload.sh <tag> <workload> [bench args ...]      (e.g. r02 skin --math exact)
# Writes gpurun_out/prof_<tag>_<workload>/ ; tools/summarize_workload.py turns that into profiles/<tag>_<workload>_*.
set -u
TAG=$1; W=$2; shift 2
R=$PWD
OUT=$R/gpurun_out/prof_${TAG}_$W
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $R
STEPS=${PROFILE_STEPS:-40}
FULL="python3 bench.py --workload $W --steps $STEPS --warmup 10 $*"
SHORT="python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-clock --arena-candidates 1 $*"
$FULL > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $FULL --no-cpu-baseline > $OUT/bench_trace.json 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- python3 tools/calibrate_fetch.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/mix_a -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/mix_b -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU --output-format csv -d $OUT/mix_c -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/mix_d -- $SHORT > /dev/null 2>&1
# the shader clock: (a) GRBM_GUI_ACTIVE / 8 / dispatch duration -- trustworthy on dispatches of 10 ms or more, so the pointwise
# kernels get a batch that makes one (CLOCK_LOG2, default: 2^28 points; the n^2-spp integrator is long enough as it is); (b) the
# in-kernel stamps after two seconds of back-to-back launches (bench.py --sustain-seconds 2: roofline.clock of that line)
CLOCK_ARGS=${CLOCK_ARGS:---log2-points 28}
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/clock_grbm -- $SHORT $CLOCK_ARGS > $OUT/clock_grbm.json 2> /dev/null
python3 bench.py --workload $W --steps $STEPS --warmup 10 --sustain-seconds 2 --no-cpu-baseline --no-other-mode --arena-candidates 1 $* > $OUT/clock_sustained.json 2> /dev/null
echo "$W: $(find $OUT -name '*.csv' | wc -l) csv files; $(tail -c 300 $OUT/bench.err)"
