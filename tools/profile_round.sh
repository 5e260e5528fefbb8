#!/bin/bash
# Profile the headline kernel on the GPU box.  usage: tools/profile_round.sh <tag>   (e.g. r01)
# Writes rocprofv3 CSVs under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into
# the committed summaries under profiles/.
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $R
BENCH="python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline"
# the un-profiled run the numbers are compared with
$BENCH > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.json 2> /dev/null
# PMC passes, one counter group each (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib -- python3 tools/calibrate_fetch.py > /dev/null 2>&1
find $OUT -name "*.csv" | wc -l
