#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py; writes gpurun_out/prof_<tag>/.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT   # (passes of an earlier run must not end up in this run's summary)
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extra $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
PARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-extra --time-spmv 0 $*"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $PARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $PARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $REPO/bench.py $PARGS > $OUT/pmc_l2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $PARGS > $OUT/pmc_sq.log 2>&1
cd $REPO
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merge small: drop the raw per-dispatch csvs beyond the summary inputs
find $OUT -name "*.csv" -size +8M -delete
