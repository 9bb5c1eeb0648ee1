#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace (+ optional counter passes) of the FIRST-CALL path of one workload -> gpurun_out/prof_<tag>/.
# usage: tools/profile_setup.sh <tag> [c3|c2|c5] [pmc]
set -u
TAG=${1:-setup}; CASE=${2:-c3}; PMC=${3:-}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
FDAPDE_DEBUG_SETUP=1 python3 $REPO/tools/first_call.py $CASE 3 > $OUT/first_call.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/first_call.py $CASE 1 > $OUT/trace.log 2>&1
if [ -n "$PMC" ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/tools/first_call.py $CASE 1 > $OUT/pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/tools/first_call.py $CASE 1 > $OUT/pmc_write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $REPO/tools/first_call.py $CASE 1 > $OUT/pmc_sq.log 2>&1
fi
cd $REPO
SUMMARY_ALL_KERNELS=1 python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
grep "^==" $OUT/first_call.log
head -60 $OUT/summary.txt
find $OUT -name "*.csv" -size +8M -delete
