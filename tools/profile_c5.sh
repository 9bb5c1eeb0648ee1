#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + separate FETCH_SIZE / WRITE_SIZE / L2 hit-miss passes of C5 on one GPU (tools/run_c5.py);
# writes gpurun_out/prof_<tag>_c5/summary.txt.   usage: tools/profile_c5.sh <tag>
set -u
TAG=${1:-r3}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}_c5
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/run_c5.py > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/tools/run_c5.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/tools/run_c5.py > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_l2 -- python3 $REPO/tools/run_c5.py > $OUT/pmc_l2.log 2>&1
cd $REPO
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
tail -3 $OUT/trace.log >> $OUT/summary.txt
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +8M -delete
