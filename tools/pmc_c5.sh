#!/bin/bash
# Runs on the GPU box (via gpurun): HBM traffic of the C5 kernels (separate FETCH_SIZE / WRITE_SIZE passes of tools/run_c5.py).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_c5pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/tools/run_c5.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/tools/run_c5.py > $OUT/pmc_write.log 2>&1
cd $REPO
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +8M -delete
