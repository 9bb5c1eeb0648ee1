#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + separate counter passes of the assembly alone (tools/asm_c5.py; ORDER / NX / TUNE in the
# environment select the workload).  Counters in passes of their own, never combined with a trace domain beyond --kernel-trace.
# usage: tools/profile_asm.sh <tag>      -> gpurun_out/prof_<tag>/summary.txt
set -u
TAG=${1:-r4_asm}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
FDAPDE_DEBUG_ASM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/asm_c5.py > $OUT/trace.log 2>&1
export REPS=1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/tools/asm_c5.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/tools/asm_c5.py > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $REPO/tools/asm_c5.py > $OUT/pmc_l2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq -- python3 $REPO/tools/asm_c5.py > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/tools/asm_c5.py > $OUT/pmc_sq2.log 2>&1
cd $REPO
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
grep -h "assembly launch\|asm order" $OUT/trace.log | sort | uniq -c >> $OUT/summary.txt
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +8M -delete
