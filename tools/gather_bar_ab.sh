#!/bin/bash
# A/B on ONE box: the record sums behind ONE barrier (default build) against three (tools/wave_sum_build.sh bar3), C2 and C3, alternating
export TMPDIR=/tmp
for rep in 1 2; do
  for lib in "" tools/bin/variants/libfdapde_hip_${VARIANT:-bar3}.so; do
    echo "== lib: ${lib:-default}"
    FDAPDE_HIP_LIB=$lib timeout 200 python tools/run_c2.py 2>&1 | tail -1
    FDAPDE_HIP_LIB=$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C3', d['value'], 'ms/step', d['ms_per_step'], 'us/it', d['config']['us_per_iteration'], 'iters', d['config']['cg_iterations'], 'relres', d['config']['relres'])"
  done
done
