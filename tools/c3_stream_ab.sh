#!/bin/bash
# Runs on the GPU box: C3 (bench.py --no-extra --no-cpu-baseline) with the regular library and with every variant of tools/c3_stream_build.sh, twice each,
# interleaved; prints us per iteration, launch time, iteration count and a checksum of the solution's bits.
set -u
REPO=$(pwd)
run() {   # tag, lib
  FDAPDE_HIP_LIB=$2 python3 bench.py --steps 10 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('%-8s us/iter %.2f  launch %.3f ms  iterations %d  ms/step %.3f  operator mean %.2f us slowest %.2f  gather %.2f  relres %.4e  err %.6e' % ('$1', c['us_per_iteration'], r['avg_launch_ms'], c['cg_iterations'], d['ms_per_step'], r['phase_stamps_us_per_iteration']['operator_mean'], r['phase_stamps_us_per_iteration']['operator_slowest_workgroup'], r['phase_stamps_us_per_iteration']['allgather'], c['relres'], c['max_abs_error_vs_analytic']))"
}
for rep in 1 2; do
  run base $REPO/fdapde-core_amd/lib/libfdapde_hip.so
  for t in u2 ntc ntv ntvc u2ntc; do run $t $REPO/tools/bin/variants/libfdapde_hip_$t.so; done
done
