#!/bin/bash
# A/B of two builds of the library on ONE box, alternating: the regular build against tools/bin/variants/libfdapde_hip_$VARIANT.so (default: head = the
# commit before, built from a scratch worktree); C2, C3, the wide form's system and a mid-size 3-D system through tools/knob_ab.py (knob left alone)
export TMPDIR=/tmp
V=tools/bin/variants/libfdapde_hip_${VARIANT:-head}.so
for rep in 1 2; do
  for lib in "" $V; do
    echo "== lib: ${lib:-regular build}"
    FDAPDE_HIP_LIB=$lib timeout 400 python tools/knob_ab.py persist_time 0 0 2 2>&1 | grep "persist_time=0" | awk 'NR%2==1' | sed 's/persist_time=0: //'
  done
done
