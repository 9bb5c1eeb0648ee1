#!/bin/bash
# A/B: the library with the all-shuffle wave_sum64 (kernels_persist.h FDAPDE_WAVE_SUM_SHFL) -> tools/bin/variants/libfdapde_hip_shfl.so
set -eu
cd "$(dirname "$0")/../fdapde-core_amd/csrc"
make -s -j8
OUT=../../tools/bin/variants
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-result -Wno-unused-function"
objs=$(ls ../build/*.o | grep -v persist_engine.o)
build() {   # tag, defines...
  tag=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c -o $OUT/persist_engine_$tag.o persist_engine.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $OUT/libfdapde_hip_$tag.so $objs $OUT/persist_engine_$tag.o -lpthread
  rm -f $OUT/persist_engine_$tag.o
}
build shfl -DFDAPDE_WAVE_SUM_SHFL -DFDAPDE_GATHER_3BAR &
build bar3 -DFDAPDE_GATHER_3BAR &
build stride6 -DFDAPDE_DOT_STRIDE6 &
wait
ls -la $OUT
