#!/bin/bash
# Builds A/B variants of the library that differ in the symmetric streaming form of k_cg_persist only (kernels_persist.h FDAPDE_SYM_*): persist_engine.hip
# is recompiled with the switches, everything else is linked from the regular build.  -> tools/bin/variants/libfdapde_hip_<tag>.so (cross-compiled here,
# shipped to the GPU box by gpurun; run there with tools/c3_stream_ab.sh)
set -eu
cd "$(dirname "$0")/../fdapde-core_amd/csrc"
make -s -j8
OUT=../../tools/bin/variants
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-result -Wno-unused-function"
build() {   # tag, defines...
  tag=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c -o $OUT/persist_engine_$tag.o persist_engine.hip
  objs=$(ls ../build/*.o | grep -v persist_engine.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $OUT/libfdapde_hip_$tag.so $objs $OUT/persist_engine_$tag.o -lpthread
  rm -f $OUT/persist_engine_$tag.o
}
build u2 -DFDAPDE_SYM_U8=2 &
build ntc -DFDAPDE_SYM_NT_C=2 &
build ntv -DFDAPDE_SYM_NT_V=2 &
build ntvc -DFDAPDE_SYM_NT_V=2 -DFDAPDE_SYM_NT_C=2 &
build u2ntc -DFDAPDE_SYM_U8=2 -DFDAPDE_SYM_NT_C=2 &
wait
ls -la $OUT
