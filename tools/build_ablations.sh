#!/bin/bash
# Builds ablated variants of the library (a sweep kernel with one stage removed, or instrumented)
# into numbskull_amd/variants/ for timing experiments on the GPU box (NSK_LIB=... python bench.py).
# The variants compute wrong samples by construction; they exist to price the stages.
# usage: tools/build_ablations.sh NAME[+NAME...] ...   (each NAME defines NSK_ABL_<NAME>)
set -e
cd "$(dirname "$0")/../numbskull_amd/csrc"
mkdir -p ../variants build
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function ${NSK_ABL_EXTRA:-}"
# NSK_ABL_TU=learn builds the variants of the learning translation unit instead of the inference one
TU=${NSK_ABL_TU:-gibbs}
for v in "$@"; do
  defs=""; for d in ${v//+/ }; do defs="$defs -DNSK_ABL_$d"; done
  /opt/rocm/bin/hipcc $FLAGS $defs -c -o build/nsk_${TU}_$v.o nsk_$TU.hip &
done
wait
for v in "$@"; do
  if [ "$TU" = learn ]; then objs="build/nsk_gibbs.o build/nsk_learn_$v.o"; else objs="build/nsk_gibbs_$v.o build/nsk_learn.o"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../variants/libnsk_$v.so build/nsk_api.o $objs build/nsk_compile.o build/nsk_host.o
done
