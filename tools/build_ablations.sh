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
# (the learning unit is compiled in four parts like the Makefile does; NSK_ABL_PARTS="0": only the int8 /
# large-table instantiation is rebuilt -- the LR graphs -- and the others come from the regular build)
PARTS=${NSK_ABL_PARTS:-0 1 2 3}
for v in "$@"; do
  defs=""; for d in ${v//+/ }; do defs="$defs -DNSK_ABL_$d"; done
  if [ "$TU" = learn ]; then
    for p in $PARTS; do /opt/rocm/bin/hipcc $FLAGS $defs -DNSK_LEARN_PART=$p -c -o build/nsk_learn_p${p}_$v.o nsk_learn.hip & done
  else
    /opt/rocm/bin/hipcc $FLAGS $defs -c -o build/nsk_${TU}_$v.o nsk_$TU.hip &
  fi
done
wait
LEARN="build/nsk_learn_p0.o build/nsk_learn_p1.o build/nsk_learn_p2.o build/nsk_learn_p3.o"
for v in "$@"; do
  if [ "$TU" = learn ]; then
    objs="build/nsk_gibbs.o"
    for p in 0 1 2 3; do if [[ " $PARTS " == *" $p "* ]]; then objs="$objs build/nsk_learn_p${p}_$v.o"; else objs="$objs build/nsk_learn_p$p.o"; fi; done
  else objs="build/nsk_gibbs_$v.o $LEARN"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../variants/libnsk_$v.so build/nsk_api.o $objs build/nsk_compile.o build/nsk_host.o
done
