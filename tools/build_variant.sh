#!/bin/bash
# tools/build_variant.sh NAME "EXTRA FLAGS" [TU]: a variant of the library whose inference (or learning) translation unit is
# compiled with extra flags (macros of experiments), into numbskull_amd/variants/libnsk_NAME.so for same-box A/B runs
# (NSK_LIB=... python bench.py; tools/ab_lib.sh).  The regular build must be up to date (the other objects come from it).
set -e
cd "$(dirname "$0")/../numbskull_amd/csrc"
mkdir -p ../variants build
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function ${NSK_PRELOAD--mllvm -amdgpu-kernarg-preload-count=16}"
v=$1; extra=$2; TU=${3:-gibbs}
LEARN="build/nsk_learn_p0.o build/nsk_learn_p1.o build/nsk_learn_p2.o build/nsk_learn_p3.o"
if [ "$TU" = learn ]; then
  for p in 0 1 2 3; do /opt/rocm/bin/hipcc $FLAGS $extra -DNSK_LEARN_PART=$p -c -o build/nsk_learn_p${p}_$v.o nsk_learn.hip & done; wait
  objs="build/nsk_gibbs.o build/nsk_learn_p0_$v.o build/nsk_learn_p1_$v.o build/nsk_learn_p2_$v.o build/nsk_learn_p3_$v.o"
else
  /opt/rocm/bin/hipcc $FLAGS $extra -c -o build/nsk_gibbs_$v.o nsk_gibbs.hip
  objs="build/nsk_gibbs_$v.o $LEARN"
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../variants/libnsk_$v.so build/nsk_api.o $objs build/nsk_compile.o build/nsk_host.o build/nsk_partition.o
echo built ../variants/libnsk_$v.so
