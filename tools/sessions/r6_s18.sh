#!/bin/bash
# Round 6, GPU session 18: the wide-quad kernel with its trips software-pipelined (-DNSK_TABW_PIPE: the next quad's
# value dwords requested before this quad's are consumed) against the shipped body, same box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
NSK_LIB=$R/numbskull_amd/variants/libnsk_PIPE.so timeout 900 python -m pytest tests/test_wide_quads_gpu.py tests/test_config3_gpu.py -m gpu -x -q 2>&1 | tail -3
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  parity %s' % (d['value'], d['ms_per_step']*1e3, d['parity'].get('ok')))"
}
for v in new PIPE new PIPE; do run $v ising10m 200 X=1; done
for v in new PIPE; do run $v ising40m 100 X=1; run $v ising100m 40 X=1; run $v ising4m 200 NSK_WIDE_MIN=0; run $v ising1m 400 NSK_WIDE_MIN=0; done
for cap in 1024 1280 1536 2048; do run PIPE ising10m 200 NSK_TABW_GRID_CAP=$cap; done
