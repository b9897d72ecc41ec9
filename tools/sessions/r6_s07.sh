#!/bin/bash
# Round 6, GPU session 7: the packed tally (one store per trip), store flavours (SC1: write-through, NT), against S03 / R5.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_wide_quads_gpu.py -m gpu -x -q 2>&1 | tail -3
run() {  # variant workload steps [env...]
  lib=""; [ "$1" != new ] && lib="$R/numbskull_amd/variants/libnsk_$1.so"
  echo -n "$2 $1 ${@:4} : "
  env NSK_LIB=$lib NSK_DIAG=1 "${@:4}" timeout 300 python bench.py --workload $2 --steps $3 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/sweep  parity %s' % (d['value'], d['ms_per_step']*1e3, d['parity'].get('ok')))"
}
for v in new SC1 NT; do run $v ising10m 200 X=1; run $v ising40m 100 X=1; run $v ising1m 400 X=1; done
for v in new SC1 NT; do run $v ising10m 200 NSK_NO_PACK_TALLY=1; done
run new ising10m 20 X=1; run S03 ising10m 200 X=1; run R5 ising10m 200 X=1; run new ising100m 50 X=1
