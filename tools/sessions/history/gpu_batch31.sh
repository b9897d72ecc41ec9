#!/bin/bash
# learning table kernel: parity of the grouped-load build and of the packed-count variant, then the
# stage ablations at HEAD (variants from NSK_ABL_TU=learn tools/build_ablations.sh ...)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
V=$R/numbskull_amd/variants
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_config3_gpu.py -m gpu -x -q -k "learn" > gpurun_out/b31_main.log 2>&1; echo "main learn parity rc $?"; tail -2 gpurun_out/b31_main.log
NSK_LIB=$V/libnsk_LPACKED.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_config3_gpu.py -m gpu -x -q -k "learn" > gpurun_out/b31_packed.log 2>&1; echo "packed learn parity rc $?"; tail -2 gpurun_out/b31_packed.log
for v in full BASE LPACKED NOAPPLY NOPHILOX LNOEV LNOSINK LNOBALLOT LNOEVST NOPHILOX+LNOEV+LNOSINK+LNOBALLOT+LNOINIT+LNOEVST; do
  lib=""; [ "$v" != full ] && lib="$V/libnsk_$v.so"
  echo -n "ising10m_learn variant=$v "
  NSK_LIB=$lib python bench.py --workload ising10m_learn --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e updates/s  %.2f us/class' % (d['value'], d['roofline']['avg_launch_us']))"
done
