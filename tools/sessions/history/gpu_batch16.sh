#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1800 python -m pytest tests -m gpu -x -q --deselect tests/test_config5_gpu.py --deselect tests/test_learning_tie_gpu.py::test_many_weight_lr_graph_chromatic_matches_sequential > gpurun_out/b16_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/b16_pytest.log
python bench.py > gpurun_out/b16_bench_default.json 2> gpurun_out/b16_bench_default.err
echo "bench rc $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/b16_bench_default.json").read().strip().splitlines()[-1])
print("10M: %.4e  %.2f us/launch frac %.3f layoutB %.2f" % (d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"]["layout_bytes_per_update"]))
for k,v in d["also"].items(): print(k, "%.4e" % v["value"], "us/launch %.2f frac %.3f" % (v["avg_launch_us"], v["roofline_frac"]), v.get("learn_clipped"))
print(d["parity"]); print(d.get("cpu_baseline"))
PY
echo -n "10M NO_AFFINE: "; NSK_DIAG=1 NSK_NO_AFFINE=1 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch frac %.3f' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac']))"
for wl in ising40m ising1m; do for na in 0 1; do
echo -n "$wl NO_AFFINE=$na: "; NSK_DIAG=1 $( [ $na = 1 ] && echo NSK_NO_AFFINE=1 ) python bench.py --workload $wl --steps 50 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e updates/s  %.2f us/launch frac %.3f' % (d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac']))"
done; done
NSK_BENCH_ONE_DEVICE=1 NSK_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2 ranks one device: %.4e' % d['value'], d.get('phases_us'))"
