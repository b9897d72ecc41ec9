for t in 1 2 4; do echo "TPW=$t"; NSK_TPW=$t python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['stream_copy_GBs'])"; done
NSK_TPW=2 python bench.py --workload ising1m --steps 1000 --warmup 100 --no-cpu-baseline --no-extra 2>&1 | tail -1 | cut -c1-200
NSK_TPW=1 python bench.py --workload ising1m --steps 1000 --warmup 100 --no-cpu-baseline --no-extra 2>&1 | tail -1 | cut -c1-200
