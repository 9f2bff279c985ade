run() { echo "== $*"; timeout 600 python bench.py "$@" 2>/tmp/err.txt | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t else {}
print({k:d.get(k) for k in ('value','n_gpus','steps','warmup','ms_per_step','scaling')}, d.get('verify',{}).get('ok'), d.get('config',{}).get('control_backend'), 'err' if 'error' in d else '')" || tail -3 /tmp/err.txt; }
run --orfs 1000 --cpu-sample 0
run --orfs 200000 --steps 1 --warmup 0 --cpu-sample 0 --no-fused
run --gpus 3 --orfs 300000 --steps 2 --warmup 1
run --gpus 2 --orfs 300000 --steps 2 --warmup 1 --scaling weak
run --cfg cfg5 --orfs 400000 --steps 2 --warmup 1 --cpu-sample 0 --no-fused-nested
RP_BENCH_BACKEND=nccl run --gpus 2 --orfs 100000 --steps 2 --warmup 1
