#!/bin/bash
# Full-size cfg3 / cfg5 bench lines (run on the GPU box) -> gpurun_out/bench_<cfg>.json
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python bench.py --cfg cfg3 --orfs 11000000 --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/bench_cfg3.json 2>gpurun_out/bench_cfg3.err
timeout 900 python bench.py --cfg cfg5 --orfs 20000000 --steps 5 --warmup 1 --cpu-sample 0 > gpurun_out/bench_cfg5.json 2>gpurun_out/bench_cfg5.err
python - <<'PY'
import json
for c in ("cfg3","cfg5"):
    try:
        d=json.load(open(f"gpurun_out/bench_{c}.json")); r=d["roofline"]
        print(c, "step_ms=%.4f kernel_ms=%.4f frac=%.3f value=%.3e"%(d["ms_per_step"], r["kernel_ms"], r["frac"], d["value"]), r["aux_kernels_ms"])
    except Exception as e:
        print(c, "failed", e, open(f"gpurun_out/bench_{c}.err").read()[-500:])
PY
