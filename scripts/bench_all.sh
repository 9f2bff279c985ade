#!/bin/bash
# The bench lines committed under profiles/: default (with CPU baseline), wave, pipe.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout 300 python bench.py --algo wave --cpu-sample 0 > gpurun_out/bench_wave.json 2> gpurun_out/bench_wave.err
timeout 300 python bench.py --algo pipe --cpu-sample 0 > gpurun_out/bench_pipe.json 2> gpurun_out/bench_pipe.err
python - <<'PY'
import json
for c in ("default","wave","pipe"):
    try:
        d=json.load(open(f"gpurun_out/bench_{c}.json")); r=d["roofline"]
        print(c, "step_ms=%.4f kernel_ms=%.4f frac=%.3f value=%.3e"%(d["ms_per_step"], r["kernel_ms"], r["frac"], d["value"]), r.get("aux_kernels_ms"), d.get("cpu_baseline"))
    except Exception as e:
        print(c, "failed", e, open(f"gpurun_out/bench_{c}.err").read()[-500:])
PY
