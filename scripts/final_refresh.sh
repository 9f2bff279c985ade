#!/bin/bash
# One box, one call: default bench line (with CPU baseline), rocprofv3 stats + PMC passes of the
# same command, wave / pipe / cfg3 / cfg5 lines.  Everything lands in gpurun_out/.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
bash scripts/prof_tile.sh final --algo tile > gpurun_out/prof_final.log 2>&1
grep '^{' gpurun_out/prof_final/stats_bench.log > gpurun_out/bench_under_rocprofv3.json
timeout 300 python bench.py --algo wave --cpu-sample 0 > gpurun_out/bench_wave.json 2> gpurun_out/bench_wave.err
timeout 300 python bench.py --algo pipe --cpu-sample 0 > gpurun_out/bench_pipe.json 2> gpurun_out/bench_pipe.err
timeout 900 python bench.py --cfg cfg3 --orfs 11000000 --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/bench_cfg3.json 2>gpurun_out/bench_cfg3.err
timeout 900 python bench.py --cfg cfg5 --orfs 20000000 --steps 5 --warmup 1 --cpu-sample 0 > gpurun_out/bench_cfg5.json 2>gpurun_out/bench_cfg5.err
python - <<'PY'
import json
for c in ("default","under_rocprofv3","wave","pipe","cfg3","cfg5"):
    try:
        d=json.load(open(f"gpurun_out/bench_{c}.json")); r=d["roofline"]
        print(c, "step_ms=%.4f kernel_ms=%.4f frac=%.3f value=%.3e"%(d["ms_per_step"], r["kernel_ms"], r["frac"], d["value"]), r.get("aux_kernels_ms"), d.get("quality"))
    except Exception as e:
        print(c, "failed", e)
PY
cat gpurun_out/prof_final/kernel_durations.csv
grep -E "k_tile_score" gpurun_out/prof_final/pmc3_summary.csv gpurun_out/prof_final/pmc4_summary.csv gpurun_out/prof_final/pmc1_summary.csv gpurun_out/prof_final/pmc2_summary.csv
