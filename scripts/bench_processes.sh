#!/bin/bash
# P fresh processes of bench.py on one box, one summary line each: the headline, what the first allocation gave, what the two
# placement searches (workspace, source) saw.  usage: bash scripts/bench_processes.sh [P] [out dir] [bench args]
P=${1:-3}; OUT=${2:-gpurun_out/bench_processes}; shift; shift
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p $OUT
for p in $(seq 1 $P); do
  bash scripts/box_kind.sh 2>/dev/null | tail -1
  timeout 600 python3 bench.py --cpu-sample 0 --no-fused-nested "$@" > $OUT/bench_$p.json 2> $OUT/bench_$p.err
  python3 - $OUT/bench_$p.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
pr = d["per_rank"][0]; wp = d["config"]["workspace_placement"]; sp = wp.get("source_placement", {})
print("value %.4e step %.4f kernel %.4f first %.4e pipelined %.4e" % (d["value"], d["ms_per_step"], pr["kernel_ms"], d["value_first_allocation"], d["value_pipelined"]))
print("  workspace search", wp["step_ms"], "chosen", wp["chosen"], "| source search", sp.get("step_ms"), sp.get("kernel_gbps"), "chosen", sp.get("chosen"), "| settled", wp.get("settled"), wp.get("settled_after_s"))
f = d.get("fused")
if f:
    fw = f["workspace_placement"]
    print("  fused kernel %.4f finish %.4f step %.4f" % (f["kernel_ms"], f["finish_ms"], f["step_device_ms"]), fw["step_ms"], (fw.get("source_placement") or {}).get("step_ms"), (fw.get("source_placement") or {}).get("chosen"))
PY
done
