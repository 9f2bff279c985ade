#!/bin/bash
# Does a verify block of bench.py fail again under rocprofv3 (fused_nested did once: box d of round 6, the exact command of
# scripts/prof_round.sh)?  N attempts of that command, each its own process; any error line is kept in gpurun_out/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-6}
mkdir -p $R/gpurun_out
cd /tmp
for i in $(seq 1 $N); do
  rm -rf /tmp/rp_$i; mkdir -p /tmp/rp_$i
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$i -o s -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-sample 0 > /tmp/rp_$i/out.json 2> /tmp/rp_$i/err.txt
  python3 - $i $R <<'PY'
import json, shutil, sys
i, R = sys.argv[1], sys.argv[2]
d = json.load(open(f"/tmp/rp_{i}/out.json"))
if "error" in d:
    shutil.copy(f"/tmp/rp_{i}/out.json", f"{R}/gpurun_out/repro_error_{i}.json")
    for k in ("verify", "fused", "fused_nested"):
        v = d.get(k)
        v = v.get("verify", v) if isinstance(v, dict) and k != "verify" else v
        if isinstance(v, dict) and not v.get("ok", True):
            print(i, "ERROR in", k, {s: (x.get("error") or "ok")[:300] for s, x in v.get("slices", {}).items()})
else:
    print(i, "ok", d["verify"]["ok"], d["fused"]["verify"]["ok"], d["fused_nested"]["verify"]["ok"], f"{d['value']:.3e}")
PY
done
