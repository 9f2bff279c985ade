export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; rm -rf /tmp/kc
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -o s -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-verify > /tmp/kc_bench.json 2>/dev/null
python3 - <<'PY'
import csv, glob, json
for f in glob.glob("/tmp/kc/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rp::" in r["Name"]:
            print(f'{r["Name"][:60]:<62} calls {r["Calls"]:>4}  avg {float(r["AverageNs"])/1e3:9.1f} us')
d=json.loads(open("/tmp/kc_bench.json").read().strip().splitlines()[-1]); print("gather_plan_build_ms", d["fused"]["gather_plan_build_ms"], "fused kernel", d["fused"]["kernel_ms"], d["fused"]["finish_ms"])
PY
