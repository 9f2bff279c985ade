#!/bin/bash
# Per-kernel durations (rocprofv3 --kernel-trace --stats) of several builds of the library on ONE box.
# usage: bash scripts/kstat_variants.sh <variant> [<variant> ...]   (variants/<name>.so)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  cd /tmp; rm -rf /tmp/ks_$v
  RIBOPHASE_LIB=$R/variants/$v.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o s -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-fused --no-verify ${KSTAT_ARGS} > /dev/null 2>&1
  echo "== $v"
  python3 - /tmp/ks_$v <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rp::" in r["Name"]:
            print(f'{r["Name"][:60]:<62} calls {r["Calls"]:>4}  avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}  max {float(r["MaxNs"])/1e3:9.1f}')
PY
done
