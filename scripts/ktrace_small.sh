export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
rm -rf /tmp/kt; mkdir -p /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python3 $R/scripts/ab_finish_tail.py child 1000000 cfg2 > /tmp/kt/log.txt 2>&1
python3 - <<'PY'
import csv, glob, statistics
rows={}
for f in glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "rp::" in n: rows.setdefault(n[:60],[]).append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(rows.items()):
    v=v[-40:]
    print(f"{k:62s} n={len(v)} median {statistics.median(v)/1e3:.1f} us  min {min(v)/1e3:.1f}  max {max(v)/1e3:.1f}")
PY
