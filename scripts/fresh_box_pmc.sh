#!/bin/bash
# Fresh boxes start slow (profiles/r06_alloc_order.txt): is it address translation?  The same child process under
# `rocprofv3 --pmc` (translation counters of the vector L1 / UTCL2, in a run of their own) FIRST on the box and again after
# four plain processes; per-launch means for rp::k_tile_score next to the kernel time each child prints.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -o "TCP_UTCL1[A-Z_0-9a-z]*\|UTCL2[A-Z_0-9a-z]*\|TCP_TCC_READ_REQ_LATENCY[A-Za-z_]*\|TCP_PENDING_STALL_CYCLES[A-Za-z_]*" | sort -u | head -30 > /tmp/counters.txt
echo "counters on offer:"; cat /tmp/counters.txt | tr '\n' ' '; echo
CNT="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"
for phase in first later; do
  rm -rf /tmp/fb_$phase; mkdir -p /tmp/fb_$phase
  rocprofv3 --pmc $CNT --output-format csv -d /tmp/fb_$phase -o p -- python3 $R/scripts/ab_alloc_order.py child counts_first > /tmp/fb_$phase/out.txt 2> /tmp/fb_$phase/err.txt
  echo "$phase: child printed (kernel finish step ms): $(tail -1 /tmp/fb_$phase/out.txt)"
  python3 - $phase <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"/tmp/fb_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_tile_score" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print(f"   {k}: {s / n:.4g} per launch ({n} launches)")
PY
  if [ $phase = first ]; then
    for i in 1 2 3 4; do echo "   plain process $i: $(python3 $R/scripts/ab_alloc_order.py child counts_first 2>/dev/null | tail -1)"; done
  fi
done
