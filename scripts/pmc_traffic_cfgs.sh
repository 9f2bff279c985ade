#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of rp::k_tile_score<false> for the other bench configs (what bench.py replays as roofline.traffic):
# separate --pmc passes, mean per production dispatch -> gpurun_out/traffic_cfgs.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/traffic_cfgs.txt
: > $OUT
cd /tmp
for spec in "cfg2:1000000" "cfg5:20000000" "gencode:11000000" "gencode_short:11000000"; do
  cfg=${spec%%:*}; n=${spec#*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pt; timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pt -o pt -- python3 $R/bench.py --cfg $cfg --orfs $n --steps 3 --warmup 1 --cpu-sample 0 --no-fused --no-pipelined --no-verify > /tmp/pt.log 2>&1
    python3 - "$cfg" "$n" "$c" >> $OUT <<'PY'
import csv, glob, sys
cfg, n, c = sys.argv[1:4]
tot = cnt = 0
for f in glob.glob("/tmp/pt/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rp::k_tile_score<false" in row.get("Kernel_Name", "") and row.get("Counter_Name") == c:
            tot += float(row["Counter_Value"]); cnt += 1
print(cfg, n, c, tot / cnt if cnt else None, cnt)
PY
  done
done
cat $OUT
