#!/bin/bash
# rocprofv3 kernel-trace stats + two PMC passes of the bench command (run on the GPU box).
# usage: bash scripts/prof_tile.sh <tag> [bench args...]
TAG=${1:-r01}; shift
ARGS=${@:---algo tile}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/prof_$TAG
W=/tmp/prof_$TAG
rm -rf $W; mkdir -p $W $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -o stats -- python3 bench.py $ARGS --steps 20 --warmup 3 --cpu-sample 0 > $OUT/stats_bench.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $W/pmc1 -o pmc1 -- python3 bench.py $ARGS --steps 3 --warmup 1 --cpu-sample 0 > $OUT/pmc1_bench.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $W/pmc2 -o pmc2 -- python3 bench.py $ARGS --steps 3 --warmup 1 --cpu-sample 0 > $OUT/pmc2_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/pmc3 -o pmc3 -- python3 bench.py $ARGS --steps 3 --warmup 1 --cpu-sample 0 > $OUT/pmc3_bench.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/pmc4 -o pmc4 -- python3 bench.py $ARGS --steps 3 --warmup 1 --cpu-sample 0 > $OUT/pmc4_bench.log 2>&1
find $W -name "*.csv" | while read f; do echo "$f $(wc -c < $f)"; done
cp $(find $W/stats -name "*kernel_stats.csv") $OUT/kernel_stats.csv 2>/dev/null
# rocprofv3's stats give mean/min/max; add the median and the mean of the timed region (the
# last 20 + 20 launches: bench.py times 20 steps, then 20 more with per-kernel events) per kernel
python3 - "$W" "$OUT" <<'PY'
import csv, glob, sys, statistics
W, OUT = sys.argv[1], sys.argv[2]
rows = {}
for f in glob.glob(f"{W}/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "?")
        if "rp::" not in name:
            continue
        rows.setdefault(name[:70], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
with open(f"{OUT}/kernel_durations.csv", "w") as fh:
    fh.write("kernel,calls,mean_ns,median_ns,min_ns,max_ns,mean_last40_ns\n")
    for name, v in sorted(rows.items()):
        v.sort()
        d = [x[1] for x in v]
        last = d[-40:]
        fh.write(f"\"{name}\",{len(d)},{statistics.mean(d):.0f},{statistics.median(d):.0f},{min(d)},{max(d)},{statistics.mean(last):.0f}\n")
print(open(f"{OUT}/kernel_durations.csv").read())
PY
# per-kernel mean of every counter (counter_collection csv is one row per dispatch x counter)
python3 - "$W" "$OUT" <<'PY'
import csv, glob, sys, collections
W, OUT = sys.argv[1], sys.argv[2]
for tag in ("pmc1", "pmc2", "pmc3", "pmc4"):
    files = glob.glob(f"{W}/{tag}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
            acc[k][0] += float(row.get("Counter_Value", 0)); acc[k][1] += 1
    with open(f"{OUT}/{tag}_summary.csv", "w") as fh:
        fh.write("kernel,counter,mean_per_dispatch,dispatches\n")
        for (k, c), (s, n) in sorted(acc.items()):
            fh.write(f"\"{k}\",{c},{s/n:.1f},{n}\n")
PY
head -12 $OUT/kernel_stats.csv
grep -E "k_tile|k_wave" $OUT/pmc1_summary.csv $OUT/pmc2_summary.csv $OUT/pmc3_summary.csv $OUT/pmc4_summary.csv
