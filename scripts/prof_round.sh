#!/bin/bash
# One box, one call: the default bench line, the same command under rocprofv3 (kernel-trace
# stats) and six separate --pmc passes (SQ wave/VALU counters; LDS/SALU/VMEM + GRBM_GUI_ACTIVE;
# FETCH_SIZE; WRITE_SIZE; the last two again for the nested fused section alone), summarised per kernel.  usage: bash scripts/prof_round.sh <tag> [bench args]
TAG=${1:-r02}; shift
ARGS="$@"
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/prof_$TAG
W=/tmp/prof_$TAG
rm -rf $W; mkdir -p $W $OUT
timeout 600 python3 bench.py $ARGS --detail $OUT/bench_default_detail.json > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -o stats -- python3 $R/bench.py $ARGS --steps 20 --warmup 3 --cpu-sample 0 > $OUT/bench_under_rocprofv3.json 2> $OUT/stats.err
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $W/pmc1 -o pmc1 -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 --cpu-sample 0 --no-pipelined --no-slice-projection --no-fused-nested > $OUT/pmc1_bench.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $W/pmc2 -o pmc2 -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 --cpu-sample 0 --no-pipelined --no-slice-projection --no-fused-nested > $OUT/pmc2_bench.log 2>&1
# (the two fused layouts launch the same kernel name: counted in separate passes -- pmc3/4 the run's own length law, pmc5/6 the nested index)
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/pmc3 -o pmc3 -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 --cpu-sample 0 --no-pipelined --no-slice-projection --no-fused-nested > $OUT/pmc3_bench.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/pmc4 -o pmc4 -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 --cpu-sample 0 --no-pipelined --no-slice-projection --no-fused-nested > $OUT/pmc4_bench.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/pmc5 -o pmc5 -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 --cpu-sample 0 --no-pipelined --no-slice-projection --no-fused-exons > $OUT/pmc5_bench.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/pmc6 -o pmc6 -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 --cpu-sample 0 --no-pipelined --no-slice-projection --no-fused-exons > $OUT/pmc6_bench.log 2>&1
cd $R
cp $(find $W/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
python3 - "$W" "$OUT" <<'PY'
import csv, glob, sys, statistics, collections
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
        fh.write(f"\"{name}\",{len(d)},{statistics.mean(d):.0f},{statistics.median(d):.0f},{min(d)},{max(d)},{statistics.mean(d[-40:]):.0f}\n")
print(open(f"{OUT}/kernel_durations.csv").read())
for tag in ("pmc1", "pmc2", "pmc3", "pmc4", "pmc5", "pmc6"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{W}/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "?")
            if "rp::" not in name:
                continue
            k = (name[:60], row.get("Counter_Name", "?"))
            acc[k][0] += float(row.get("Counter_Value", 0)); acc[k][1] += 1
    with open(f"{OUT}/{tag}_summary.csv", "w") as fh:
        fh.write("kernel,counter,mean_per_dispatch,dispatches\n")
        for (k, c), (s, n) in sorted(acc.items()):
            fh.write(f"\"{k}\",{c},{s/n:.1f},{n}\n")
    print(open(f"{OUT}/{tag}_summary.csv").read())
PY
grep -E "rp::" $OUT/kernel_stats.csv | cut -c1-160
