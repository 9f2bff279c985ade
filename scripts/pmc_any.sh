#!/bin/bash
# rocprofv3 kernel-trace stats + separate --pmc passes (SQ; LDS/SALU; FETCH_SIZE; WRITE_SIZE; TCC hit/miss) around ANY
# python script of this repo, summarised per rp:: kernel into gpurun_out/pmc_<tag>/.
# usage: bash scripts/pmc_any.sh <tag> <script.py> [args...]      (the program after `--` is python3 itself)
TAG=$1; shift
SCRIPT=$1; shift
ARGS="$@"
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
W=/tmp/pmc_$TAG
rm -rf $W; mkdir -p $W $OUT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -o stats -- python3 $R/$SCRIPT $ARGS > $OUT/stats_run.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $W/pmc1 -o pmc1 -- python3 $R/$SCRIPT $ARGS > $OUT/pmc1_run.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $W/pmc2 -o pmc2 -- python3 $R/$SCRIPT $ARGS > $OUT/pmc2_run.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/pmc3 -o pmc3 -- python3 $R/$SCRIPT $ARGS > $OUT/pmc3_run.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/pmc4 -o pmc4 -- python3 $R/$SCRIPT $ARGS > $OUT/pmc4_run.log 2>&1
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $W/pmc5 -o pmc5 -- python3 $R/$SCRIPT $ARGS > $OUT/pmc5_run.log 2>&1
cd $R
python3 - "$W" "$OUT" <<'PY'
import csv, glob, sys, statistics, collections
W, OUT = sys.argv[1], sys.argv[2]
rows = {}
for f in glob.glob(f"{W}/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "?")
        if "rp::" not in name:
            continue
        rows.setdefault(name[:80], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(f"{OUT}/kernel_durations.csv", "w") as fh:
    fh.write("kernel,calls,mean_ns,median_ns,min_ns,max_ns\n")
    for name, d in sorted(rows.items()):
        fh.write(f"\"{name}\",{len(d)},{statistics.mean(d):.0f},{statistics.median(d):.0f},{min(d)},{max(d)}\n")
print(open(f"{OUT}/kernel_durations.csv").read())
for tag in ("pmc1", "pmc2", "pmc3", "pmc4", "pmc5"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{W}/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "?")
            if "rp::" not in name:
                continue
            k = (name[:70], row.get("Counter_Name", "?"))
            acc[k][0] += float(row.get("Counter_Value", 0)); acc[k][1] += 1
    with open(f"{OUT}/{tag}_summary.csv", "w") as fh:
        fh.write("kernel,counter,mean_per_dispatch,dispatches\n")
        for (k, c), (s, n) in sorted(acc.items()):
            fh.write(f"\"{k}\",{c},{s/n:.1f},{n}\n")
    print(open(f"{OUT}/{tag}_summary.csv").read())
PY
