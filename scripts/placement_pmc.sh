#!/bin/bash
# scripts/placement_pmc.py under rocprofv3: a kernel-trace pass and PMC passes (address translation,
# L1 / L2 stalls, memory-side credit stalls), summarised per (copy of the counts, engine) combination.
# usage: bash scripts/placement_pmc.sh <tag> [n_orfs]
TAG=${1:-pl}; N=${2:-11000000}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/placement_$TAG
W=/tmp/placement_$TAG
rm -rf $W; mkdir -p $W $OUT
cd /tmp
timeout -s KILL 150 python3 $R/scripts/placement_pmc.py $N > $OUT/plain.json 2> $OUT/plain.err
timeout -s KILL 150 rocprofv3 --kernel-trace --output-format csv -d $W/trace -o trace -- python3 $R/scripts/placement_pmc.py $N > $OUT/trace.json 2> $OUT/trace.err
i=0
# (<= 4 counters of one block per pass: a request the hardware cannot serve aborts rocprofv3, which then hangs)
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $W/pmc$i -o pmc$i -- python3 $R/scripts/placement_pmc.py $N > $OUT/pmc$i.json 2> $OUT/pmc$i.err
done
python3 - "$W" "$OUT" <<'PY'
import csv, glob, json, sys, collections, statistics
W, OUT = sys.argv[1], sys.argv[2]
with open(f"{OUT}/summary.txt", "w") as fh:
    def say(*a):
        print(*a); print(*a, file=fh)
    for tag in ["trace"] + [f"pmc{i}" for i in range(1, 6)]:
        try:
            doc = json.loads([l for l in open(f"{OUT}/{tag}.json") if l.startswith("{")][-1])
        except Exception as e:
            say(tag, "no output", e); continue
        labels = [c["label"] for c in doc["combos"]]
        L, skip = doc["launches"], doc["skip"]
        dur = {}
        for f in glob.glob(f"{W}/{tag}/**/*kernel_trace.csv", recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "k_tile_score" in r["Kernel_Name"]]
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            for j, r in enumerate(rows[skip:]):
                dur.setdefault(labels[min(j // L, len(labels) - 1)], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        acc = collections.defaultdict(list)
        for f in glob.glob(f"{W}/{tag}/**/*counter_collection.csv", recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "k_tile_score" in r["Kernel_Name"]]
            by = collections.defaultdict(dict)
            for r in rows:
                by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            for j, d in enumerate(sorted(by)[skip:]):
                for c, v in by[d].items():
                    acc[(labels[min(j // L, len(labels) - 1)], c)].append(v)
        say(f"== {tag}: own HIP events (median ms): " + ", ".join(f"{c['label']} {c['main_ms_median']}" for c in doc["combos"]))
        say("   rocprofv3 durations (median us): " + ", ".join(f"{k} {statistics.median(v) / 1e3:.1f}" for k, v in dur.items()))
        counters = sorted({c for _, c in acc})
        for c in counters:
            say(f"   {c:<48}" + " ".join(f"{lab}={statistics.mean(acc[(lab, c)]):.4g}" for lab in labels if (lab, c) in acc))
PY
