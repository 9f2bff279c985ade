#!/bin/bash
# What do the record stores cost, and where?  scripts/placement_pmc.py (two copies of the counts x two
# engines, 10 launches each) under rocprofv3 --kernel-trace --pmc for several builds of the library:
# write / read latency at the L1-L2 boundary, tag-conflict and translation stalls, L2 write-backs, VMEM issue.
# usage: bash scripts/variant_pmc.sh <tag> <n_orfs> lib1.so lib2.so ...     ("product" = the in-tree build)
TAG=${1:-v}; N=${2:-11000000}; shift 2
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/variant_pmc_$TAG
W=/tmp/variant_pmc_$TAG
rm -rf $W; mkdir -p $W $OUT
cd /tmp
SETS=(
 "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
 "TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"
 "TCC_WRITE_sum TCC_WRITEBACK_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
 "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum"
 "SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL"
)
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = product ]; then unset RIBOPHASE_LIB; else export RIBOPHASE_LIB=$R/$lib; fi
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $W/${name}_$i -o p -- python3 $R/scripts/placement_pmc.py $N > $OUT/${name}_$i.json 2> $OUT/${name}_$i.err
  done
done
python3 - "$W" "$OUT" <<'PY'
import csv, glob, json, os, sys, collections, statistics
W, OUT = sys.argv[1], sys.argv[2]
with open(f"{OUT}/summary.txt", "w") as fh:
    def say(*a):
        print(*a); print(*a, file=fh)
    for d in sorted(os.listdir(W)):
        try:
            doc = json.loads([l for l in open(f"{OUT}/{d}.json") if l.startswith("{")][-1])
        except Exception as e:
            say(d, "no output", e); continue
        labels = [c["label"] for c in doc["combos"]]
        L, skip = doc["launches"], doc["skip"]
        dur = {}
        for f in glob.glob(f"{W}/{d}/**/*kernel_trace.csv", recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "k_tile_score" in r["Kernel_Name"]]
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            for j, r in enumerate(rows[skip:]):
                dur.setdefault(labels[min(j // L, len(labels) - 1)], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        acc = collections.defaultdict(list)
        for f in glob.glob(f"{W}/{d}/**/*counter_collection.csv", recursive=True):
            by = collections.defaultdict(dict)
            for r in csv.DictReader(open(f)):
                if "k_tile_score" in r["Kernel_Name"]:
                    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            for j, k in enumerate(sorted(by)[skip:]):
                for c, v in by[k].items():
                    acc[(labels[min(j // L, len(labels) - 1)], c)].append(v)
        say(f"== {d}: k_tile_score, rocprofv3 median us: " + ", ".join(f"{k} {statistics.median(v) / 1e3:.1f}" for k, v in dur.items()))
        for c in sorted({c for _, c in acc}):
            say(f"   {c:<44}" + " ".join(f"{lab}={statistics.mean(acc[(lab, c)]):.4g}" for lab in labels if (lab, c) in acc))
PY
