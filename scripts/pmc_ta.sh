#!/bin/bash
# Texture-addresser / L1 counters of the fused and the plain scoring kernel on the same profiles (two bench layouts,
# 6 M ORFs): what the fused kernel's 4x smaller LDS-DMA requests cost at the TA / TCP.  -> gpurun_out/pmc_ta/
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_ta
W=/tmp/pmc_ta
rm -rf $W; mkdir -p $W $OUT
cd /tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
grep -o "\b\(TA\|TCP\|TD\)_[A-Za-z0-9_]*" $OUT/avail.txt | sort -u > $OUT/avail_ta_tcp.txt
export RP_EXPERIMENT_PLAIN=1
pass() {  # pass <tag> counters...
  local tag=$1; shift
  local ok=""
  for c in "$@"; do if grep -qx "$c" $OUT/avail_ta_tcp.txt; then ok="$ok $c"; else echo "not available: $c" >> $OUT/missing.txt; fi; done
  [ -z "$ok" ] && return
  timeout 150 rocprofv3 --pmc $ok --output-format csv -d $W/$tag -o $tag -- python3 $R/scripts/fused_variant_experiment.py > $OUT/${tag}_run.log 2>&1
}
# (at most two counters of one block per pass: four TA counters at once is "Request exceeds the capabilities of the hardware")
pass ta1 TA_TA_BUSY_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum
pass tcp2 TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass td1 TD_TD_BUSY_sum TD_TC_STALL_sum
pass grbm GRBM_GUI_ACTIVE
cd $R
python3 - "$W" "$OUT" <<'PY'
import csv, glob, sys, collections
W, OUT = sys.argv[1], sys.argv[2]
with open(f"{OUT}/summary.csv", "w") as fh:
    fh.write("kernel,counter,mean_per_dispatch,dispatches\n")
    for tag in ("ta1", "ta2", "tcp1", "tcp2", "td1", "grbm"):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob(f"{W}/{tag}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                name = row.get("Kernel_Name", "?")
                if "k_tile_score" not in name:
                    continue
                # (dispatches of the 6 M-ORF launches only: grid size tells the two layouts apart)
                k = (name[:60] + " grid " + row.get("Grid_Size", "?"), row.get("Counter_Name", "?"))
                acc[k][0] += float(row.get("Counter_Value", 0)); acc[k][1] += 1
        for (k, c), (s, n) in sorted(acc.items()):
            fh.write(f"\"{k}\",{c},{s/n:.1f},{n}\n")
print(open(f"{OUT}/summary.csv").read())
PY
