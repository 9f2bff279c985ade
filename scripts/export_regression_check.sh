#!/bin/bash
# Is the first sample of the 11 M-line export slower in round 5 than in round 4 because of the CODE or because of the BOX?
# (round-5 verdict, Next #9.)  The r04 and r05 trees (git archive of the round-end commits, built in variants/tree_rNN by
# the caller) and the working tree run scripts/bench_export_big.py one after the other ON ONE BOX, twice round the ring.
# Output: gpurun_out/export_regression.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/export_regression.txt
: > $OUT
for pass in 1 2 3; do
  if [ $pass = 1 ]; then ORDER="variants/tree_r04 variants/tree_r05 ."; elif [ $pass = 2 ]; then ORDER=". variants/tree_r05 variants/tree_r04"; else ORDER="variants/tree_r05 variants/tree_r04 ."; fi
  for tree in $ORDER; do
    J=/tmp/exp_$$.json
    rm -f $J
    (cd $R/$tree && timeout 900 python3 scripts/bench_export_big.py 11000000 100000000 $J > /tmp/exp_$$.log 2>&1)
    python3 - "$tree" "$pass" "$J" >> $OUT <<'PY'
import json, sys
tree, p, path = sys.argv[1:4]
try:
    d = json.load(open(path))
except Exception as e:
    print(f"{tree} pass {p}: failed ({e}); log tail: " + " | ".join(open(path.replace(".json", ".log")).read().splitlines()[-6:])); sys.exit(0)
f, s, ra = d["default"], d["default_second_sample"], d["report_all"]
st = f["stages_s"]
print(f"{tree:18s} pass {p}: first sample {f['wall_s']:.3f} s (index_parse {st['index_parse']:.3f}, plan {st.get('interval_table_gather_plan', 0):.3f}, "
      f"coverage {st.get('coverage_build', 0):.3f}, score+d2h {st.get('fused_score_results_d2h', 0):.3f}, tsv {st.get('profiles_d2h_tsv_render_write', 0):.3f}); "
      f"second sample {s['wall_s']:.3f} s; report_all {ra['wall_s']:.3f} s")
PY
  done
done
cat $OUT
