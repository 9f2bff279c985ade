#!/bin/bash
# Everything the round's BASELINE.md / DESIGN.md numbers quote, from ONE box and one gpurun call:
# prof_round.sh (default bench line, rocprofv3 stats, four PMC passes) + the other configs' bench
# lines + shapes + the 1 M-ORF export bench + the front end.  usage: bash scripts/round_evidence.sh r03
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/prof_$TAG
bash scripts/prof_round.sh $TAG > $R/gpurun_out/prof_${TAG}.log 2>&1
for spec in "cfg2:1000000" "cfg5:20000000" "gencode:11000000" "gencode_short:11000000"; do
  cfg=${spec%%:*}; n=${spec#*:}
  timeout 900 python3 bench.py --cfg $cfg --orfs $n --cpu-sample 0 --no-fused-nested > $OUT/bench_${cfg}.json 2> $OUT/bench_${cfg}.err
done
timeout 900 python3 bench.py --cfg cfg5 --orfs 50000000 --steps 10 --warmup 2 --cpu-sample 0 --no-fused > $OUT/bench_cfg5_50M.json 2> $OUT/bench_cfg5_50M.err
timeout 600 python3 scripts/bench_shapes.py > $OUT/shapes.txt 2>&1
timeout 600 python3 scripts/bench_export.py 1000000 10 > $OUT/export_e2e_1M.json 2> $OUT/export_e2e_1M.err
timeout 600 python3 scripts/bench_frontend.py 2000000 6000 > $OUT/frontend.json 2> $OUT/frontend.err
ls -la $OUT
