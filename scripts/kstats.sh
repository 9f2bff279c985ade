#!/bin/bash
# rocprofv3 per-kernel stats of a short bench run -> stdout (top rows)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
W=/tmp/kstats_$$; mkdir -p $W
rocprofv3 --kernel-trace --stats --output-format csv -d $W -o ks -- python3 bench.py "$@" --steps 20 --warmup 3 --cpu-sample 0 > $W/log.txt 2>&1
f=$(find $W -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-200
