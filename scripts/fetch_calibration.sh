#!/bin/bash
# One `rocprofv3 --pmc FETCH_SIZE` pass (counters in a run of their own) over scripts/fetch_calibration.py, then the parse.
# Output: gpurun_out/fetch_calibration.txt   (copy into profiles/r06_fetch_calibration.txt)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=/tmp/fetch_cal
rm -rf $W; mkdir -p $W $R/gpurun_out
cd /tmp
python3 $R/scripts/fetch_calibration.py > $W/timing.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/pmc -o fc -- python3 $R/scripts/fetch_calibration.py > $W/run.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/pmcw -o fw -- python3 $R/scripts/fetch_calibration.py > $W/runw.log 2>&1
cd $R
{ echo "# scripts/fetch_calibration.sh on $(date -u +%F): timings (no profiler)"; cat $W/timing.txt;
  echo; echo "# rocprofv3 --pmc FETCH_SIZE pass of the same program"; python3 scripts/fetch_calibration.py --parse $W/pmc;
  echo; echo "# rocprofv3 --pmc WRITE_SIZE pass of the same program"; python3 scripts/fetch_calibration.py --parse $W/pmcw; } > gpurun_out/fetch_calibration.txt
cat gpurun_out/fetch_calibration.txt
