#!/bin/bash
# Serial-stream kernel-trace profiles of bench.py under two (or more) environment settings, for per-kernel A/B tables.
# usage: bash tools/prof_ab.sh <tag> "<VAR=val ...>" "<VAR=val ...>" ...   -> gpurun_out/<tag>/<i>_kernel_stats.csv
set -o pipefail
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  ( export $v PU_NO_SIDE_STREAM=1; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$i -o p$i -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > $O/p$i.log 2>&1 ) || { echo "profile $i failed"; tail -5 $O/p$i.log; exit 1; }
  f=$(find $O/p$i -name "*kernel_stats.csv" | head -1); cp $f $O/${i}_kernel_stats.csv
  echo "== $v"; python3 $R/tools/prof_summary.py $O/${i}_kernel_stats.csv 11 45
  i=$((i+1))
done
find $O -name "*.csv" -size +3M -delete
