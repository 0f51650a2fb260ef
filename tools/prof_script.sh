#!/bin/bash
# per-kernel stats of any tools/*.py script via rocprofv3; usage: tools/prof_script.sh <tag> <script.py> [args]
TAG=$1; SCRIPT=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ps_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/$SCRIPT "$@" > $OUT/out.txt 2> $OUT/err.txt
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/trace/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-60s calls %5s avg %10.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
