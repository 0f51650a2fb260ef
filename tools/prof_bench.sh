#!/bin/bash
# per-kernel stats of a bench.py run; usage: tools/prof_bench.sh <tag> [pattern] [bench.py arguments ...]
TAG=$1; PAT=${2:-.}; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pb_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv,glob,re
f=glob.glob('$OUT/trace/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f))):
    if re.search(r'$PAT', r["Name"]):
        print("%-64s calls %5s avg %9.1f min %9.1f max %9.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
