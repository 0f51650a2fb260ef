#!/bin/bash
# per-kernel stats of the config-5-size frame leg (512^3, 8 views, 2 048 nodes) on one GPU; usage: tools/prof_frame512.sh <tag>
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pf512_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --res 512 --gn-nodes 2048 --no-cpu-baseline --no-k1-512 --no-ceiling --steps 8 --warmup 2 > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/trace/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:28]:
    print("%-64s calls %5s avg %10.1f us  tot %9.1f ms %5s%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
