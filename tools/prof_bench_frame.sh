#!/bin/bash
# kernel stats of the whole default bench (fusion + gn + frame legs)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pf
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 --gn-solves 1 > $OUT/out.json 2> $OUT/err.txt
python3 - <<PY
import csv,glob
f=sorted(glob.glob('$OUT/trace/*/*_kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:22]:
    print("%-62s calls %5s avg %9.1f us tot %9.1f ms %5s%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
