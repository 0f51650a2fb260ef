#!/bin/bash
# per-kernel times of K3 with the constant-live skip on the probe scene; usage: tools/prof_k3_skip.sh <tag> [--res 512 --nodes 2048 ...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ps_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/k3_skip_probe.py "$@" > $OUT/out.txt 2> $OUT/err.txt
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/trace/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f))):
    if 'dqb' in r["Name"]:
        print("%-72s calls %5s avg %10.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"])/1e3))
PY
grep -i "skip\|ms\|us" $OUT/out.txt | tail -12
