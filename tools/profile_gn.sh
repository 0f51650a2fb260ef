#!/bin/bash
# kernel trace of the GN leg only
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 2 "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
cat $OUT/trace/*/*_kernel_stats.csv
