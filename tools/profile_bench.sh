#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then HBM traffic counters in their
# own passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: tools/profile_bench.sh <tag> [bench args...]
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --no-cpu-baseline --no-gn --no-frame --steps 8 --warmup 4 > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --no-cpu-baseline --no-gn --no-frame --steps 8 --warmup 4 > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
