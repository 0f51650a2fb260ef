#!/bin/bash
# Round profile on the GPU box (writes gpurun_out/prof_<tag>/, condensed into profiles/ by tools/summarize_profile.py):
#   trace      rocprofv3 --kernel-trace --stats of the default bench.py command
#   pmc_fetch / pmc_write   FETCH_SIZE and WRITE_SIZE (separate passes, MI355X_MICROARCH.md) of the WHOLE bench (K1, the multi-view
#              sweep, the 512^3 leg, the GN kernels, K3, extraction, marching cubes), short step counts
#   fv_*       the same two counters for tools/kbench_fv.py (K2 rigid, K3 with and without stored neighbourhoods)
#   mv_*       ... for tools/kbench_views.py --res 512 --views 8 --orbit (config 5's sweep)
# usage: tools/profile_round.sh <tag>
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-ceiling > $OUT/bench_trace.json 2> $OUT/trace.err
echo trace done
SHORT="--no-cpu-baseline --no-ceiling --steps 8 --warmup 4 --gn-solves 1 --launch eager"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $SHORT > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $SHORT > $OUT/bench_write.json 2> $OUT/write.err
echo write done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/fv_$c -- python3 $ROOT/tools/kbench_fv.py --reps 3 > $OUT/fv_$c.txt 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/mv_$c -- python3 $ROOT/tools/kbench_views.py --res 512 --views 8 --orbit --reps 3 > $OUT/mv_$c.txt 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fv_trace -- python3 $ROOT/tools/kbench_fv.py --reps 10 > $OUT/fv_trace.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mv_trace -- python3 $ROOT/tools/kbench_views.py --res 512 --views 8 --orbit --reps 10 > $OUT/mv_trace.txt 2>&1
echo all done
find $OUT -name "*_kernel_stats.csv" -o -name "*counter_collection.csv" | head -20
