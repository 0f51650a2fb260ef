#!/bin/bash
# true kernel durations of tools/kbench.py via rocprofv3 kernel trace; usage: tools/prof_kbench.sh <tag> [kbench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kb_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/tools/kbench.py "$@" > $OUT/kbench.txt 2> $OUT/err.txt
python3 - <<PY
import csv,glob,collections
f=glob.glob('$OUT/trace/*/*_kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'integrate' in r['Kernel_Name'] or 'pyramid' in r['Kernel_Name']]
# group consecutive launches of the big kernel into runs of equal duration class (per view)
durs=collections.defaultdict(list)
for r in rows:
    durs[r['Kernel_Name'][:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in durs.items():
    import statistics
    # views are timed in blocks of (5 warm + reps + 1); print per-chunk medians
    n=len(v); chunk=max(1,n//5)
    print(k, n, "medians per fifth:", [round(statistics.median(v[i:i+chunk]),1) for i in range(0,n,chunk)][:6])
PY
cat $OUT/kbench.txt | grep view
