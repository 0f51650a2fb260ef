#!/bin/bash
# SQ counters for tools/kbench.py (one view); usage: tools/pmc_kbench.sh <tag> <angle> <res>
TAG=$1; ANG=$2; RES=${3:-256}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/tools/kbench.py --res $RES --reps 3 --angles $ANG > $OUT/$name.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob('$OUT/*/*/*_counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'integrate_depth' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print("%-26s n=%d last=%.4g" % (k, len(v), v[-1]))
PY
