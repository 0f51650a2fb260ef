#!/bin/bash
# memory-pipeline counters for one K1 variant / view: usage tools/pmc_k1_mem.sh <tag> <angle> <res>
# few counters per pass (TA/TCP have 2 slots), each pass under its own timeout, progress printed.
TAG=$1; ANG=$2; RES=${3:-256}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TA_TA_BUSY TA_TOTAL_WAVEFRONTS" "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ" "TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ" \
           "TCC_REQ TCC_HIT TCC_MISS" "TCC_EA0_RDREQ TCC_EA0_WRREQ" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -- python3 $ROOT/tools/kbench.py --res $RES --reps 3 --angles $ANG > $OUT/s$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob('$OUT/s*/*/*_counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'integrate_depth' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print("  ".join("%s=%.4g" % (k, v[-1]) for k,v in acc.items()))
PY
