#!/bin/bash
# memory-pipeline counters of the GN build / gather kernels (BASELINE config 3's system, tools/gather_trace.py):
# usage tools/pmc_gn.sh <tag>.  Few counters per pass, each pass under its own timeout, progress printed.
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcgn_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ TCC_HIT TCC_MISS" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ" \
           "TA_TA_BUSY TA_TOTAL_WAVEFRONTS" "TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -- python3 $ROOT/tools/gather_trace.py > $OUT/s$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob('$OUT/s*/*/*_counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        for tag in ('gn_gather', 'gn_build_data', 'pcg_cg1'):
            if tag in k:
                acc[tag][r['Counter_Name']].append(float(r['Counter_Value']))
    for tag, d in acc.items():
        print(tag, "  ".join("%s=%.4g" % (k, v[-1]) for k,v in d.items()))
PY
