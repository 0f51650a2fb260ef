#!/bin/bash
# SQ / TA / TCP / traffic counters of the kernels whose name contains <substr>, for any python script:
#   tools/pmc_any.sh <outdir> <substr> <script.py> [args...]          (each counter set is its own rocprofv3 pass)
OUT=$1; SUB=$2; SCRIPT=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES TA_TA_BUSY TCC_HIT TCC_MISS TCC_REQ" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS TA_BUSY_AVR TCP_TCC_READ_REQ TCP_GATE_EN1"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$name -- python3 $ROOT/$SCRIPT "$@" > $ROOT/$OUT/pmc_$name.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob('$OUT/pmc_*/*/*_counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        kn=r['Kernel_Name']
        if '$SUB' in kn:
            acc[kn.split('(')[0][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for kn,d in acc.items():
        for k,v in d.items():
            print("%-72s %-28s n=%d last=%.5g" % (kn, k, len(v), v[-1]))
PY
