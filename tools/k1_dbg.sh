#!/bin/bash
# Where does the single-view brick sweep spend its time?  512^3, one view: full kernel, no T/w loads, no stores, neither;
# then SQ / TCP counters of the full kernel.   usage: tools/k1_dbg.sh <outdir> [angle]
OUT=${1:-gpurun_out/k1_dbg}; ANG=${2:-0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
for d in 0 1 2 3; do
  DFH_K1_DBG=$d python tools/kbench.py --res 512 --reps 20 --angles $ANG 2>/dev/null | grep view > $OUT/dbg$d.txt
done
DFH_K1_NO_BRICKS=1 python tools/kbench.py --res 512 --reps 20 --angles $ANG 2>/dev/null | grep view > $OUT/rows.txt
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES TA_TA_BUSY TCC_HIT TCC_MISS TCC_REQ"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$name -- python3 $ROOT/tools/kbench.py --res 512 --reps 3 --angles $ANG > $ROOT/$OUT/pmc_$name.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob('$OUT/pmc_*/*/*_counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        kn=r['Kernel_Name']
        if 'integrate_depth' in kn or 'classify' in kn or 'pyramid' in kn:
            acc[kn.split('<')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for kn,d in acc.items():
        for k,v in d.items():
            print("%-42s %-28s n=%d last=%.5g" % (kn, k, len(v), v[-1]))
PY
cat $OUT/dbg*.txt $OUT/rows.txt
