#!/bin/bash
# memory-pipeline counters of the GN build / gather kernels (BASELINE config 3's system, tools/gather_trace.py):
# usage tools/pmc_gn.sh <tag> [sq].  Few counters per pass, each pass under its own timeout, progress printed.
# "sq": the issue-side counters instead (what the waves of the build kernel spend their cycles on).
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcgn_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
if [ "$2" = "sq" ]; then
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
      "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" \
      "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" \
      "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM")
else
SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_REQ TCC_HIT TCC_MISS" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ" \
           "TA_TA_BUSY TA_TOTAL_WAVEFRONTS" "TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU")
fi
for set in "${SETS[@]}"; do
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
