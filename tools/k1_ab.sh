#!/bin/bash
# A/B of the K1 sweeps on the GPU box: bricks + culling (default), bricks without culling, rows.
# usage: tools/k1_ab.sh <outdir>
OUT=${1:-gpurun_out/k1_ab}
mkdir -p $OUT
for res in 256 512; do
  python tools/kbench.py --res $res --reps 30 > $OUT/kb_${res}_bricks.txt 2>&1
  DFH_OPTIONS=k1_bricks_nocull=1 python tools/kbench.py --res $res --reps 30 > $OUT/kb_${res}_bricks_nocull.txt 2>&1
  DFH_OPTIONS=k1_no_bricks=1 python tools/kbench.py --res $res --reps 30 > $OUT/kb_${res}_rows.txt 2>&1
done
python tools/kbench_views.py --res 512 --views 8 --orbit > $OUT/kv_512_8_bricks.txt 2>&1
DFH_OPTIONS=k1_bricks_nocull=1 python tools/kbench_views.py --res 512 --views 8 --orbit > $OUT/kv_512_8_bricks_nocull.txt 2>&1
DFH_OPTIONS=k1_no_bricks=1 python tools/kbench_views.py --res 512 --views 8 --orbit > $OUT/kv_512_8_rows.txt 2>&1
python tools/kbench_views.py --res 256 --views 4 > $OUT/kv_256_4_bricks.txt 2>&1
DFH_OPTIONS=k1_no_bricks=1 python tools/kbench_views.py --res 256 --views 4 > $OUT/kv_256_4_rows.txt 2>&1
tail -n 7 $OUT/*.txt
