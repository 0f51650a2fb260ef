# K3 stored-neighbourhood kernel variants (tools/kbench_fv.py --options ...)
for o in "k3_no_lds=1" "k3_tpb=1024" "k3_tpb=512" "k3_tpb=256" "k3_tpb=256" "k3_tpb=512" "k3_tpb=1024" "k3_tpb=1024,k3_wg_per_cu=1" "k3_tpb=512,k3_wg_per_cu=2"; do
  echo "== $o"; timeout -k 10 100 python tools/kbench_fv.py --options $o 2>&1 | grep "indices + weights"
done
