# what the steady-state K3 kernel is bound by: builds with parts stubbed out (tools/build_variant.sh k3stub<bits> -DDFH_K3_STUB=<bits>;
# 1 live gathers, 2 stores, 4 weight loads, 8 DQ rows, 16 index loads)
for b in 0 1 2 4 16 23; do
  lib=build_variants/k3stub$b.so; [ $b = 0 ] && lib=dynamicfusion_body_amd/libdfusion_hip.so
  for o in "k3_tpb=1024" "k3_tpb=1024"; do
    echo "== stub $b $o"; timeout -k 10 100 python tools/kbench_fv.py --lib $lib --options $o 2>&1 | grep "indices + weights"
  done
done
