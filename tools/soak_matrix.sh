mkdir -p gpurun_out/r5o
for cfg in "RELAX=1.0" "RELAX=0.8" "RELAX=0.5" "RELAX=0.2" "RELAX=0.0"; do
  env $cfg timeout -k 10 200 python tools/soak.py 400 1000 > "gpurun_out/r5o/soakc_${cfg// /_}.txt" 2>&1
  echo "== $cfg"; grep -A1 "last 50" "gpurun_out/r5o/soakc_${cfg// /_}.txt" | grep -B1 "frame 100,\|frame 200,\|frame 400,\|Error" | cut -c1-200
done
