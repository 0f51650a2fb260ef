#!/bin/bash
# true kernel durations (rocprofv3) of K1 variants; prints median per view chunk
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "seg:" "old:DFH_NO_PYRAMID=1 DFH_BATCH=0" "b2:DFH_NO_PYRAMID=1 DFH_BATCH=2" "b4:DFH_NO_PYRAMID=1 DFH_BATCH=4" "b8:DFH_NO_PYRAMID=1 DFH_BATCH=8"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  for res in 256 512; do
    reps=30; [ $res = 512 ] && reps=12
    echo "== $tag res=$res ($envs)"
    env $envs $ROOT/tools/prof_kbench.sh ${tag}_$res --res $res --reps $reps --angles 0,-45,60,135 2>&1 | grep -E "integrate" | cut -c1-160
  done
done
