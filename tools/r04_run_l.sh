#!/bin/bash
# rehearsals of bench.py --gpus N on one GPU (functional only) + the round's profile
mkdir -p gpurun_out/r04
for cfg in "2 1000000 20" "8 250000 20" "8 1000000 8" "4 1000000 20"; do
  set -- $cfg
  echo "== rehearse_multi $cfg" >> gpurun_out/r04/l_rehearse.log
  timeout 600 bash tools/rehearse_multi.sh $1 $2 $3 2>&1 | grep "exchange self-test:\|^{\|exit code\|Error\|error" | cut -c1-1800 >> gpurun_out/r04/l_rehearse.log
done
bash tools/profile_round4.sh > gpurun_out/r04/l_profile.log 2>&1
tail -30 gpurun_out/r04/l_rehearse.log | cut -c1-600
