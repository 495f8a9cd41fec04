#!/bin/bash
# round 5, GPU call S: ts_hybhol<20> with THREE locations per sweep and no register items (-DTSAMD_HH_SUB=3: no spill once the
# register items go; streamed 13/16 x 2 sub-batches per batch of 6 instead of 11/16 x 3): validation block A/B, timers, parity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
O=gpurun_out/r05
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
UNIT=hhol bash tools/variant.sh hh3_k20 20 -DTSAMD_HH_SUB=3 -DTSAMD_HH_BUDGET=100 > /dev/null 2>&1 &
UNIT=hhol bash tools/variant.sh hh3t_k20 20 -DTSAMD_HH_SUB=3 -DTSAMD_HH_BUDGET=100 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
wait
{
for rep in 1 2; do
  echo "-- default (2 per sweep)"; python3 tools/validation_block.py 1000000 20 2>&1 | grep "^report [12]"
  echo "-- 3 per sweep"; TSAMD_LIB=$V/libtsamd_hh3_k20.so python3 tools/validation_block.py 1000000 20 2>&1 | grep "^report [12]"
done
TSAMD_LIB=$V/libtsamd_hh3t_k20.so python3 tools/validation_block.py 200000 20 2>&1 | grep "ts_hybhol n=" | tail -1
} > $O/s_hh3.txt 2>&1
cat $O/s_hh3.txt
TSAMD_LIB=$V/libtsamd_hh3_k20.so timeout 900 python3 -m pytest tests/test_gpu_hybhol.py -q -k "20 or bitwise" > $O/s_tests.log 2>&1
tail -3 $O/s_tests.log
rm -f $V/*.so
