#!/bin/bash
# round 5, GPU call T: ts_hybhol's streamed requests made unconditional (the conditional ones forced s_waitcnt vmcnt(0) where the
# paths join: every second item's load did not overlap the arithmetic) -- the validation block at N = 1M, K = 20 with 2 locations
# per sweep (2 and 3 register items) and with 3 per sweep (no register items); timers; parity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
O=gpurun_out/r05
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
UNIT=hhol bash tools/variant.sh hh2r3_k20 20 -DTSAMD_HH_BUDGET=185 > /dev/null 2>&1 &
UNIT=hhol bash tools/variant.sh hh3_k20 20 -DTSAMD_HH_SUB=3 -DTSAMD_HH_BUDGET=100 > /dev/null 2>&1 &
UNIT=hhol bash tools/variant.sh hht_k20 20 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
UNIT=hhol bash tools/variant.sh hh3t_k20 20 -DTSAMD_HH_SUB=3 -DTSAMD_HH_BUDGET=100 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
wait
{
for rep in 1 2; do
  echo "-- 2 per sweep, 2 register items (the build)"; python3 tools/validation_block.py 1000000 20 2>&1 | grep "^report [12]"
  echo "-- 2 per sweep, 3 register items"; TSAMD_LIB=$V/libtsamd_hh2r3_k20.so python3 tools/validation_block.py 1000000 20 2>&1 | grep "^report [12]"
  echo "-- 3 per sweep, no register items"; TSAMD_LIB=$V/libtsamd_hh3_k20.so python3 tools/validation_block.py 1000000 20 2>&1 | grep "^report [12]"
done
TSAMD_LIB=$V/libtsamd_hht_k20.so python3 tools/validation_block.py 200000 20 2>&1 | grep "ts_hybhol n=" | tail -1
TSAMD_LIB=$V/libtsamd_hh3t_k20.so python3 tools/validation_block.py 200000 20 2>&1 | grep "ts_hybhol n=" | tail -1
} > $O/t_hh.txt 2>&1
cat $O/t_hh.txt
rm -f $V/*.so
timeout 900 python3 -m pytest tests/test_gpu_hybhol.py -q > $O/t_tests.log 2>&1
tail -3 $O/t_tests.log
