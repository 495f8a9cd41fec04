#!/bin/bash
# round 5, GPU call B: gamma-step store experiments; bench contract tests (1 and 2 ranks); 4-rank K = 20 rehearsal; multirank tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
build() { UNIT=sched bash tools/variant.sh "$@" > /dev/null 2>&1; }
build t_same 8 -DTSAMD_SCHED_TIME -DTSAMD_EXP_STORE_SAME &
build t_same_nogload 8 -DTSAMD_SCHED_TIME -DTSAMD_EXP_STORE_SAME -DTSAMD_EXP_NOGLOAD &
build t_wbl2 8 -DTSAMD_SCHED_TIME -DTSAMD_EXP_WBL2 &
build wbl2 8 -DTSAMD_EXP_WBL2 &
wait
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
ARGS="--snps 50000 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile"
for v in t_same t_same_nogload t_wbl2; do
  echo "== $v" >> $O/b_gamma_timers.txt
  TSAMD_LIB=$V/libtsamd_$v.so timeout 300 python3 bench.py $ARGS 2>&1 | grep "ts_schedule n=2000\|rror" | tail -2 >> $O/b_gamma_timers.txt
done
VARIANTS="wbl2" bash tools/ab_sched.sh $ARGS > $O/b_gamma_ab.txt 2>&1
rm -f $V/*.so
timeout 1500 python3 -m pytest tests/test_gpu_bench_contract.py -x -q --durations=5 > $O/b_contract.log 2>&1
echo "== rehearse_multi 4 1000000 20" > $O/b_rehearse.log
timeout 900 bash tools/rehearse_multi.sh 4 1000000 20 >> $O/b_rehearse.log 2>&1
timeout 1500 python3 -m pytest tests/test_gpu_multirank.py -x -q --durations=10 > $O/b_multirank.log 2>&1
tail -3 $O/b_gamma_timers.txt $O/b_gamma_ab.txt; tail -5 $O/b_contract.log; grep -c roofline $O/b_rehearse.log; tail -5 $O/b_multirank.log
