#!/bin/bash
# round 5, experiment set 1: where the gamma step's non-arithmetic time goes (ts_schedule<8>, N = 1M).  Variants built on the box.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
build() { UNIT=sched bash tools/variant.sh "$@" > /dev/null 2>&1; }
build time0 8 -DTSAMD_SCHED_TIME &
build t_nogload 8 -DTSAMD_SCHED_TIME -DTSAMD_EXP_NOGLOAD &
build t_pinlds 8 -DTSAMD_SCHED_TIME -DTSAMD_EXP_PIN_LDS &
build t_pinall 8 -DTSAMD_SCHED_TIME -DTSAMD_EXP_PIN_ALL &
wait
build t_l2pf 8 -DTSAMD_SCHED_TIME -DTSAMD_EXP_L2PF &
build t_lds5 8 -DTSAMD_SCHED_TIME -DTSAMD_SCHED_LDS_ITEMS=5 &
build pinlds 8 -DTSAMD_EXP_PIN_LDS &
build l2pf 8 -DTSAMD_EXP_L2PF &
wait
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
ls -la $V > $O/a_variants.txt
ARGS="--snps 50000 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile"
for v in time0 t_nogload t_pinlds t_pinall t_l2pf t_lds5; do
  echo "== $v" >> $O/a_gamma_timers.txt
  TSAMD_LIB=$V/libtsamd_$v.so timeout 300 python3 bench.py $ARGS 2>&1 | grep "ts_schedule n=2000" | tail -1 >> $O/a_gamma_timers.txt
done
VARIANTS="pinlds l2pf" bash tools/ab_sched.sh $ARGS > $O/a_gamma_ab.txt 2>&1
cat $O/a_gamma_timers.txt $O/a_gamma_ab.txt
