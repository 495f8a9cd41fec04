#!/bin/bash
# round 5, GPU call C: the lane epilogue -- whole GPU suite with durations, A/B of the three epilogue forms over the target
# shapes (0 = shared, 1 = lane form where the shared form ran, 2 = lane form everywhere), in-kernel timers
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
build() { UNIT=sched bash tools/variant.sh "$@" > /dev/null 2>&1; }
for k in 8 20; do
  build lane0_k$k $k -DTSAMD_LANE_EPILOGUE=0 &
  build lane2_k$k $k -DTSAMD_LANE_EPILOGUE=2 &
  build time_k$k $k -DTSAMD_SCHED_TIME &
  build time2_k$k $k -DTSAMD_SCHED_TIME -DTSAMD_LANE_EPILOGUE=2 &
done
build lane0_k16 16 -DTSAMD_LANE_EPILOGUE=0 &
build lane0_k6 6 -DTSAMD_LANE_EPILOGUE=0 &
build lane2_k6 6 -DTSAMD_LANE_EPILOGUE=2 &
wait
ab() { # label, K, bench args
  for rep in 1 2; do for v in "" lane0 lane2; do
    L="TSAMD_X=1"; [ -n "$v" ] && L="TSAMD_LIB=$V/libtsamd_${v}_k$2.so"
    [ -n "$v" ] && [ ! -f $V/libtsamd_${v}_k$2.so ] && continue
    env $L python3 bench.py $3 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '${v:-default(lane1)}', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')"
  done; done
}
{
ab "N=1M K=8" 8 "--snps 50000 --steps 2000 --warmup 200"
ab "N=100K K=8" 8 "--individuals 100000 --snps 100000 --steps 6000 --warmup 500"
ab "N=125K K=20" 20 "--individuals 125000 --snps 100000 --pops 20 --steps 4000 --warmup 400"
ab "N=500K K=16" 16 "--individuals 500000 --snps 100000 --pops 16 --steps 2000 --warmup 200"
ab "N=10K K=6" 6 "--individuals 10000 --snps 100000 --pops 6 --steps 10000 --warmup 1000"
ab "N=4096 K=8" 8 "--individuals 4096 --snps 100000 --steps 10000 --warmup 1000"
} > $O/c_lane_ab.txt 2>&1
{
for spec in "time_k8 8 1000000" "time_k8 8 100000" "time2_k8 8 100000" "time_k20 20 125000"; do
  set -- $spec
  echo "== $1 N=$3 K=$2"
  TSAMD_LIB=$V/libtsamd_$1.so timeout 300 python3 bench.py --individuals $3 --pops $2 --snps 50000 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>&1 | grep "ts_schedule n=2000" | tail -1
done
} > $O/c_lane_timers.txt 2>&1
rm -f $V/*.so
timeout 1700 python3 -m pytest tests -m gpu -q --durations=120 -x > $O/c_suite.log 2>&1
cat $O/c_lane_ab.txt $O/c_lane_timers.txt; tail -4 $O/c_suite.log
