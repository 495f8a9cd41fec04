#!/bin/bash
# round 5, GPU call Z: the per-wave epilogue with the pair sum's exp(psi) split on the upper half-wave (lane 32 + j) beside the value's
# own on lane j -- one split on a pass' critical path instead of two (-DTSAMD_REPL_HALVES=0 = before): A/B, timers, parity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
O=gpurun_out/r05
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
for k in 8 6; do UNIT=sched bash tools/variant.sh hv0_k$k $k -DTSAMD_REPL_HALVES=0 > /dev/null 2>&1 & done
UNIT=sched bash tools/variant.sh time_k8 8 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
wait
ab() { for rep in 1 2 3; do for v in "" hv0; do
    L="TSAMD_X=1"; [ -n "$v" ] && L="TSAMD_LIB=$V/libtsamd_${v}_k$2.so"
    env $L python3 bench.py $3 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '${v:-halves (the build)}', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')"
  done; done; }
{
ab "N=1M K=8" 8 "--snps 50000 --steps 2000 --warmup 200"
ab "N=100K K=8" 8 "--individuals 100000 --snps 100000 --steps 6000 --warmup 500"
ab "N=4096 K=8" 8 "--individuals 4096 --snps 100000 --steps 10000 --warmup 1000"
ab "N=10K K=6" 6 "--individuals 10000 --snps 100000 --pops 6 --steps 10000 --warmup 1000"
echo "== timers N=1M K=8"
TSAMD_LIB=$V/libtsamd_time_k8.so python3 bench.py --snps 50000 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>/dev/null | grep "ts_schedule n=2000" | tail -1
} > $O/z_halves_ab.txt 2>&1
cat $O/z_halves_ab.txt
rm -f $V/*.so
timeout 1200 python3 -m pytest tests/test_gpu_geometry.py tests/test_gpu_launch_modes.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_parity_at_size.py tests/test_gpu_recovery.py -q > $O/z_tests.log 2>&1
tail -3 $O/z_tests.log
