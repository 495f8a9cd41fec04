#!/bin/bash
# round 5, GPU call U: ts_schedule's launch prologue (gamma / c_n of the LDS items requested BEFORE the weights fill the register
# file: all loads in flight at once instead of forty dependent round trips) -- A/B on short and long launches
# (-DTSAMD_LATE_LDS_FILL = the old order); ts_hybhol with unconditional requests and 3 register items at K = 20; parity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
O=gpurun_out/r05
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
UNIT=sched bash tools/variant.sh late_k8 8 -DTSAMD_LATE_LDS_FILL > /dev/null 2>&1
run() { env $1 python3 bench.py $2 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$3', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us per update')"; }
{
for rep in 1 2 3 4; do
  run TSAMD_X=1 "--steps 20 --warmup 5" "20 updates, early fill (the build):"
  run TSAMD_LIB=$V/libtsamd_late_k8.so "--steps 20 --warmup 5" "20 updates, late fill (before):   "
done
for rep in 1 2; do
  run TSAMD_X=1 "--steps 2000 --warmup 200 --snps 50000" "2000 updates, early fill:"
  run TSAMD_LIB=$V/libtsamd_late_k8.so "--steps 2000 --warmup 200 --snps 50000" "2000 updates, late fill: "
done
run TSAMD_X=1 "--steps 100 --warmup 10" "100 updates, early fill:"
run TSAMD_LIB=$V/libtsamd_late_k8.so "--steps 100 --warmup 10" "100 updates, late fill: "
echo "-- validation block, N = 1M, K = 20"
python3 tools/validation_block.py 1000000 20 2>&1 | grep "^report [12]"
} > $O/u_prologue.txt 2>&1
cat $O/u_prologue.txt
rm -f $V/*.so
timeout 1500 python3 -m pytest tests/test_gpu_hybhol.py tests/test_gpu_parity.py tests/test_gpu_launch_modes.py tests/test_gpu_edges.py tests/test_gpu_multirank.py -q -x > $O/u_tests.log 2>&1
tail -3 $O/u_tests.log
