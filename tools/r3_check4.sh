cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c4; mkdir -p $O
for i in 1 2 3; do python3 bench.py --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>>$O/bench_err.log | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=8 N=1M long:', d['value'], 'updates/s')"; done | tee -a $O/summary.txt
VARIANTS="nodefer wlds" bash tools/ab_sched.sh 2>&1 | tee -a $O/summary.txt
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
T="--steps 2000 --warmup 200 --cpu-seconds 0 --no-profile"
TSAMD_LIB=$V/libtsamd_time8.so python3 bench.py $T 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | tee -a $O/summary.txt
TSAMD_LIB=$V/libtsamd_wldstime.so python3 bench.py $T 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | sed "s/^/wlds: /" | tee -a $O/summary.txt
TSAMD_LIB=$V/libtsamd_wlds.so timeout 600 python3 -m pytest tests/test_gpu_launch_modes.py -x -q -m gpu -k "40000-8 or 200000-3 or deferred or cuts" 2>&1 | tail -2 | sed "s/^/wlds parity: /" | tee -a $O/summary.txt
TSAMD_LIB=$V/libtsamd_time16.so python3 bench.py --pops 16 --individuals 500000 --snps 100000 $T 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | tee -a $O/summary.txt
TSAMD_LIB=$V/libtsamd_time20.so python3 bench.py --pops 20 --individuals 125000 --snps 100000 $T 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | tee -a $O/summary.txt
for cfg in "16 500000" "20 125000" "20 327680" "12 600000" "8 100000" "6 10000"; do set -- $cfg
  python3 bench.py --pops $1 --individuals $2 --snps 100000 --steps 1000 --warmup 100 --cpu-seconds 0 --no-profile 2>>$O/bench_err.log | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$1 N=$2 schedule:', d['value'], 'updates/s')" | tee -a $O/summary.txt
done
python3 tools/single_update_rate.py 1000000 8 600 2>/dev/null | tee -a $O/summary.txt
timeout 3300 python3 -m pytest tests -x -q -m gpu --durations=8 > $O/t_all.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $O/summary.txt
tail -14 $O/t_all.log
