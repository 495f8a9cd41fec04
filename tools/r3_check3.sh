# round 3, third GPU call: single-individual items as the default geometry -- the whole -m gpu suite, then A/B + timers
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c3; mkdir -p $O
python3 bench.py --steps 2000 --warmup 200 --cpu-seconds 6 2>$O/bench_err.log | grep '^{' > $O/bench_default.json; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print('default bench:', d['value'], 'updates/s; parity', d['parity_vs_cpu_baseline']); print('roofline', {k: v for k, v in d['roofline'].items() if k in ('bound','achieved','frac','per_update_us')})" | tee -a $O/summary.txt
timeout 3300 python3 -m pytest tests -x -q -m gpu --durations=15 > $O/t_all.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $O/summary.txt
tail -25 $O/t_all.log
VARIANTS="vec2 allpartial exptab treesum" bash tools/ab_sched.sh 2>&1 | tee -a $O/summary.txt
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
T="--steps 2000 --warmup 200 --cpu-seconds 0 --no-profile"
TSAMD_LIB=$V/libtsamd_time8.so python3 bench.py $T 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | tee -a $O/summary.txt
TSAMD_LIB=$V/libtsamd_time16.so python3 bench.py --pops 16 --individuals 500000 --snps 100000 $T 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | tee -a $O/summary.txt
TSAMD_LIB=$V/libtsamd_time20.so python3 bench.py --pops 20 --individuals 125000 --snps 100000 $T 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | tee -a $O/summary.txt
TSAMD_LIB=$V/libtsamd_time20.so python3 bench.py --pops 20 --individuals 327680 --snps 100000 $T 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | tee -a $O/summary.txt
for cfg in "16 500000" "20 125000" "20 327680" "12 600000" "8 100000" "6 10000"; do set -- $cfg
  python3 bench.py --pops $1 --individuals $2 --snps 100000 --steps 1000 --warmup 100 --cpu-seconds 0 --no-profile 2>>$O/bench_err.log | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$1 N=$2 schedule:', d['value'], 'updates/s')" | tee -a $O/summary.txt
done
python3 tools/single_update_rate.py 1000000 8 600 2>/dev/null | tee -a $O/summary.txt
python3 tools/single_update_rate.py 125000 20 600 2>/dev/null | tee -a $O/summary.txt
