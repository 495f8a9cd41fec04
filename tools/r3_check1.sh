# round 3, first GPU check: the new resident kernels (K <= 32, deferred exchange, entry probe + replay)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c1; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 1500 python3 -m pytest tests/test_gpu_launch_modes.py tests/test_gpu_recovery.py -x -q -m gpu > $O/t_modes.log 2>&1; echo "modes+recovery rc=$?" | tee -a $O/summary.txt
tail -5 $O/t_modes.log
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q -m gpu > $O/t_parity.log 2>&1; echo "parity+edges rc=$?" | tee -a $O/summary.txt
tail -3 $O/t_parity.log
for i in 1 2; do python3 bench.py --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>$O/bench_err.log | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=8 N=1M long:', d['value'], 'updates/s')"; done | tee -a $O/summary.txt
python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-profile 2>>$O/bench_err.log | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=8 N=1M short:', d['value'], 'updates/s')" | tee -a $O/summary.txt
for cfg in "16 500000" "20 125000" "20 327680" "12 600000"; do set -- $cfg
  python3 bench.py --pops $1 --individuals $2 --snps 100000 --steps 1000 --warmup 100 --cpu-seconds 0 --no-profile 2>>$O/bench_err.log | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$1 N=$2 schedule:', d['value'], 'updates/s')" | tee -a $O/summary.txt
  TSAMD_RESIDENT=0 python3 bench.py --pops $1 --individuals $2 --snps 100000 --steps 300 --warmup 50 --cpu-seconds 0 --no-profile 2>>$O/bench_err.log | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$1 N=$2 per pass:', d['value'], 'updates/s')" | tee -a $O/summary.txt
done
tail -5 $O/bench_err.log
