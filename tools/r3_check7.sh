cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c7; mkdir -p $O
timeout 1500 python3 tools/stress_parity.py 300 31 > $O/stress.log 2>&1; echo "stress rc=$?" | tee -a $O/summary.txt; tail -4 $O/stress.log | tee -a $O/summary.txt
timeout 900 python3 tools/soak.py 300000 > $O/soak.log 2>&1; echo "soak rc=$?" | tee -a $O/summary.txt; tail -4 $O/soak.log | tee -a $O/summary.txt
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-style run:', d['value'], 'updates/s; roofline', d['roofline']['bound'], d['roofline']['frac'], 'hbm frac', d['roofline']['hbm']['frac'], 'latency', d['roofline']['latency']['frac_of_update'], 'parity', d['parity_vs_cpu_baseline']['ok'], 'cpu', d['cpu_baseline']['value'])" | tee -a $O/summary.txt; done
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/bench_driver_style.json
