# the whole -m gpu suite, the driver-style bench line and the other configurations on the final build
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 3300 python3 -m pytest tests -x -q -m gpu --durations=5 > $O/t_all.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $O/summary.txt
tail -9 $O/t_all.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_err.log | grep '^{' > $O/bench.json; python3 -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']
print('bench --steps 20 --warmup 5:', d['value'], 'updates/s;', r['bound'], r['frac'], 'hbm', r['hbm']['frac'], 'latency', r['latency']['frac_of_update'], 'cpu', d['cpu_baseline']['value'], 'parity', d['parity_vs_cpu_baseline']['ok'])" | tee -a $O/summary.txt
bash tools/configs.sh > $O/other_configs.txt 2>&1
python3 - <<'PY' | tee -a gpurun_out/final/summary.txt
import json
for ln in open('gpurun_out/final/other_configs.txt'):
    if ln.startswith('###') or ln.startswith('K='): print(ln.strip())
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline'] or {}
        print('   ', d['value'], 'updates/s', 'per update us', r.get('per_update_us'), r.get('bound'), 'frac', r.get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('value_1_thread'), 'parity', (d.get('parity_vs_cpu_baseline') or {}).get('ok'))
PY
for cfg in "3 200" "8 940" "8 1718" "8 4096"; do set -- $cfg
  python3 bench.py --pops $1 --individuals $2 --snps 50000 --steps 6000 --warmup 500 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$1 N=$2:', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')" | tee -a $O/summary.txt
done
# the bench under rocprofv3 --kernel-trace for the three resident shapes (the roofline objects then use the committed counter record)
K16="--pops 16 --individuals 500000 --snps 200000"; K20S="--pops 20 --individuals 125000 --snps 200000"
bash tools/prof.sh default -- > $O/k8_kernel_trace.txt 2>&1; grep '^{' gpurun_out/prof_default/bench.log > $O/k8_bench_under_rocprof.json
bash tools/prof.sh k16 -- $K16 > $O/k16_n500k_kernel_trace.txt 2>&1; grep '^{' gpurun_out/prof_k16/bench.log > $O/k16_n500k_bench_under_rocprof.json
bash tools/prof.sh k20s -- $K20S > $O/k20_n125k_kernel_trace.txt 2>&1; grep '^{' gpurun_out/prof_k20s/bench.log > $O/k20_n125k_bench_under_rocprof.json
find gpurun_out -name "*.db" -delete
