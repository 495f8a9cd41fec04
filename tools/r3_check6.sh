cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c6; mkdir -p $O
timeout 1200 python3 -m pytest tests -x -q -m gpu -k "16 or bench" > $O/t_k16.log 2>&1; echo "pytest k16+bench rc=$?" | tee -a $O/summary.txt; tail -3 $O/t_k16.log
bash tools/configs.sh > $O/other_configs.txt 2>&1
python3 - <<'PY' | tee -a gpurun_out/r3c6/summary.txt
import json
for ln in open('gpurun_out/r3c6/other_configs.txt'):
    if ln.startswith('###') or ln.startswith('K='): print(ln.strip())
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline'] or {}
        print('   ', d['value'], 'updates/s', 'per update us', r.get('per_update_us'), r.get('bound'), 'frac', r.get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('value_1_thread'), 'parity', (d.get('parity_vs_cpu_baseline') or {}).get('ok'))
PY
python3 tools/single_update_rate.py 1000000 8 600 2>/dev/null | tee -a $O/summary.txt
python3 tools/single_update_rate.py 500000 16 600 2>/dev/null | tee -a $O/summary.txt
bash tools/rehearse_multi.sh 8 250000 20 > $O/rehearse_8x_k20.log 2>&1; tail -4 $O/rehearse_8x_k20.log | cut -c1-1500 | tee -a $O/summary.txt
bash tools/rehearse_multi.sh 2 400000 8 > $O/rehearse_2x_k8.log 2>&1; tail -3 $O/rehearse_2x_k8.log | cut -c1-1500 | tee -a $O/summary.txt
