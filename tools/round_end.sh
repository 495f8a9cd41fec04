# End-of-round evidence on the GPU box: the whole -m gpu suite, then the round's profiles (tools/profile_round3.sh) and the
# other configurations (tools/configs.sh).  Writes under gpurun_out/round_end/ and gpurun_out/prof_r03/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/round_end; mkdir -p $O
timeout 3300 python3 -m pytest tests -x -q -m gpu --durations=6 > $O/t_all.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $O/summary.txt
tail -10 $O/t_all.log
bash tools/profile_round3.sh > $O/profile.log 2>&1; echo "profile rc=$?" | tee -a $O/summary.txt
bash tools/configs.sh > $O/other_configs.txt 2>&1
python3 - <<'PY' | tee -a gpurun_out/round_end/summary.txt
import json
for ln in open('gpurun_out/round_end/other_configs.txt'):
    if ln.startswith('###') or ln.startswith('K='): print(ln.strip())
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline'] or {}
        print('   ', d['value'], 'updates/s', 'per update us', r.get('per_update_us'), r.get('bound'), 'frac', r.get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('value_1_thread'), 'parity', (d.get('parity_vs_cpu_baseline') or {}).get('ok'))
PY
for n in 1718 4096; do python3 bench.py --pops 8 --individuals $n --snps 50000 --steps 6000 --warmup 500 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=8 N=$n:', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')" | tee -a $O/summary.txt; done
cat gpurun_out/prof_r03/short_vs_long.txt gpurun_out/prof_r03/sched_timers.txt | tee -a $O/summary.txt
