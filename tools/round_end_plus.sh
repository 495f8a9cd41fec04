# tools/round_end.sh plus the driver-style bench line on the same box (smoke first)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round_end
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/round_end/smoke.log 2>&1; echo "smoke rc=$?" | tee gpurun_out/round_end/summary.txt
bash tools/round_end.sh
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/round_end/bench_err.log | grep '^{' > gpurun_out/round_end/bench.json; python3 -c "
import json; d=json.load(open('gpurun_out/round_end/bench.json')); r=d['roofline']
print('bench --steps 20 --warmup 5:', d['value'], 'updates/s;', r['bound'], r['frac'], 'hbm', r['hbm']['frac'], 'latency', r['latency']['frac_of_update'], 'cpu', d['cpu_baseline']['value'], 'parity', d['parity_vs_cpu_baseline']['ok'], d['parity_vs_cpu_baseline'].get('other_launch_modes'))" | tee -a gpurun_out/round_end/summary.txt
for cfg in "3 200" "8 940"; do set -- $cfg
  python3 bench.py --pops $1 --individuals $2 --snps 50000 --steps 6000 --warmup 500 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$1 N=$2:', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')" | tee -a gpurun_out/round_end/summary.txt
done
