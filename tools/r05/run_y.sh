#!/bin/bash
# round 5, GPU call Y: the bench lines of the final build as the driver runs them (plain, not under rocprof) and smoke()
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
O=gpurun_out/r05
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/y_smoke.log 2>&1; tail -1 $O/y_smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' > $O/y_bench_driver_args.json
python3 bench.py 2>/dev/null | grep '^{' > $O/y_bench_default_args.json
python3 -c "
import json
for f in ('driver','default'):
    d=json.load(open('$O/y_bench_%s_args.json'%f)); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['frac'], r['executed']['frac'], d['cpu_baseline']['value'], d['parity_vs_cpu_baseline'].get('ok'), set(d['legs'].values()))
"
