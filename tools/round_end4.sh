#!/bin/bash
# what the driver runs at round end, on one box: the GPU suite, smoke, the bench line with the driver's arguments and the default one
mkdir -p gpurun_out/r04
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r04/z_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/z_smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/z_bench_driver.json 2> gpurun_out/r04/z_bench_driver.err
python bench.py > gpurun_out/r04/z_bench_default.json 2> gpurun_out/r04/z_bench_default.err
tail -3 gpurun_out/r04/z_tests.log; tail -1 gpurun_out/r04/z_smoke.log
python - <<'PY'
import json
for f in ('driver','default'):
    for ln in open(f'gpurun_out/r04/z_bench_{f}.json'):
        try: d=json.loads(ln)
        except Exception: continue
        r=d['roofline']
        print(f, d['value'], d['ms_per_step'], r['frac'], r.get('flops_source','')[:60], '|', r.get('counter_records','')[:60], '|', (d.get('validation_block') or {}).get('seconds_per_report'), d['parity_vs_cpu_baseline']['ok'])
PY
