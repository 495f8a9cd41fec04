#!/bin/bash
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r04/f_tests.log
python bench.py > gpurun_out/r04/f_bench.json 2> gpurun_out/r04/f_bench.err
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 20:', d['value'], d['ms_per_step'])" > gpurun_out/r04/f_short.log
bash tools/r04_timers.sh > gpurun_out/r04/f_timers.log 2>&1
tail -6 gpurun_out/r04/f_tests.log; cat gpurun_out/r04/f_short.log; grep "ts_holblock n=\|report 2" gpurun_out/r04/f_timers.log | tail -3
python - <<'PY'
import json
for ln in open('gpurun_out/r04/f_bench.json'):
    try: d=json.loads(ln)
    except Exception: continue
    print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('validation_block'))
PY
