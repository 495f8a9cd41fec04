#!/bin/bash
mkdir -p gpurun_out/r04
TSAMD_DEBUG=1 python -m pytest tests/test_gpu_multirank.py -m gpu -q -k "cannot_be_resident" 2>&1 | grep "tsamd rank\|synchronize failed\|passed\|failed" | head -20 > gpurun_out/r04/i_dbg.log
python -m pytest tests -m gpu -q --deselect tests/test_gpu_multirank.py::test_sharded_schedule_that_cannot_be_resident_is_replayed_on_every_rank 2>&1 | tail -12 > gpurun_out/r04/i_tests.log
python bench.py > gpurun_out/r04/i_bench.json 2> gpurun_out/r04/i_bench.err
for i in 1 2; do python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 20:', d['value'], d['ms_per_step'])"; done > gpurun_out/r04/i_short.log
cat gpurun_out/r04/i_dbg.log; tail -4 gpurun_out/r04/i_tests.log; cat gpurun_out/r04/i_short.log
python - <<'PY'
import json
for ln in open('gpurun_out/r04/i_bench.json'):
    try: d=json.loads(ln)
    except Exception: continue
    print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('validation_block'))
PY
