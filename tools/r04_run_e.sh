#!/bin/bash
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_holblock.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r04/e_tests.log
bash tools/r04_timers.sh > gpurun_out/r04/e_timers.log 2>&1
rm -f gpurun_out/r04/e_bench.log
for cfg in "20 1000000 200000 300" "20 500000 200000 500" "8 2000000 100000 500"; do
  set -- $cfg
  python bench.py --pops $1 --individuals $2 --snps $3 --steps $4 --warmup 50 --cpu-seconds 0 --no-profile 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try:
        d = json.loads(ln)
    except Exception:
        continue
    print(d['config']['n'], d['config']['k'], d['value'], d['ms_per_step'])
" >> gpurun_out/r04/e_bench.log 2>&1
done
tail -4 gpurun_out/r04/e_tests.log; cat gpurun_out/r04/e_timers.log | tail -12; cat gpurun_out/r04/e_bench.log
