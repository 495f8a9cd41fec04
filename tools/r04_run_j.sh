#!/bin/bash
mkdir -p gpurun_out/r04
TSAMD_DEBUG=1 python -m pytest tests/test_gpu_multirank.py -m gpu -q -k "cannot_be_resident" 2>&1 | grep "tsamd rank\|synchronize failed\|passed\|failed\|---- rank\|PASSED\|FAILED" | head -40 > gpurun_out/r04/j_dbg.log
python -m pytest tests/test_gpu_hybrid.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r04/j_tests.log
rm -f gpurun_out/r04/j_bench.log
for cfg in "20 1000000 200000 300" "20 600000 200000 300" "8 2000000 100000 500" "12 1200000 100000 300"; do
  set -- $cfg
  python bench.py --pops $1 --individuals $2 --snps $3 --steps $4 --warmup 50 --cpu-seconds 0 --no-profile 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    try:
        d = json.loads(ln)
    except Exception:
        continue
    print(d['config']['n'], d['config']['k'], d['value'], d['ms_per_step'])
" >> gpurun_out/r04/j_bench.log 2>&1
done
UNIT=hyb bash tools/variant.sh hybtime20 20 -DTSAMD_SCHED_TIME > /dev/null 2>&1
TSAMD_LIB=terastructure_amd/lib/variants/libtsamd_hybtime20.so python bench.py --pops 20 --individuals 1000000 --snps 100000 --steps 300 --warmup 50 --cpu-seconds 0 --no-profile 2>&1 | grep "ts_hybrid n=300" | tail -1 >> gpurun_out/r04/j_bench.log
cat gpurun_out/r04/j_dbg.log; tail -3 gpurun_out/r04/j_tests.log; cat gpurun_out/r04/j_bench.log
