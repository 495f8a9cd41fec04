#!/bin/bash
# round 4: ts_hybrid / ts_holblock parity tests and the above-capacity shapes through bench.py (gpurun -- 'bash tools/r04_hybrid_check.sh')
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_hybrid.py tests/test_gpu_holblock.py tests/test_gpu_launch_modes.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r04/c_tests.log
rm -f gpurun_out/r04/c_bench.log
for cfg in "20 1000000 200000 300" "20 500000 200000 500" "8 2000000 100000 500" "16 786432 100000 500"; do
  set -- $cfg
  python bench.py --pops $1 --individuals $2 --snps $3 --steps $4 --warmup 50 --cpu-seconds 0 2>gpurun_out/r04/c_bench_$1_$2.err | python -c "
import sys, json
for ln in sys.stdin:
    try:
        d = json.loads(ln)
    except Exception:
        continue
    print(d['config'], d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('kernel'))
" >> gpurun_out/r04/c_bench.log 2>&1
done
python tools/validation_block.py 200000 > gpurun_out/r04/c_valblock.log 2>&1
tail -5 gpurun_out/r04/c_tests.log; cat gpurun_out/r04/c_bench.log; tail -3 gpurun_out/r04/c_valblock.log
