#!/bin/bash
mkdir -p gpurun_out/r04
TSAMD_DEBUG=1 python -m pytest tests/test_gpu_multirank.py -m gpu -q -k "cannot_be_resident and 2-40000" 2>&1 | grep "tsamd rank\|synchronize failed\|passed\|failed" | head -20 > gpurun_out/r04/h_dbg.log
bash tools/r04_timers.sh > gpurun_out/r04/h_timers.log 2>&1
cat gpurun_out/r04/h_dbg.log; grep "ts_holblock n=\|report 2" gpurun_out/r04/h_timers.log | tail -3
