#!/bin/bash
# round 4: in-kernel timers of ts_holblock<8> (validation block at N = 1M) and ts_hybrid<20> / <8> (diagnostic builds)
mkdir -p gpurun_out/r04
UNIT=hol bash tools/variant.sh holtime 8 -DTSAMD_SCHED_TIME > /dev/null 2>&1
UNIT=hyb bash tools/variant.sh hybtime20 20 -DTSAMD_SCHED_TIME > /dev/null 2>&1
UNIT=hyb bash tools/variant.sh hybtime8 8 -DTSAMD_SCHED_TIME > /dev/null 2>&1
V=terastructure_amd/lib/variants
TSAMD_LIB=$V/libtsamd_holtime.so python tools/validation_block.py 100000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/r04/d_holblock_timers.txt
TSAMD_LIB=$V/libtsamd_hybtime20.so python bench.py --pops 20 --individuals 1000000 --snps 100000 --steps 300 --warmup 50 --cpu-seconds 0 --no-profile 2>&1 | grep "ts_hybrid n=" | tail -3 > gpurun_out/r04/d_hybrid_timers.txt
TSAMD_LIB=$V/libtsamd_hybtime20.so python bench.py --pops 20 --individuals 500000 --snps 100000 --steps 300 --warmup 50 --cpu-seconds 0 --no-profile 2>&1 | grep "ts_hybrid n=" | tail -2 >> gpurun_out/r04/d_hybrid_timers.txt
TSAMD_LIB=$V/libtsamd_hybtime8.so python bench.py --pops 8 --individuals 2000000 --snps 100000 --steps 300 --warmup 50 --cpu-seconds 0 --no-profile 2>&1 | grep "ts_hybrid n=" | tail -2 >> gpurun_out/r04/d_hybrid_timers.txt
cat gpurun_out/r04/d_holblock_timers.txt gpurun_out/r04/d_hybrid_timers.txt
