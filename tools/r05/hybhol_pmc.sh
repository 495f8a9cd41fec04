#!/bin/bash
# round 5: HBM / fabric traffic of ts_hybhol<20> (config 5 on one GPU) from the PMC counters: FETCH_SIZE and WRITE_SIZE in separate
# passes of the same bench command (its validation-block leg launches the kernel on 299 locations at a time)
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r05; mkdir -p $O
A="--pops 20 --steps 50 --warmup 10 --ramp-seconds 0 --cpu-seconds 0 --validation-locs 300 --l 20000"
bash tools/pmc.sh hhfetch FETCH_SIZE -- $A > $O/k20_n1m_hybhol_pmc_fetch_size.txt 2>&1
bash tools/pmc.sh hhwrite WRITE_SIZE -- $A > $O/k20_n1m_hybhol_pmc_write_size.txt 2>&1
find gpurun_out -name "*.db" -delete
grep "ts_hybhol" $O/k20_n1m_hybhol_pmc_*.txt | cut -c1-220
