# round 3, second GPU call: the whole -m gpu suite, single-update rates, A/B of ts_schedule variants, in-kernel timers
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c2; mkdir -p $O
timeout 3000 python3 -m pytest tests -x -q -m gpu --durations=15 > $O/t_all.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $O/summary.txt
tail -25 $O/t_all.log
python3 tools/single_update_rate.py 1000000 8 600 2>/dev/null | tee -a $O/summary.txt
python3 tools/single_update_rate.py 125000 20 600 2>/dev/null | tee -a $O/summary.txt
VARIANTS="nodefer vec1" bash tools/ab_sched.sh 2>&1 | tee -a $O/summary.txt
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
TSAMD_LIB=$V/libtsamd_time8.so python3 bench.py --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | tee -a $O/summary.txt
TSAMD_LIB=$V/libtsamd_time16.so python3 bench.py --pops 16 --individuals 500000 --snps 100000 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | tee -a $O/summary.txt
TSAMD_LIB=$V/libtsamd_time20.so python3 bench.py --pops 20 --individuals 125000 --snps 100000 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | tee -a $O/summary.txt
TSAMD_LIB=$V/libtsamd_time20.so python3 bench.py --pops 20 --individuals 327680 --snps 100000 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | tee -a $O/summary.txt
