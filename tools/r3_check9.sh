cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c9; mkdir -p $O
timeout 3300 python3 -m pytest tests -x -q -m gpu --durations=6 > $O/t_all.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $O/summary.txt
tail -12 $O/t_all.log
for cfg in "6 10000" "3 2000" "8 1718" "8 16000" "8 30000" "8 100000" "20 4000" "8 1000000"; do set -- $cfg
  python3 bench.py --pops $1 --individuals $2 --snps 100000 --steps 4000 --warmup 400 --cpu-seconds 0 --no-profile 2>>$O/bench_err.log | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$1 N=$2 schedule:', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')" | tee -a $O/summary.txt
done
timeout 900 bash tools/rehearse_multi.sh 4 160000 16 > $O/rehearse_4x_k16.log 2>&1; grep -E "exchange self-test:|^\{|exit code" $O/rehearse_4x_k16.log | cut -c1-900 | tee -a $O/summary.txt
