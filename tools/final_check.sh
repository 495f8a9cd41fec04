cd $GRAFT_REPO_ROOT
O=gpurun_out/final3; mkdir -p $O
timeout 3300 python3 -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $O/summary.txt
tail -3 $O/t_all.log
for cfg in "3 200" "8 940" "8 1718" "8 4096" "8 10000" "6 10000" "20 512" "20 1024" "8 1000000"; do set -- $cfg
  python3 bench.py --pops $1 --individuals $2 --snps 50000 --steps 6000 --warmup 500 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$1 N=$2:', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')" | tee -a $O/summary.txt
done
cd tests/golden/ref_data 2>/dev/null && ls | head -3
