cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c10; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu > $O/t_bench.log 2>&1; echo "bench contract rc=$?" | tee -a $O/summary.txt; tail -3 $O/t_bench.log
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
for rep in 1 2; do for n in 1718 4096 10000 16000; do for v in default one16 one0; do
  L="TSAMD_X=1"; [ $v != default ] && L="TSAMD_LIB=$V/libtsamd_$v.so"
  env $L python3 bench.py --pops 8 --individuals $n --snps 50000 --steps 6000 --warmup 500 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=$n $v:', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')" | tee -a $O/summary.txt
done; done; done
