cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2 | head -1
for rep in 1 2 3; do for cfg in "8 4096" "8 100000" "8 1000000" "20 125000"; do set -- $cfg; for v in default old; do
  L="TSAMD_X=1"; [ $v != default ] && L="TSAMD_LIB=$V/libtsamd_$v.so"
  env $L python3 bench.py --pops $1 --individuals $2 --snps 50000 --steps 3000 --warmup 300 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$1 N=$2 $v:', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')"
done; done; done
