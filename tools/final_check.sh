cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
for rep in 1 2 3; do
for cfg in "12 600000 shared12" "12 50000 shared12" "20 125000 shared20" "20 327680 shared20" "20 20000 shared20"; do set -- $cfg
  for v in default $3; do
    L="TSAMD_X=1"; [ $v != default ] && L="TSAMD_LIB=$V/libtsamd_$v.so"
    env $L python3 bench.py --pops $1 --individuals $2 --snps 50000 --steps 3000 --warmup 300 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$1 N=$2 $v:', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')"
  done
done; done
