# A/B of the default library against variants built from another tree / with other flags, over several shapes on one box:
#   VARIANTS="old" bash tools/ab_lib.sh            (terastructure_amd/lib/variants/libtsamd_<name>.so)
# each line: shape, library, updates/s, us per update (two rounds, interleaved)
cd $GRAFT_REPO_ROOT
ab() { # label, bench args
  for rep in 1 2; do for v in "" $VARIANTS; do
    L="TSAMD_X=1"; [ -n "$v" ] && L="TSAMD_LIB=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants/libtsamd_$v.so"
    env $L python3 bench.py $2 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '${v:-default}', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')"
  done; done
}
if [ -n "$SHAPES" ]; then
  while read -r k n steps; do [ -n "$k" ] && ab "K=$k N=$n" "--pops $k --individuals $n --snps 50000 --steps $steps --warmup 300"; done <<< "$SHAPES"
else
  ab "N=1M K=8" "--steps 2000 --warmup 200"
  ab "N=100K K=8" "--individuals 100000 --snps 100000 --steps 6000 --warmup 500"
  ab "N=500K K=16" "--individuals 500000 --snps 100000 --pops 16 --steps 2000 --warmup 200"
  ab "N=125K K=20" "--individuals 125000 --snps 100000 --pops 20 --steps 4000 --warmup 400"
  ab "N=10K K=6" "--individuals 10000 --snps 100000 --pops 6 --steps 10000 --warmup 1000"
fi
