# in-kernel stage stamps (variants built with -DTSAMD_TRACE): usage: VARIANTS="tr pftr" bash tools/trace2.sh [bench args]
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
ARGS="${@:---snps 2000 --pops 8 --steps 60 --warmup 0}"
for v in $VARIANTS; do
  echo "### $v $ARGS"
  TSAMD_LIB=$V/libtsamd_$v.so python3 bench.py $ARGS --cpu-seconds 0 --no-profile 2>&1 | grep "^trace" | head -24
done
