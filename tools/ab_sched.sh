# A/B of whole-schedule-kernel variants on one box: updates/s of the default library and of the variants
# built by `UNIT=sched tools/variant.sh <name> 8 <flags>`.   usage: VARIANTS="a b" bash tools/ab_sched.sh [bench args]
cd $GRAFT_REPO_ROOT
ARGS="${@:---steps 2000 --warmup 200}"
for rep in 1 2 3; do
  for v in "" $VARIANTS; do
    L="TSAMD_X=1"; [ -n "$v" ] && L="TSAMD_LIB=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants/libtsamd_$v.so"
    env $L python3 bench.py $ARGS --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v:-default}', d['value'], 'updates/s')"
  done
done
