cd $GRAFT_REPO_ROOT
run() { # variant, bench args
  v=$1; shift
  L="TSAMD_X=1"; [ -n "$v" ] && L="TSAMD_LIB=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants/libtsamd_$v.so"
  echo "### variant '$v' $*"
  bash tools/prof.sh v$v $L -- "$@" --cpu-seconds 0 2>&1 | grep -E "ts_pass<8|^value" | cut -c1-150
}
for v in "" $VARIANTS "" $VARIANTS; do
  run "$v" --snps 4000 --steps 300 --warmup 50
  run "$v" --individuals 125000 --snps 20000 --steps 2000 --warmup 100
done
