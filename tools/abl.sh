cd $GRAFT_REPO_ROOT
run() { # variant, env...
  v=$1; shift
  L=""; [ -n "$v" ] && L="TSAMD_LIB=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants/libtsamd_$v.so"
  echo "### variant '$v' $*"
  bash tools/prof.sh v$v $L "$@" -- --snps 4000 --steps 300 --warmup 50 --cpu-seconds 0 2>&1 | grep -E "ts_pass<8, true|^value" | cut -c1-150
}
for v in $VARIANTS; do run "$v"; run "$v"; done
