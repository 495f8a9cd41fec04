# A/B harness on the GPU box: per-kernel times (rocprofv3 kernel trace) of the default library and of
# variant libraries built by tools/variant.sh.   usage: VARIANTS="a b" bash tools/ab.sh [bench args]
cd $GRAFT_REPO_ROOT
ARGS="${@:---snps 4000 --steps 300 --warmup 50}"
run() { # variant
  v=$1
  L="TSAMD_X=1"; [ -n "$v" ] && L="TSAMD_LIB=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants/libtsamd_$v.so"
  echo "### variant '${v:-default}' $ARGS"
  bash tools/prof.sh v${v:-default} $L -- $ARGS --cpu-seconds 0 --no-profile 2>&1 | grep -E "ts_pass<|^value" | cut -c1-190
}
for rep in 1 2; do for v in "" $VARIANTS; do run "$v"; done; done
