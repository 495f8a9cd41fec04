cd $GRAFT_REPO_ROOT
run() { # variant, env...
  v=$1; shift
  L=""; [ -n "$v" ] && L="TSAMD_LIB=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants/libtsamd_$v.so"
  echo "### variant '$v' $*"
  bash tools/prof.sh v$v $L "$@" -- --snps 4000 --steps 300 --warmup 50 --cpu-seconds 0 2>&1 | grep -E "ts_pass<8, true" | cut -c1-150
}
for g in 512 1024 2048; do run abl2 TSAMD_GRID_FIRST=$g; done
for g in 512 1024 2048; do run abl2 TSAMD_GRID_FIRST=$g TSAMD_FIRST_VEC=2; done
for g in 768 1024 2048; do run w4 TSAMD_GRID_FIRST=$g; done
for g in 768 1024; do run w3 TSAMD_GRID_FIRST=$g; done
