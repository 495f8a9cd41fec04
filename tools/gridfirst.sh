# first-pass grid sweep (TSAMD_GRID_FIRST): per-kernel times from rocprofv3
cd $GRAFT_REPO_ROOT
for g in ${GRIDS:-512 768 1024 1536 2048}; do
  echo "### TSAMD_GRID_FIRST=$g"
  bash tools/prof.sh gf$g TSAMD_GRID_FIRST=$g -- --snps 4000 --steps 300 --warmup 50 --cpu-seconds 0 --no-profile 2>&1 | grep -E "ts_pass<|^value" | cut -c1-190
done
