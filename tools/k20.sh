# first-pass times across K with the occupancy-derived grid
cd $GRAFT_REPO_ROOT
for K in 8 10 12 16 20 24; do
  echo "### K=$K"
  bash tools/prof.sh x -- --pops $K --snps 4000 --steps 150 --warmup 20 --cpu-seconds 0 2>&1 | grep -E "ts_pass<.*true|^value" | cut -c1-140
done
