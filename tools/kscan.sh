# per-kernel times across K at fixed N (run on the GPU box): tools/kscan.sh [N]
cd $GRAFT_REPO_ROOT
N=${1:-1000000}
for K in 4 8 12 16 20 24 32; do
  echo "### N=$N K=$K"
  bash tools/prof.sh k$K -- --individuals $N --snps 4000 --pops $K --steps 200 --warmup 30 --cpu-seconds 0 2>&1 | grep -E "ts_pass<|^value" | cut -c1-140
done
