cd $GRAFT_REPO_ROOT
echo "### config 2: N=10K L=100K K=6"
timeout 600 python bench.py --individuals 10000 --snps 100000 --pops 6 --steps 20000 --warmup 1000 --cpu-seconds 8 2>/dev/null | cut -c1-8000
echo "### config 3: N=100K L=500K K=8"
timeout 600 python bench.py --individuals 100000 --snps 500000 --pops 8 --steps 10000 --warmup 500 --cpu-seconds 8 2>/dev/null | cut -c1-8000
echo "### K=20 N=1M (config 5 single GPU, L limited)"
timeout 900 python bench.py --pops 20 --snps 200000 --steps 500 --warmup 50 --cpu-seconds 0 2>/dev/null | cut -c1-8000
