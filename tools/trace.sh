# in-kernel stage timestamps of the pass kernels (variant built with -DTSAMD_TRACE; run on the GPU box)
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
echo "### N=10K K=6"; TSAMD_LIB=$V/libtsamd_trace6.so python bench.py --individuals 10000 --snps 2000 --pops 6 --steps 60 --warmup 0 --cpu-seconds 0 --no-profile 2>&1 | grep "^trace" | head -44
echo "### N=125K K=8"; TSAMD_LIB=$V/libtsamd_trace8.so python bench.py --individuals 125000 --snps 2000 --pops 8 --steps 60 --warmup 0 --cpu-seconds 0 --no-profile 2>&1 | grep "^trace" | head -44
echo "### N=1M K=8"; TSAMD_LIB=$V/libtsamd_trace8.so python bench.py --snps 2000 --pops 8 --steps 60 --warmup 0 --cpu-seconds 0 --no-profile 2>&1 | grep "^trace" | head -44
