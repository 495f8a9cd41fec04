# the BASELINE configurations (and the K > 8 / above-capacity shapes of rounds 3 and 4) through the same bench.py on one MI355X, all data resident
cd $GRAFT_REPO_ROOT
echo "### config 2: N=10K L=100K K=6"
timeout 600 python bench.py --individuals 10000 --snps 100000 --pops 6 --steps 20000 --warmup 1000 --cpu-seconds 8 2>/dev/null | cut -c1-9000
echo "### config 3: N=100K L=500K K=8"
timeout 600 python bench.py --individuals 100000 --snps 500000 --pops 8 --steps 10000 --warmup 500 --cpu-seconds 8 2>/dev/null | cut -c1-9000
echo "### N=600K K=12"
timeout 600 python bench.py --individuals 600000 --snps 200000 --pops 12 --steps 2000 --warmup 200 --cpu-seconds 6 2>/dev/null | cut -c1-9000
echo "### N=500K K=16"
timeout 600 python bench.py --individuals 500000 --snps 200000 --pops 16 --steps 2000 --warmup 200 --cpu-seconds 6 2>/dev/null | cut -c1-9000
echo "### config 5's 8-GPU shard on one GPU: N=125K K=20"
timeout 600 python bench.py --individuals 125000 --snps 200000 --pops 20 --steps 4000 --warmup 400 --cpu-seconds 6 2>/dev/null | cut -c1-9000
echo "### config 5's 4-GPU-class shard on one GPU: N=327680 K=20"
timeout 600 python bench.py --individuals 327680 --snps 200000 --pops 20 --steps 2000 --warmup 200 --cpu-seconds 6 2>/dev/null | cut -c1-9000
echo "### K=20 N=1M (config 5 on ONE GPU: ts_hybrid, L limited)"
timeout 900 python bench.py --pops 20 --snps 200000 --steps 500 --warmup 50 --cpu-seconds 6 2>/dev/null | cut -c1-9000
echo "### K=20 N=500K (config 5's 2-GPU shard on one GPU: ts_hybrid, all on chip)"
timeout 900 python bench.py --pops 20 --individuals 500000 --snps 200000 --steps 1000 --warmup 100 --cpu-seconds 6 2>/dev/null | cut -c1-9000
echo "### K=8 N=2M (ts_hybrid)"
timeout 900 python bench.py --pops 8 --individuals 2000000 --snps 100000 --steps 1000 --warmup 100 --cpu-seconds 6 2>/dev/null | cut -c1-9000
echo "### config 1's shape: N=200 K=3"
timeout 600 python bench.py --individuals 200 --snps 10000 --pops 3 --steps 20000 --warmup 1000 --cpu-seconds 4 2>/dev/null | cut -c1-9000
echo "### the same shapes, one launch per pass (TSAMD_RESIDENT=0)"
for cfg in "6 10000" "8 100000" "12 600000" "16 500000" "20 125000" "20 327680" "20 500000" "20 1000000" "8 2000000"; do set -- $cfg
  TSAMD_RESIDENT=0 python bench.py --pops $1 --individuals $2 --snps 100000 --steps 500 --warmup 50 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$1 N=$2 launch per pass:', d['value'], 'updates/s')"
done
