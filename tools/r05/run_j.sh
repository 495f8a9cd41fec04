#!/bin/bash
# round 5, GPU call J: per-wave epilogue form with exp(Elogbeta) taken from the lanes' registers (v_readlane) instead of the wave's
# LDS row: A/B at the small-shard shapes (variant: -DTSAMD_REPL_READLANE=0 = rounds 3-4); parity tests of the small geometries
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
for k in 8 6 3; do UNIT=sched bash tools/variant.sh rl0_k$k $k -DTSAMD_REPL_READLANE=0 > /dev/null 2>&1 & done
wait
ab() { for rep in 1 2 3; do for v in "" rl0; do
    L="TSAMD_X=1"; [ -n "$v" ] && L="TSAMD_LIB=$V/libtsamd_${v}_k$2.so"
    env $L python3 bench.py $3 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '${v:-default(readlane)}', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')"
  done; done; }
{
ab "N=100K K=8" 8 "--individuals 100000 --snps 100000 --steps 6000 --warmup 500"
ab "N=300K K=8" 8 "--individuals 300000 --snps 100000 --steps 4000 --warmup 400"
ab "N=10K K=6" 6 "--individuals 10000 --snps 100000 --pops 6 --steps 10000 --warmup 1000"
ab "N=4096 K=8" 8 "--individuals 4096 --snps 100000 --steps 10000 --warmup 1000"
ab "N=200 K=3" 3 "--individuals 200 --snps 10000 --pops 3 --steps 20000 --warmup 1000"
} > $O/j_readlane_ab.txt 2>&1
cat $O/j_readlane_ab.txt
rm -f $V/*.so
timeout 1200 python3 -m pytest tests/test_gpu_geometry.py tests/test_gpu_launch_modes.py tests/test_gpu_parity.py tests/test_gpu_edges.py -q > $O/j_tests.log 2>&1
tail -3 $O/j_tests.log
