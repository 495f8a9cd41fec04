#!/bin/bash
# round 5, GPU call M: the per-wave form's convergence decision from the lanes' registers (A/B against the sequential LDS sum,
# -DTSAMD_SEQ_DECISION); parity tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
for k in 8 6; do UNIT=sched bash tools/variant.sh seq_k$k $k -DTSAMD_SEQ_DECISION > /dev/null 2>&1 & done
wait
ab() { for rep in 1 2 3; do for v in "" seq; do
    L="TSAMD_X=1"; [ -n "$v" ] && L="TSAMD_LIB=$V/libtsamd_${v}_k$2.so"
    env $L python3 bench.py $3 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '${v:-default(row16)}', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')"
  done; done; }
{
ab "N=1M K=8" 8 "--snps 50000 --steps 2000 --warmup 200"
ab "N=100K K=8" 8 "--individuals 100000 --snps 100000 --steps 6000 --warmup 500"
ab "N=10K K=6" 6 "--individuals 10000 --snps 100000 --pops 6 --steps 10000 --warmup 1000"
ab "N=4096 K=8" 8 "--individuals 4096 --snps 100000 --steps 10000 --warmup 1000"
} > $O/m_row16_ab.txt 2>&1
cat $O/m_row16_ab.txt
rm -f $V/*.so
timeout 1500 python3 -m pytest tests/test_gpu_geometry.py tests/test_gpu_launch_modes.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_parity_at_size.py -q > $O/m_tests.log 2>&1
tail -3 $O/m_tests.log
