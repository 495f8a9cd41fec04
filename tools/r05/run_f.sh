#!/bin/bash
# round 5, GPU call F: ts_hybrid's L2 touches issued by the wave that does not poll (A/B, timers); the split-commit test
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
UNIT=hyb bash tools/variant.sh l2w2_k20 20 -DTSAMD_HY_L2PF=2 > /dev/null 2>&1 &
UNIT=hyb bash tools/variant.sh l2w1_k20 20 -DTSAMD_HY_L2PF=1 > /dev/null 2>&1 &
UNIT=hyb bash tools/variant.sh l2wt_k20 20 -DTSAMD_HY_L2PF=2 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
UNIT=hyb bash tools/variant.sh l2w2_k8 8 -DTSAMD_HY_L2PF=2 > /dev/null 2>&1 &
wait
ab() { for rep in 1 2 3; do for v in "" $3; do
    L="TSAMD_X=1"; [ -n "$v" ] && L="TSAMD_LIB=$V/libtsamd_$v.so"
    env $L python3 bench.py $2 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '${v:-default}', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')"
  done; done; }
{
ab "N=1M K=20" "--pops 20 --individuals 1000000 --snps 100000 --steps 600 --warmup 100" "l2w2_k20 l2w1_k20"
ab "N=2M K=8" "--pops 8 --individuals 2000000 --snps 100000 --steps 600 --warmup 100" "l2w2_k8"
echo "== timers, touches by the last wave, L2PF = 2, N=1M K=20"
TSAMD_LIB=$V/libtsamd_l2wt_k20.so python3 bench.py --pops 20 --snps 100000 --steps 300 --warmup 50 --cpu-seconds 0 --no-profile 2>&1 | grep "ts_hybrid n=300" | tail -1
} > $O/f_hybrid_l2w.txt 2>&1
cat $O/f_hybrid_l2w.txt
rm -f $V/*.so
timeout 900 python3 -m pytest tests/test_gpu_multirank.py -q -k "split_commit or cannot_be_resident or hybrid_validation" --durations=5 > $O/f_tests.log 2>&1
tail -12 $O/f_tests.log
