#!/bin/bash
# round 5, GPU call E: ts_hybrid with the next pass' streamed items touched into the L2 during the exchange (A/B); ts_hybhol with
# three locations per sweep at K = 20 (A/B of the validation block); the multi-rank tests; the whole suite after the trims
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
UNIT=hyb bash tools/variant.sh l2pf2_k20 20 -DTSAMD_HY_L2PF=2 > /dev/null 2>&1 &
UNIT=hyb bash tools/variant.sh l2pf3_k20 20 -DTSAMD_HY_L2PF=3 > /dev/null 2>&1 &
UNIT=hyb bash tools/variant.sh l2pf2_k8 8 -DTSAMD_HY_L2PF=2 > /dev/null 2>&1 &
UNIT=hyb bash tools/variant.sh l2pft_k20 20 -DTSAMD_HY_L2PF=2 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
UNIT=hhol bash tools/variant.sh hh3_k20 20 -DTSAMD_HH_SUB_MID=3 > /dev/null 2>&1 &
UNIT=hhol bash tools/variant.sh hh3t_k20 20 -DTSAMD_HH_SUB_MID=3 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
wait
ab() { for rep in 1 2; do for v in "" $3; do
    L="TSAMD_X=1"; [ -n "$v" ] && L="TSAMD_LIB=$V/libtsamd_$v.so"
    env $L python3 bench.py $2 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '${v:-default}', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')"
  done; done; }
{
ab "N=1M K=20" "--pops 20 --individuals 1000000 --snps 100000 --steps 600 --warmup 100" "l2pf2_k20 l2pf3_k20"
ab "N=2M K=8" "--pops 8 --individuals 2000000 --snps 100000 --steps 600 --warmup 100" "l2pf2_k8"
echo "== timers, L2PF = 2, N=1M K=20"
TSAMD_LIB=$V/libtsamd_l2pft_k20.so python3 bench.py --pops 20 --snps 100000 --steps 300 --warmup 50 --cpu-seconds 0 --no-profile 2>&1 | grep "ts_hybrid n=300" | tail -1
} > $O/e_hybrid_l2pf.txt 2>&1
cat $O/e_hybrid_l2pf.txt
{
for v in "" hh3_k20; do
  L="TSAMD_X=1"; [ -n "$v" ] && L="TSAMD_LIB=$V/libtsamd_$v.so"
  echo "== validation block N=1M K=20, ${v:-default (two locations per sweep)}"
  env $L python3 tools/validation_block.py 200000 20 2>&1 | grep "^report"
done
TSAMD_LIB=$V/libtsamd_hh3t_k20.so python3 tools/validation_block.py 200000 20 2>&1 | grep "ts_hybhol n=" | tail -1
echo "== hybhol tests on the three-location build"
TSAMD_LIB=$V/libtsamd_hh3_k20.so timeout 600 python3 -m pytest tests/test_gpu_hybhol.py -q -k "20" 2>&1 | tail -3
} > $O/e_hybhol_sub3.txt 2>&1
cat $O/e_hybhol_sub3.txt
rm -f $V/*.so
timeout 2400 python3 -m pytest tests -m gpu -q --durations=40 > $O/e_suite.log 2>&1
tail -6 $O/e_suite.log
