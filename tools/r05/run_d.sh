#!/bin/bash
# round 5, GPU call D: ts_hybhol (tests, timers, the validation block at N = 1M, K = 20 and N = 2M, K = 8); A/B of the wave-cooperative
# convergence decision; the whole GPU suite with durations
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
timeout 1500 python3 -m pytest tests/test_gpu_hybhol.py tests/test_gpu_hybrid.py -q --durations=15 > $O/d_hybhol_tests.log 2>&1
tail -5 $O/d_hybhol_tests.log
UNIT=hhol bash tools/variant.sh hhtime20 20 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
UNIT=hhol bash tools/variant.sh hhtime8 8 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
UNIT=sched bash tools/variant.sh seq20 20 -DTSAMD_SEQ_DECISION > /dev/null 2>&1 &
UNIT=sched bash tools/variant.sh seq16 16 -DTSAMD_SEQ_DECISION > /dev/null 2>&1 &
wait
{
echo "== N=1M K=20 (config 5 on one GPU): bench line with the validation block"
TSAMD_LIB=$V/libtsamd_hhtime20.so timeout 900 python3 bench.py --pops 20 --individuals 1000000 --snps 200000 --steps 300 --warmup 50 --cpu-seconds 0 2>&1 | grep "ts_hybhol n=\|^{" | tail -3
echo "== N=2M K=8"
TSAMD_LIB=$V/libtsamd_hhtime8.so timeout 900 python3 bench.py --pops 8 --individuals 2000000 --snps 100000 --steps 300 --warmup 50 --cpu-seconds 0 2>&1 | grep "ts_hybhol n=\|^{" | tail -3
} > $O/d_hybhol_bench.txt 2>&1
python3 - <<'PY'
import json
for ln in open("gpurun_out/r05/d_hybhol_bench.txt"):
    if ln.startswith("{"):
        d = json.loads(ln); print(d["metric"], d["value"], json.dumps(d["validation_block"]))
    else:
        print(ln.strip()[:400])
PY
ab() { for rep in 1 2; do for v in "" $4; do
    L="TSAMD_X=1"; [ -n "$v" ] && L="TSAMD_LIB=$V/libtsamd_$v.so"
    env $L python3 bench.py $3 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '${v:-default}', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')"
  done; done; }
{
ab "N=125K K=20" 20 "--individuals 125000 --snps 100000 --pops 20 --steps 4000 --warmup 400" seq20
ab "N=500K K=16" 16 "--individuals 500000 --snps 100000 --pops 16 --steps 2000 --warmup 200" seq16
} > $O/d_decision_ab.txt 2>&1
cat $O/d_decision_ab.txt
rm -f $V/*.so
timeout 2400 python3 -m pytest tests -m gpu -q --durations=150 > $O/d_suite.log 2>&1
tail -6 $O/d_suite.log
