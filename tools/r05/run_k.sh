#!/bin/bash
# round 5, GPU call K: the per-wave epilogue form (now with exp(Elogbeta) from the lanes' registers) on the FULL-SIZE instantiation
# (-DTSAMD_REPL_ALL; round 3: 2.4 % slower than the shared form there) -- A/B at N = 1M, K = 8
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
UNIT=sched bash tools/variant.sh replall_k8 8 -DTSAMD_REPL_ALL > /dev/null 2>&1
{ for rep in 1 2 3; do for v in "" replall_k8; do
    L="TSAMD_X=1"; [ -n "$v" ] && L="TSAMD_LIB=$V/libtsamd_$v.so"
    env $L python3 bench.py --snps 50000 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=1M K=8', '${v:-default}', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us')"
  done; done; } > $O/k_replall_ab.txt 2>&1
cat $O/k_replall_ab.txt
rm -f $V/*.so
