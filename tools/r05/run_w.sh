#!/bin/bash
# round 5, GPU call W: what do the gamma step's CONDITIONAL stores cost?  Behind `if (mine)` the compiler must assume the stores may
# not have been issued, so the next item's first use of its loaded gamma waits with a count that, when they were, also waits for
# them (vmcnt counts stores on gfx950).  Timing experiment at N = 1 048 576 (every thread owns all 16 items: the stores can be
# unconditional there): -DTSAMD_EXP_UNCOND_STORES against the build.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
O=gpurun_out/r05
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
UNIT=sched bash tools/variant.sh unc_k8 8 -DTSAMD_EXP_UNCOND_STORES > /dev/null 2>&1 &
UNIT=sched bash tools/variant.sh unct_k8 8 -DTSAMD_EXP_UNCOND_STORES -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
UNIT=sched bash tools/variant.sh t_k8 8 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
wait
ab() { env $1 python3 bench.py --individuals 1048576 --snps 50000 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], 'updates/s', round(1e3*d['ms_per_step'],2), 'us per update')"; }
{
for rep in 1 2 3; do
  ab TSAMD_X=1 "conditional stores (the build):"
  ab TSAMD_LIB=$V/libtsamd_unc_k8.so "unconditional stores:          "
done
TSAMD_LIB=$V/libtsamd_t_k8.so python3 bench.py --individuals 1048576 --snps 50000 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | sed "s/^/conditional:   /"
TSAMD_LIB=$V/libtsamd_unct_k8.so python3 bench.py --individuals 1048576 --snps 50000 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | sed "s/^/unconditional: /"
} > $O/w_stores.txt 2>&1
cat $O/w_stores.txt
rm -f $V/*.so
