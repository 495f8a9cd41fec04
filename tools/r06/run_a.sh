# Round 6, call a: the sharded validation blocks after the order fix, the kernel families whose item counts changed (ts_hybrid K = 9 / 29..32,
# K = 22, sharded K = 14 / 16), then the geometry sweep of the exchange-bound shard shapes: workgroups per rank
# 64 / 128 / 256 at (125K, K = 8) and (125K, K = 20) on one rank (TSAMD_SCHED_WORKGROUPS), three repeats each.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
(timeout 1500 python3 -m pytest tests/test_gpu_multirank.py tests/test_gpu_hybrid.py tests/test_gpu_holblock.py tests/test_gpu_hybhol.py tests/test_gpu_launch_modes.py -q -m gpu --durations=12 > $O/a_tests.log 2>&1; echo "exit $?" >> $O/a_tests.log)
tail -25 $O/a_tests.log
A="--steps 2000 --warmup 200 --cpu-seconds 0 --no-profile --validation-locs 0 --snps 100000"
for rep in 1 2 3; do
  for shape in "8 125000" "20 125000" "8 100000"; do
    set -- $shape
    for wg in 64 128 256; do
      TSAMD_SCHED_WORKGROUPS=$wg python3 bench.py --pops $1 --individuals $2 $A 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('sweep K=$1 N=$2 workgroups<=$wg rep $rep:', round(d['value'], 1), 'updates/s', round(1e3 * d['ms_per_step'], 2), 'us/update', d['config'].get('kernel', ''))
" >> $O/a_geometry_sweep.txt
    done
  done
done
cat $O/a_geometry_sweep.txt
