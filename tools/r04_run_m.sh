#!/bin/bash
# hybrid benches under rocprof again (roofline object present now), the other configurations, the sharded hybrid tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r04; mkdir -p $O gpurun_out/r04
python -m pytest tests/test_gpu_multirank.py -m gpu -q -k "hybrid" 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r04/m_tests.log
K20="--pops 20 --snps 200000 --steps 300 --warmup 50 --validation-locs 0"
K20H="--pops 20 --individuals 500000 --snps 200000 --steps 500 --warmup 50 --validation-locs 0"
K8B="--pops 8 --individuals 2000000 --snps 100000 --steps 500 --warmup 50 --validation-locs 0"
bash tools/prof.sh k20 -- $K20 --cpu-seconds 0 > $O/k20_n1m_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_k20/bench.log > $O/k20_n1m_bench_under_rocprof.json
bash tools/prof.sh k20h -- $K20H --cpu-seconds 0 > $O/k20_n500k_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_k20h/bench.log > $O/k20_n500k_bench_under_rocprof.json
bash tools/prof.sh k8b -- $K8B --cpu-seconds 0 > $O/k8_n2m_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_k8b/bench.log > $O/k8_n2m_bench_under_rocprof.json
bash tools/configs.sh > $O/other_configs.txt 2>&1
find gpurun_out -name "*.db" -delete
tail -5 gpurun_out/r04/m_tests.log
python3 - <<'PY'
import json
for f in ('k20_n1m','k20_n500k','k8_n2m'):
    d=json.loads(open(f'gpurun_out/prof_r04/{f}_bench_under_rocprof.json').read())
    r=d['roofline'] or {}
    print(f, d['value'], d['ms_per_step'], r.get('bound'), r.get('achieved'), r.get('frac'), r.get('traffic'))
PY
grep "^###\|launch per pass" -A0 $O/other_configs.txt | head -40
