# Round profile (run on the GPU box): kernel-trace summaries of the default bench (K = 8: one ts_schedule launch per
# schedule), of the same workload with one launch per SNP (TSAMD_PERSISTENT=0: first pass + ts_resident) and of K = 20
# (one launch per pass), HBM traffic counters and SQ counters in separate --pmc runs, the short-vs-long bench comparison.
# Writes under gpurun_out/prof_r02/ ; copy what is to be judged into profiles/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r02; mkdir -p $O
bash tools/prof.sh default -- > $O/k8_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_default/bench.log > $O/k8_bench_under_rocprof.json
bash tools/prof.sh persnp TSAMD_PERSISTENT=0 -- --cpu-seconds 0 > $O/k8_per_snp_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_persnp/bench.log > $O/k8_per_snp_bench_under_rocprof.json
bash tools/prof.sh k20 -- --pops 20 --snps 200000 --steps 300 --warmup 50 --cpu-seconds 0 > $O/k20_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_k20/bench.log > $O/k20_bench_under_rocprof.json
# counters: the largest launch of a run is the 200-update ts_schedule launch (per update = max / 200)
A="--steps 200 --warmup 10 --ramp-seconds 0 --cpu-seconds 0 --no-profile --l 20000"
bash tools/pmc.sh fetch FETCH_SIZE -- $A > $O/k8_pmc_fetch_size.txt 2>&1
bash tools/pmc.sh write WRITE_SIZE -- $A > $O/k8_pmc_write_size.txt 2>&1
bash tools/pmc.sh sq1 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY" -- $A > $O/k8_pmc_sq1.txt 2>&1
bash tools/pmc.sh sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQC_ICACHE_REQ SQC_ICACHE_MISSES" -- $A > $O/k8_pmc_sq2.txt 2>&1
B="--steps 60 --warmup 10 --ramp-seconds 0 --cpu-seconds 0 --no-profile --l 20000"
TSAMD_PERSISTENT=0 bash tools/pmc.sh fetchps FETCH_SIZE -- $B > $O/k8_per_snp_pmc_fetch_size.txt 2>&1
TSAMD_PERSISTENT=0 bash tools/pmc.sh writeps WRITE_SIZE -- $B > $O/k8_per_snp_pmc_write_size.txt 2>&1
bash tools/pmc.sh fetch20 FETCH_SIZE -- $B --pops 20 > $O/k20_pmc_fetch_size.txt 2>&1
bash tools/pmc.sh write20 WRITE_SIZE -- $B --pops 20 > $O/k20_pmc_write_size.txt 2>&1
{ for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 20 warmup 5:', d['value'], 'updates/s', d['ms_per_step'], 'ms/step')"; done
  python3 bench.py --gpus 1 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 2000 warmup 200:', d['value'], 'updates/s', d['ms_per_step'], 'ms/step')"; } > $O/short_vs_long.txt 2>&1
find gpurun_out -name "*.db" -delete  # (the summaries are what is kept; gpurun merges at most 64 MiB back)
tail -n 14 $O/k8_kernel_trace.txt $O/k8_per_snp_kernel_trace.txt $O/k20_kernel_trace.txt $O/short_vs_long.txt | cut -c1-200
for f in $O/k8_pmc_*.txt $O/k8_per_snp_pmc_*.txt $O/k20_pmc_*.txt; do echo "== $f"; grep -E "ts_pass|ts_resident|ts_schedule" $f | cut -c1-170 | head -20; done
