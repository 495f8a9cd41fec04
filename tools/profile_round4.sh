# Round-4 profile (run on the GPU box): kernel-trace summaries of the default bench (K = 8, N = 1M: one ts_schedule launch per
# schedule, ts_holblock launches of the validation-block leg), of BASELINE config 5 on ONE GPU (K = 20, N = 1M) and of its
# 2-GPU shard (N = 500K) -- ts_hybrid, new this round -- and of N = 2M, K = 8; HBM traffic, fp64 instruction and SQ counters in
# separate --pmc runs; in-kernel timers of the diagnostic builds (ts_schedule<8>, ts_holblock<8>, ts_hybrid<20>, ts_hybrid<8>);
# the validation block at config 4; short-vs-long bench comparison.
# Writes under gpurun_out/prof_r04/ ; copy what is to be judged into profiles/ (tools/pmc_record.py reads it from there).
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r04; mkdir -p $O
K20="--pops 20 --snps 200000 --steps 300 --warmup 50 --validation-locs 0"
K20H="--pops 20 --individuals 500000 --snps 200000 --steps 500 --warmup 50 --validation-locs 0"
K8B="--pops 8 --individuals 2000000 --snps 100000 --steps 500 --warmup 50 --validation-locs 0"
bash tools/prof.sh default -- > $O/k8_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_default/bench.log > $O/k8_bench_under_rocprof.json
bash tools/prof.sh k20 -- $K20 --cpu-seconds 0 > $O/k20_n1m_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_k20/bench.log > $O/k20_n1m_bench_under_rocprof.json
bash tools/prof.sh k20h -- $K20H --cpu-seconds 0 > $O/k20_n500k_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_k20h/bench.log > $O/k20_n500k_bench_under_rocprof.json
bash tools/prof.sh k8b -- $K8B --cpu-seconds 0 > $O/k8_n2m_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_k8b/bench.log > $O/k8_n2m_bench_under_rocprof.json
# counters: the largest launch of a run is the 200-update launch (per update = max / 200)
A="--steps 200 --warmup 10 --ramp-seconds 0 --cpu-seconds 0 --no-profile --l 20000"
F64="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQC_ICACHE_REQ SQC_ICACHE_MISSES"
bash tools/pmc.sh fetch FETCH_SIZE -- $A > $O/k8_pmc_fetch_size.txt 2>&1
bash tools/pmc.sh write WRITE_SIZE -- $A > $O/k8_pmc_write_size.txt 2>&1
bash tools/pmc.sh f64 "$F64" -- $A > $O/k8_pmc_f64.txt 2>&1
bash tools/pmc.sh sq1 "$SQ1" -- $A > $O/k8_pmc_sq1.txt 2>&1
bash tools/pmc.sh sq2 "$SQ2" -- $A > $O/k8_pmc_sq2.txt 2>&1
bash tools/pmc.sh fetch20 FETCH_SIZE -- $A --pops 20 > $O/k20_n1m_pmc_fetch_size.txt 2>&1
bash tools/pmc.sh write20 WRITE_SIZE -- $A --pops 20 > $O/k20_n1m_pmc_write_size.txt 2>&1
bash tools/pmc.sh f6420 "$F64" -- $A --pops 20 > $O/k20_n1m_pmc_f64.txt 2>&1
bash tools/pmc.sh sq120 "$SQ1" -- $A --pops 20 > $O/k20_n1m_pmc_sq1.txt 2>&1
# in-kernel timers (diagnostic builds: tools/variant.sh with -DTSAMD_SCHED_TIME)
UNIT=sched bash tools/variant.sh time8 8 -DTSAMD_SCHED_TIME > /dev/null 2>&1
UNIT=hol bash tools/variant.sh holtime 8 -DTSAMD_SCHED_TIME > /dev/null 2>&1
UNIT=hyb bash tools/variant.sh hybtime20 20 -DTSAMD_SCHED_TIME > /dev/null 2>&1
UNIT=hyb bash tools/variant.sh hybtime8 8 -DTSAMD_SCHED_TIME > /dev/null 2>&1
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
T="--steps 2000 --warmup 200 --cpu-seconds 0 --no-profile"
{ TSAMD_LIB=$V/libtsamd_time8.so python3 bench.py $T 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | sed "s/^/N=1000000 K=8: /"
} > $O/sched_timers.txt 2>&1
{ TSAMD_LIB=$V/libtsamd_hybtime20.so python3 bench.py --pops 20 --snps 100000 --steps 300 --warmup 50 --cpu-seconds 0 --no-profile 2>&1 | grep "ts_hybrid n=300" | tail -1 | sed "s/^/N=1000000 K=20: /"
  TSAMD_LIB=$V/libtsamd_hybtime20.so python3 bench.py --pops 20 --individuals 500000 --snps 100000 --steps 300 --warmup 50 --cpu-seconds 0 --no-profile 2>&1 | grep "ts_hybrid n=300" | tail -1 | sed "s/^/N=500000 K=20: /"
  TSAMD_LIB=$V/libtsamd_hybtime8.so python3 bench.py --pops 8 --individuals 2000000 --snps 100000 --steps 300 --warmup 50 --cpu-seconds 0 --no-profile 2>&1 | grep "ts_hybrid n=300" | tail -1 | sed "s/^/N=2000000 K=8: /"
} > $O/hybrid_timers.txt 2>&1
# the validation block at config 4 (5 000 locations x 10 000 held-out individuals): batched, entry by entry, with the batch's timers
{ python3 tools/validation_block.py 1000000 2>&1 | grep "validation sample\|^report"
  TSAMD_HOLBLOCK=0 python3 tools/validation_block.py 1000000 2>&1 | grep "^report" | sed "s/^/TSAMD_HOLBLOCK=0 /"
  TSAMD_LIB=$V/libtsamd_holtime.so python3 tools/validation_block.py 200000 2>&1 | grep "ts_holblock n=" | tail -1
} > $O/validation_block.txt 2>&1
{ for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 20 warmup 5:', d['value'], 'updates/s', d['ms_per_step'], 'ms/step')"; done
  python3 bench.py --gpus 1 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 2000 warmup 200:', d['value'], 'updates/s', d['ms_per_step'], 'ms/step')"; } > $O/short_vs_long.txt 2>&1
find gpurun_out -name "*.db" -delete  # (the summaries are what is kept; gpurun merges at most 64 MiB back)
tail -n 14 $O/*kernel_trace.txt $O/short_vs_long.txt $O/sched_timers.txt $O/hybrid_timers.txt $O/validation_block.txt | cut -c1-260
for f in $O/*pmc_*.txt; do echo "== $f"; grep -E "ts_pass|ts_resident|ts_schedule|ts_hybrid|ts_holblock" $f | cut -c1-170 | head -12; done
