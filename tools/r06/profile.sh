# Round-6 profile (run on the GPU box, on the FINAL sources): kernel-trace summaries of the default bench (K = 8, N = 1M: one
# ts_schedule launch per schedule, ts_holblock launches of the validation-block leg), of BASELINE config 5 on ONE GPU (K = 20,
# N = 1M: ts_hybrid, and -- new this round -- ts_hybhol launches of its validation-block leg), of its 8-GPU shard (N = 125K, K = 20)
# and of N = 2M, K = 8; HBM traffic, fp64 instruction and SQ counters in separate --pmc runs; in-kernel timers of the diagnostic
# builds; short-vs-long bench comparison; the 2-rank and 4-rank rehearsals of `bench.py --gpus N` (all ranks on device 0).
# Writes under gpurun_out/prof_r06/ ; copy what is to be judged into profiles/r06_* (tools/pmc_record.py reads it from there).
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r06; mkdir -p $O
K20="--pops 20 --snps 200000 --steps 300 --warmup 50"
K20S="--pops 20 --individuals 125000 --snps 200000 --steps 2000 --warmup 200 --validation-locs 0"
K8B="--pops 8 --individuals 2000000 --snps 100000 --steps 500 --warmup 50"
bash tools/prof.sh default -- > $O/k8_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_default/bench.log > $O/k8_bench_under_rocprof.json
bash tools/prof.sh short -- --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --validation-locs 0 > $O/k8_short_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_short/bench.log > $O/k8_short_bench_under_rocprof.json
bash tools/prof.sh k20 -- $K20 --cpu-seconds 0 > $O/k20_n1m_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_k20/bench.log > $O/k20_n1m_bench_under_rocprof.json
bash tools/prof.sh k20s -- $K20S --cpu-seconds 0 > $O/k20_n125k_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_k20s/bench.log > $O/k20_n125k_bench_under_rocprof.json
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
bash tools/pmc.sh fetch20s FETCH_SIZE -- $A --pops 20 --individuals 125000 > $O/k20_n125k_pmc_fetch_size.txt 2>&1
bash tools/pmc.sh write20s WRITE_SIZE -- $A --pops 20 --individuals 125000 > $O/k20_n125k_pmc_write_size.txt 2>&1
bash tools/pmc.sh f6420s "$F64" -- $A --pops 20 --individuals 125000 > $O/k20_n125k_pmc_f64.txt 2>&1
# in-kernel timers (diagnostic builds: tools/variant.sh with -DTSAMD_SCHED_TIME)
UNIT=sched bash tools/variant.sh time8 8 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
UNIT=sched bash tools/variant.sh time20 20 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
UNIT=hol bash tools/variant.sh holtime 8 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
UNIT=hyb bash tools/variant.sh hybtime20 20 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
UNIT=hhol bash tools/variant.sh hhtime20 20 -DTSAMD_SCHED_TIME > /dev/null 2>&1 &
wait
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
T="--steps 2000 --warmup 200 --cpu-seconds 0 --no-profile --snps 50000"
{ TSAMD_LIB=$V/libtsamd_time8.so python3 bench.py $T 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | sed "s/^/N=1000000 K=8: /"
  TSAMD_LIB=$V/libtsamd_time8.so python3 bench.py $T --individuals 100000 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | sed "s/^/N=100000 K=8: /"
  TSAMD_LIB=$V/libtsamd_time20.so python3 bench.py $T --individuals 125000 --pops 20 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | sed "s/^/N=125000 K=20: /"
} > $O/sched_timers.txt 2>&1
{ TSAMD_LIB=$V/libtsamd_hybtime20.so python3 bench.py --pops 20 --snps 100000 --steps 300 --warmup 50 --cpu-seconds 0 --no-profile 2>&1 | grep "ts_hybrid n=300" | tail -1 | sed "s/^/N=1000000 K=20: /"
} > $O/hybrid_timers.txt 2>&1
# the validation block: config 4 (5 000 locations x 10 000 held-out individuals; ts_holblock) and config 5 on one GPU (N = 1M, K = 20: ts_hybhol)
{ python3 tools/validation_block.py 1000000 2>&1 | grep "validation sample\|^report"
  TSAMD_LIB=$V/libtsamd_holtime.so python3 tools/validation_block.py 200000 2>&1 | grep "ts_holblock n=" | tail -1
  echo "-- K = 20, N = 1M (ts_hybrid context): batched (ts_hybhol), then entry by entry"
  python3 tools/validation_block.py 1000000 20 2>&1 | grep "validation sample\|^report"
  TSAMD_HOLBLOCK=0 python3 tools/validation_block.py 200000 20 2>&1 | grep "^report" | sed "s/^/TSAMD_HOLBLOCK=0 (1 000 locations) /"
  TSAMD_LIB=$V/libtsamd_hhtime20.so python3 tools/validation_block.py 200000 20 2>&1 | grep "ts_hybhol n=" | tail -1
} > $O/validation_block.txt 2>&1
rm -f $V/*.so
{ for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 20 warmup 5:', d['value'], 'updates/s', d['ms_per_step'], 'ms/step')"; done
  python3 bench.py --gpus 1 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 2000 warmup 200:', d['value'], 'updates/s', d['ms_per_step'], 'ms/step')"; } > $O/short_vs_long.txt 2>&1
# `bench.py --gpus N` as the driver launches it, all ranks on device 0 (functional rehearsal; the rate means nothing)
{ echo "== rehearse_multi 8 1000000 8  (BASELINE config 4's 8-GPU form: ts_schedule<8,true,32> on every rank)"; timeout 900 bash tools/rehearse_multi.sh 8 1000000 8 2>&1 | grep "^\[bench\]\|^{\|exit code"
  echo "== rehearse_multi 8 250000 20 (config 5 on 8 ranks does not fit ONE device eight times over: the same kernel, ts_schedule<20,true,32>, on shards of 31 250)"; timeout 900 bash tools/rehearse_multi.sh 8 250000 20 2>&1 | grep "^\[bench\]\|^{\|exit code"
  echo "== rehearse_multi 4 1000000 8"; timeout 900 bash tools/rehearse_multi.sh 4 1000000 8 2>&1 | grep "^\[bench\]\|^{\|exit code"
} > $O/rehearsals.txt 2>&1
find gpurun_out -name "*.db" -delete  # (the summaries are what is kept; gpurun merges at most 64 MiB back)
tail -n 14 $O/*kernel_trace.txt $O/short_vs_long.txt $O/sched_timers.txt $O/hybrid_timers.txt $O/validation_block.txt | cut -c1-260
for f in $O/*pmc_*.txt; do echo "== $f"; grep -E "ts_pass|ts_resident|ts_schedule|ts_hybrid|ts_holblock|ts_hybhol" $f | cut -c1-170 | head -12; done
