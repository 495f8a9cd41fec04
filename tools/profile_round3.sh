# Round-3 profile (run on the GPU box): kernel-trace summaries of the default bench (K = 8, N = 1M: one ts_schedule launch per
# schedule), of K = 16 / N = 500K and K = 20 / N = 125K (BASELINE config 5's 8-GPU shard on one GPU) in the same mode, and of
# K = 20 / N = 1M (config 5 on ONE GPU: the weights do not fit the register file, one launch per pass); HBM traffic, fp64
# instruction and SQ counters in separate --pmc runs; in-kernel timers of the diagnostic build; short-vs-long bench comparison.
# Writes under gpurun_out/prof_r03/ ; copy what is to be judged into profiles/ (tools/pmc_record.py reads it from there).
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03; mkdir -p $O
K16="--pops 16 --individuals 500000 --snps 200000"
K20S="--pops 20 --individuals 125000 --snps 200000"
K20="--pops 20 --snps 200000 --steps 300 --warmup 50"
bash tools/prof.sh default -- > $O/k8_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_default/bench.log > $O/k8_bench_under_rocprof.json
bash tools/prof.sh k16 -- $K16 > $O/k16_n500k_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_k16/bench.log > $O/k16_n500k_bench_under_rocprof.json
bash tools/prof.sh k20s -- $K20S > $O/k20_n125k_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_k20s/bench.log > $O/k20_n125k_bench_under_rocprof.json
bash tools/prof.sh k20 -- $K20 --cpu-seconds 0 > $O/k20_n1m_kernel_trace.txt 2>&1
grep '^{' gpurun_out/prof_k20/bench.log > $O/k20_n1m_bench_under_rocprof.json
# counters: the largest launch of a run is the 200-update ts_schedule launch (per update = max / 200)
A="--steps 200 --warmup 10 --ramp-seconds 0 --cpu-seconds 0 --no-profile --l 20000"
F64="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQC_ICACHE_REQ SQC_ICACHE_MISSES"
bash tools/pmc.sh fetch FETCH_SIZE -- $A > $O/k8_pmc_fetch_size.txt 2>&1
bash tools/pmc.sh write WRITE_SIZE -- $A > $O/k8_pmc_write_size.txt 2>&1
bash tools/pmc.sh f64 "$F64" -- $A > $O/k8_pmc_f64.txt 2>&1
bash tools/pmc.sh sq1 "$SQ1" -- $A > $O/k8_pmc_sq1.txt 2>&1
bash tools/pmc.sh sq2 "$SQ2" -- $A > $O/k8_pmc_sq2.txt 2>&1
bash tools/pmc.sh fetch16 FETCH_SIZE -- $A $K16 > $O/k16_n500k_pmc_fetch_size.txt 2>&1
bash tools/pmc.sh write16 WRITE_SIZE -- $A $K16 > $O/k16_n500k_pmc_write_size.txt 2>&1
bash tools/pmc.sh f6416 "$F64" -- $A $K16 > $O/k16_n500k_pmc_f64.txt 2>&1
bash tools/pmc.sh fetch20s FETCH_SIZE -- $A $K20S > $O/k20_n125k_pmc_fetch_size.txt 2>&1
bash tools/pmc.sh write20s WRITE_SIZE -- $A $K20S > $O/k20_n125k_pmc_write_size.txt 2>&1
bash tools/pmc.sh f6420s "$F64" -- $A $K20S > $O/k20_n125k_pmc_f64.txt 2>&1
bash tools/pmc.sh sq120s "$SQ1" -- $A $K20S > $O/k20_n125k_pmc_sq1.txt 2>&1
B="--steps 60 --warmup 10 --ramp-seconds 0 --cpu-seconds 0 --no-profile --l 20000"
TSAMD_PERSISTENT=0 bash tools/pmc.sh fetchps FETCH_SIZE -- $B > $O/k8_per_snp_pmc_fetch_size.txt 2>&1
TSAMD_PERSISTENT=0 bash tools/pmc.sh writeps WRITE_SIZE -- $B > $O/k8_per_snp_pmc_write_size.txt 2>&1
bash tools/pmc.sh fetch20 FETCH_SIZE -- $B --pops 20 > $O/k20_n1m_pmc_fetch_size.txt 2>&1
bash tools/pmc.sh write20 WRITE_SIZE -- $B --pops 20 > $O/k20_n1m_pmc_write_size.txt 2>&1
# in-kernel timers (diagnostic build of the ts_schedule unit: UNIT=sched tools/variant.sh time<K> <K> -DTSAMD_SCHED_TIME)
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
T="--steps 2000 --warmup 200 --cpu-seconds 0 --no-profile"
{ [ -f $V/libtsamd_time8.so ] && TSAMD_LIB=$V/libtsamd_time8.so python3 bench.py $T 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | sed "s/^/N=1000000 K=8: /"
  [ -f $V/libtsamd_time16.so ] && TSAMD_LIB=$V/libtsamd_time16.so python3 bench.py $T $K16 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | sed "s/^/N=500000 K=16: /"
  [ -f $V/libtsamd_time20.so ] && TSAMD_LIB=$V/libtsamd_time20.so python3 bench.py $T $K20S 2>/dev/null | grep "ts_schedule n=2000" | tail -1 | sed "s/^/N=125000 K=20: /"
} > $O/sched_timers.txt 2>&1
{ for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 20 warmup 5:', d['value'], 'updates/s', d['ms_per_step'], 'ms/step')"; done
  python3 bench.py --gpus 1 --steps 2000 --warmup 200 --cpu-seconds 0 --no-profile 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps 2000 warmup 200:', d['value'], 'updates/s', d['ms_per_step'], 'ms/step')"; } > $O/short_vs_long.txt 2>&1
find gpurun_out -name "*.db" -delete  # (the summaries are what is kept; gpurun merges at most 64 MiB back)
tail -n 14 $O/*kernel_trace.txt $O/short_vs_long.txt $O/sched_timers.txt | cut -c1-220
for f in $O/*pmc_*.txt; do echo "== $f"; grep -E "ts_pass|ts_resident|ts_schedule" $f | cut -c1-170 | head -12; done
