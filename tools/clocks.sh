# engine clock while the default workload runs (sampled from sysfs / rocm-smi)
cd $GRAFT_REPO_ROOT
python bench.py --steps 150000 --warmup 200 --cpu-seconds 0 --no-profile > gpurun_out/clk_bench.log 2>&1 &
BP=$!
sleep 14
for i in 1 2 3 4 5 6; do
  cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | grep '\*' | head -2
  rocm-smi --showclocks 2>/dev/null | grep -iE "sclk|mclk" | head -3
  sleep 1
done
wait $BP
grep '^{' gpurun_out/clk_bench.log | cut -c1-160
