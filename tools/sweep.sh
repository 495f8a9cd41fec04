cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for A in "--snps 4000 --steps 300 --warmup 50 --cpu-seconds 0" "--individuals 125000 --snps 20000 --steps 2000 --warmup 100 --cpu-seconds 0"; do
for cfg in "TSAMD_SWEEP=0" "TSAMD_SWEEP=1" "TSAMD_SWEEP=1 TSAMD_BLOCK=256" "TSAMD_SWEEP=1 TSAMD_GRID=512 TSAMD_BLOCK=256"; do
  echo "### $A $cfg"
  bash tools/prof.sh x $cfg -- $A 2>&1 | grep -E "ts_pass<8, false|^value" | cut -c1-130
done; done
