cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for A in "--individuals 125000 --snps 20000 --steps 2000 --warmup 100 --cpu-seconds 0" "--snps 4000 --steps 300 --warmup 50 --cpu-seconds 0"; do
  echo "### $A"
  bash tools/prof.sh x -- $A 2>&1 | grep -E "ts_pass|^value" | cut -c1-130
done
