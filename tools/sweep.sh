cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
A="--l 4000 --steps 300 --warmup 50 --cpu-seconds 0"
for cfg in "TSAMD_BLOCK=256 TSAMD_GRID=256" "TSAMD_BLOCK=512 TSAMD_GRID=256" "TSAMD_GRID_FIRST=1024" "TSAMD_GRID_FIRST=768"; do
  echo "### $cfg"
  bash tools/prof.sh x $cfg -- $A 2>&1 | grep -E "ts_pass|^value" | cut -c1-230
done
