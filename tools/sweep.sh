cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
A="--l 4000 --steps 300 --warmup 50 --cpu-seconds 0"
for cfg in "TSAMD_FIRST_VEC=1 TSAMD_GRID_FIRST=256" "TSAMD_FIRST_VEC=1 TSAMD_GRID_FIRST=512" "TSAMD_FIRST_VEC=1 TSAMD_GRID_FIRST=1024" "TSAMD_FIRST_VEC=1 TSAMD_GRID_FIRST=2048" "TSAMD_FIRST_VEC=2 TSAMD_GRID_FIRST=512"; do
  echo "### $cfg"
  bash tools/prof.sh x $cfg -- $A 2>&1 | grep -E "ts_pass|ts_finish|^value" | cut -c1-160
done
