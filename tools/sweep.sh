# launch-geometry sweep of the plain pass at mid-size shards (run on the GPU box)
cd $GRAFT_REPO_ROOT
for A in "--individuals 125000 --snps 20000 --pops 8 --steps 2000 --warmup 100 --cpu-seconds 0" "--individuals 10000 --snps 20000 --pops 6 --steps 3000 --warmup 100 --cpu-seconds 0"; do
for cfg in "TSAMD_GRID=256 TSAMD_BLOCK=256" "TSAMD_GRID=128 TSAMD_BLOCK=256" "TSAMD_GRID=64 TSAMD_BLOCK=256" "TSAMD_GRID=128 TSAMD_BLOCK=512" "TSAMD_GRID=64 TSAMD_BLOCK=512" "TSAMD_GRID=128 TSAMD_BLOCK=256 TSAMD_GRID_FIRST=256" "TSAMD_GRID=128 TSAMD_BLOCK=256 TSAMD_GRID_FIRST=128"; do
  echo "### $A $cfg"
  bash tools/prof.sh x $cfg -- $A 2>&1 | grep -E "ts_pass<|^value" | cut -c1-130
done; done
