cd $GRAFT_REPO_ROOT
A="--steps 60 --warmup 10 --cpu-seconds 0 --no-profile --snps 20000"
for sw in 0 1; do
  export TSAMD_SWEEP=$sw
  echo "### TSAMD_SWEEP=$sw"
  bash tools/pmc.sh l2_$sw "TCC_HIT_sum TCC_MISS_sum" -- $A 2>&1 | grep -E "ts_pass<8, false" | cut -c1-200
done
