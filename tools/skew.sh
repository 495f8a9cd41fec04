# first-pass uneven split sweep (TSAMD_FIRST_SKEW="even1,odd1,even2,odd2"): rocprofv3 kernel times + workgroup finish times
cd $GRAFT_REPO_ROOT
for sk in "$@"; do
  echo "### TSAMD_FIRST_SKEW=$sk"
  bash tools/prof.sh sk TSAMD_FIRST_SKEW=$sk -- --snps 4000 --steps 300 --warmup 50 --cpu-seconds 0 --no-profile 2>&1 | grep -E "ts_pass<8, true|^value" | cut -c1-190
  TSAMD_FIRST_SKEW=$sk bash tools/wgtime.sh wgt 2>&1 | grep -A16 "== wgtime" | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$13,$14,$15,$16,$17}' | head -17
done
