cd $GRAFT_REPO_ROOT
A="--snps 4000 --steps 300 --warmup 50 --cpu-seconds 0 --no-profile"
V=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants
for rep in 1 2; do
echo "### default";  bash tools/prof.sh d TSAMD_X=1 -- $A 2>&1 | grep -E "ts_pass<|^value" | cut -c1-190
echo "### TSAMD_PF=1 (scalar column words)"; bash tools/prof.sh p TSAMD_PF=1 -- $A 2>&1 | grep -E "ts_pass<|^value" | cut -c1-190
echo "### TSAMD_PF=1 vector column words"; bash tools/prof.sh pv TSAMD_PF=1 TSAMD_LIB=$V/libtsamd_pfv.so -- $A 2>&1 | grep -E "ts_pass<|^value" | cut -c1-190
done
