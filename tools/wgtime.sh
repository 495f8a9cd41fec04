# start / finish times of the workgroups of one first pass and one plain pass (variant built with -DTSAMD_WGTIME)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TSAMD_LIB=$GRAFT_REPO_ROOT/terastructure_amd/lib/variants/libtsamd_${1:-wgt}.so python3 bench.py --snps 2000 --steps 60 --warmup 0 --cpu-seconds 0 --no-profile 2>&1 | grep -E "^wg(time|plain)" > gpurun_out/wgtime_${1:-wgt}.txt
python3 tools/wgtime.py gpurun_out/wgtime_${1:-wgt}.txt
