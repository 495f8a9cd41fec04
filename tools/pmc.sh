# usage: pmc.sh <tag> <counter> -- <bench args>; one counter group per run, kernel-trace only
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; CTR=$2; shift; shift; shift
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc $CTR -d $OUT -o run -- python3 bench.py "$@" > $OUT/bench.log 2>&1
python3 tools/pmcstats.py "$OUT/*.db" | tee $OUT/pmcstats.txt
