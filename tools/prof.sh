# usage: prof.sh <tag> [ENV=VAL ...] -- <bench args>; rocprofv3 kernel-trace summary (no PMC)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
while [ "$1" != "--" ] && [ -n "$1" ]; do export "$1"; shift; done
shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o run -- python3 bench.py "$@" > $OUT/bench.log 2>&1
python3 tools/kstats.py "$OUT/*.db" | tee $OUT/kstats.txt
grep '^{' $OUT/bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value',d['value'],'roofline',d['roofline'])"
