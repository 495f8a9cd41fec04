cd $GRAFT_REPO_ROOT
for w in 1 2 4; do python tools/local_shards_bench.py $w 1000000 2>&1 | grep world; done
for w in 1 2 4; do python tools/local_shards_bench.py $w 125000 2>&1 | grep world; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ls; mkdir -p gpurun_out/prof_ls
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ls -o run -- python3 tools/local_shards_bench.py 2 250000 > gpurun_out/prof_ls/log.txt 2>&1
python3 tools/kstats.py "gpurun_out/prof_ls/*.db" | grep -E "ts_pass|kernel"
