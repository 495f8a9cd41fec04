# round profile: kernel-trace summary of the default bench command + HBM traffic counters
cd $GRAFT_REPO_ROOT
bash tools/prof.sh default -- > gpurun_out/prof_default_summary.txt 2>&1
cat gpurun_out/prof_default_summary.txt | cut -c1-200
A="--steps 60 --warmup 10 --cpu-seconds 0 --no-profile --l 20000"
bash tools/pmc.sh fetch FETCH_SIZE -- $A 2>&1 | cut -c1-200
bash tools/pmc.sh write WRITE_SIZE -- $A 2>&1 | cut -c1-200
