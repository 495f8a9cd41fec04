# Round 6, last call, on the FINAL sources: the round's profiles (tools/r06/profile.sh -> gpurun_out/prof_r06/), then what the driver runs at round
# end (tools/r06/suite.sh: smoke, the GPU suite, bench.py with the driver's arguments and the defaults)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/r06/profile.sh > gpurun_out/prof_r06.log 2>&1
tail -5 gpurun_out/prof_r06/short_vs_long.txt gpurun_out/prof_r06/sched_timers.txt | cut -c1-300
bash tools/r06/suite.sh
