cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c5; mkdir -p $O
timeout 3300 python3 -m pytest tests -x -q -m gpu --durations=8 > $O/t_all.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $O/summary.txt
tail -14 $O/t_all.log
bash tools/profile_round3.sh > $O/profile.log 2>&1; echo "profile rc=$?" | tee -a $O/summary.txt
tail -60 $O/profile.log
