# Shows that tests/test_gpu_multirank.py::test_p2p_early_convergence_with_a_lagging_rank sees the slot-reuse hazard of
# the peer-to-peer exchange: the same test with the guard disabled (TSAMD_TEST_XCHG_NOGUARD=1) is expected to FAIL.
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  TSAMD_TEST_XCHG_NOGUARD=1 python3 -m pytest -x -q tests/test_gpu_multirank.py -m gpu -k "early_convergence or small_pass_caps" 2>&1 | grep -E "passed|failed|assert|Error" | head -4
done
