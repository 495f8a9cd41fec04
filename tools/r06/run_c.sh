# Round 6, call c (on the final kernel sources): the `slow` half of the suite once (every K in both whole-schedule families, the
# 40-case stress slice), the randomised parity stress by hand, the soak run, the other BASELINE configurations through bench.py.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
(TS_RUN_SLOW=1 timeout 1500 python3 -m pytest tests/test_gpu_holblock.py tests/test_gpu_hybrid.py tests/test_gpu_hybhol.py tests/test_gpu_geometry.py -q -m gpu --durations=5 \
   -k "every_instantiation or instantiations or slice or every_k" > $O/c_slow.log 2>&1; echo "exit $?" >> $O/c_slow.log)
tail -12 $O/c_slow.log
(timeout 1500 python3 tools/stress_parity.py 300 61 > $O/c_stress.log 2>&1; echo "exit $?" >> $O/c_stress.log); tail -4 $O/c_stress.log
(timeout 900 python3 tools/soak.py > $O/c_soak.log 2>&1; echo "exit $?" >> $O/c_soak.log); tail -6 $O/c_soak.log
bash tools/configs.sh > $O/other_configs.txt 2>&1
grep -E "^###|launch per pass" $O/other_configs.txt | head -40
python3 - <<'PY'
import json
for ln in open("gpurun_out/r06/other_configs.txt"):
    if ln.startswith("{"):
        d = json.loads(ln)
        vb = d.get("validation_block") or {}
        cb = d.get("cpu_baseline") or {}
        print(d["metric"][-22:], d["value"], round(1e3 * d["ms_per_step"], 2), "us", "| val us/loc", vb.get("us_per_location"), vb.get("entry_by_entry_us_per_location"),
              "| cpu", cb.get("value"), cb.get("value_1_thread"), "| legs ok" if all(v == "ok" for v in d["legs"].values()) else d["legs"])
PY
