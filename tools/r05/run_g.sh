#!/bin/bash
# round 5, last GPU call: what the driver runs at round end (GPU suite with durations, smoke, the bench line with the driver's
# arguments and the default ones) and the other BASELINE configurations through bench.py (tools/configs.sh)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q --durations=25 > $O/g_suite.log 2>&1
tail -4 $O/g_suite.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/g_smoke.log 2>&1; tail -1 $O/g_smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/g_bench_driver.json 2> $O/g_bench_driver.err
python3 bench.py > $O/g_bench_default.json 2> $O/g_bench_default.err
python3 - <<'PY'
import json
for f in ("g_bench_driver.json", "g_bench_default.json"):
    d = json.loads([ln for ln in open("gpurun_out/r05/" + f) if ln.startswith("{")][0])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["legs"], d["validation_block"]["seconds_per_report"], d["parity_vs_cpu_baseline"]["ok"])
PY
bash tools/configs.sh > $O/g_other_configs.txt 2>&1
python3 - <<'PY'
import json
for ln in open("gpurun_out/r05/g_other_configs.txt"):
    if ln.startswith("{"):
        d = json.loads(ln); vb = d.get("validation_block") or {}
        print("  ", d["value"], "updates/s", round(1e3 * d["ms_per_step"], 2), "us; roofline", d["roofline"] and (d["roofline"]["bound"], d["roofline"]["frac"]),
              "; validation", vb.get("us_per_location"), vb.get("entry_by_entry_us_per_location"), "; cpu", d["cpu_baseline"] and (d["cpu_baseline"]["value"], d["cpu_baseline"]["value_1_thread"]),
              "; parity", d["parity_vs_cpu_baseline"] and d["parity_vs_cpu_baseline"]["ok"])
    else:
        print(ln.strip()[:200])
PY
