# Round 6: what the driver runs at round end -- smoke, the GPU suite (with its slowest tests), bench.py with the driver's arguments and the defaults
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(timeout 1300 python3 -m pytest tests -m gpu -q --durations=45 > $O/suite.log 2>&1; echo "exit $?" >> $O/suite.log)
grep -E "passed|failed|exit" $O/suite.log | tail -3
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python3 bench.py > $O/bench_default_args.json 2> $O/bench_default_args.err
python3 - <<'PY'
import json
for f in ("bench_driver_args", "bench_default_args"):
    for ln in open(f"gpurun_out/r06/{f}.json"):
        if ln.startswith("{"):
            d = json.loads(ln)
            print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["legs"])
PY
