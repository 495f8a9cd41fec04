# per-kernel times at small / mid / full shard sizes (run on the GPU box)
cd $GRAFT_REPO_ROOT
for A in "--individuals 10000 --snps 20000 --pops 6 --steps 3000 --warmup 100" "--individuals 125000 --snps 20000 --pops 8 --steps 2000 --warmup 100" "--snps 4000 --steps 300 --warmup 50"; do
  echo "### $A"
  bash tools/prof.sh x -- $A --cpu-seconds 0 2>&1 | grep -E "ts_pass<" | cut -c1-130
  python bench.py $A --cpu-seconds 0 --no-profile 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value',d['value'])"
done
