# does the placement of kernel arguments (host vs device memory) matter for the ~4.6 us fixed cost per launch?
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  echo "### HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v python bench.py --individuals 125000 --snps 20000 --steps 3000 --warmup 100 --cpu-seconds 0 --no-profile 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=125K value',d['value'])"
  HIP_FORCE_DEV_KERNARG=$v python bench.py --steps 1500 --warmup 100 --snps 4000 --cpu-seconds 0 --no-profile 2>&1 | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=1M  value',d['value'])"
done
