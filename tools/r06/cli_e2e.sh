# Round 6: the drop-in binary end to end at BASELINE scale: host/terastructure on a synthetic PSD .bed of
#   (a) config 3 at its full size: N = 100 000, L = 500 000, K = 8 (12.5 GB .bed), -rfreq 100000, three report periods;
#   (b) config 4's individuals: N = 1 000 000, K = 8, L = 40 000 (10 GB .bed: the scratch disk, not the engine, bounds L),
#       -rfreq 100000, two report periods -- where a report's gamma.txt / theta.txt are 2 x 8M values.
# timing.txt of each run (ingest, validation sample, training, reports, save_model blocked / overlapped) -> gpurun_out/r06/cli_*
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
W=${TMPDIR:-/tmp}/ts_e2e; mkdir -p $W; df -h $W | tail -1
run() {  # name n l k extra...
  name=$1; n=$2; l=$3; k=$4; shift 4
  python3 tools/make_synth_bed.py $W/$name $n $l $k 2>&1 | tail -2
  t0=$(date +%s.%N)
  (cd $W && $GRAFT_REPO_ROOT/host/terastructure -file $name.bed -n $n -l $l -k $k -stochastic -nthreads 1 -label $name "$@" > $O/cli_$name.stdout 2> $O/cli_$name.stderr)
  echo "exit $? wall $(python3 -c "import time; print(round(time.time() - $t0, 2))") s" >> $O/cli_$name.stdout
  d=$(ls -d $W/n$n-k$k-l$l-$name* | head -1)
  cp $d/timing.txt $O/cli_${name}_timing.txt; cp $d/validation.txt $O/cli_${name}_validation.txt; cp $d/param.txt $O/cli_${name}_param.txt
  ls -la $d > $O/cli_${name}_files.txt
  head -c 400 $d/theta.txt > $O/cli_${name}_theta_head.txt
  tail -1 $O/cli_$name.stdout
  cat $O/cli_${name}_timing.txt
  rm -rf $W/$name.bed $W/$name.bim $W/$name.fam $d
}
run cfg3 100000 500000 8 -rfreq 100000 -max-iter 300000
run n1m 1000000 40000 8 -rfreq 100000 -max-iter 200000
