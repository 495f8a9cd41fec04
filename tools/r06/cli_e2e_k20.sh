# Round 6: the drop-in binary at BASELINE config 5's shape on ONE GPU: -n 1000000 -k 20 (ts_hybrid: half the weights streamed), -l 40000 (10 GB .bed),
# -rfreq 10000, three report periods; gamma.txt / theta.txt are 2 x 20M values per save.  timing.txt -> gpurun_out/r06/cli_k20_*
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
W=${TMPDIR:-/tmp}/ts_e2e; mkdir -p $W
name=k20; n=1000000; l=40000; k=20
python3 tools/make_synth_bed.py $W/$name $n $l $k 2>&1 | tail -1
t0=$(date +%s.%N)
(cd $W && $GRAFT_REPO_ROOT/host/terastructure -file $name.bed -n $n -l $l -k $k -stochastic -nthreads 1 -label $name -rfreq 10000 -max-iter 30000 > $O/cli_$name.stdout 2> $O/cli_$name.stderr)
echo "exit $? wall $(python3 -c "import time; print(round(time.time() - $t0, 2))") s" >> $O/cli_$name.stdout
d=$(ls -d $W/n$n-k$k-l$l-$name* | head -1)
cp $d/timing.txt $O/cli_${name}_timing.txt; cp $d/validation.txt $O/cli_${name}_validation.txt; ls -la $d > $O/cli_${name}_files.txt
tail -1 $O/cli_$name.stdout; cat $O/cli_${name}_timing.txt; cat $O/cli_${name}_validation.txt
rm -rf $W
