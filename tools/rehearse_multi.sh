# Functional rehearsal of `bench.py --gpus N` on a box with ONE GPU: N ranks launched exactly as the
# driver launches them (torch.distributed.run, one process per rank), all on device 0
# (TSAMD_BENCH_DEVICE=0), small shards.  Exercises the torchrun launch, the exchange self-test
# against the oracle, the JSON line and the failure agreement -- NOT a performance proxy: the
# ranks time-share one GPU and the xGMI hop does not exist here.
#   usage: bash tools/rehearse_multi.sh [N=8] [individuals=80000] [K=8]
# BASELINE config 5's geometry (K = 20, 8 shards of 125 000) does not fit ONE device eight times over (8 x 245 resident
# workgroups); `rehearse_multi.sh 8 250000 20` runs the same code path -- ts_schedule<20> with its level 2 across 8
# ranks, 31 workgroups per rank -- on shards of 31 250.
cd $GRAFT_REPO_ROOT
N=${1:-8}; NI=${2:-80000}; K=${3:-8}
export TSAMD_BENCH_DEVICE=0 GPU_MAX_HW_QUEUES=$((N>4?N:4)) HSA_ENABLE_IPC_MODE_LEGACY=0
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus $N --individuals $NI --pops $K --snps 2000 --steps 40 --warmup 10
echo "exit code $?"
