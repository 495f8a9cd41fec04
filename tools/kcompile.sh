#!/bin/bash
# compile the ts_schedule unit (or UNIT=inst: the per-K unit) for the given K values in parallel and print every kernel's
# register / LDS / scratch use:  tools/kcompile.sh 12 16 20  [EXTRA="-DTSAMD_RES_ITEMS=6"]
UNIT=${UNIT:-sched}
OUT=${OUT:-/tmp/kb}
SRC=/root/repo/terastructure_amd/csrc/tsamd_$UNIT.hip
LICM=""; [ "$UNIT" != inst ] && LICM="-mllvm -disable-machine-licm"
for k in "$@"; do
  (mkdir -p $OUT/$UNIT$k && cd $OUT/$UNIT$k && /opt/rocm/bin/hipcc -c --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=16 \
     -I/root/repo/include -I/root/repo/terastructure_amd/csrc -DTSAMD_K=$k $LICM $EXTRA -o unit.o $SRC -save-temps=obj > log.txt 2>&1) &
done
wait
for k in "$@"; do
  echo "== K=$k"; grep -v "^$" $OUT/$UNIT$k/log.txt | head -${ERRLINES:-6}
  python3 /root/repo/tools/kres.py $OUT/$UNIT$k/tsamd_$UNIT-hip-amdgcn-amd-amdhsa-gfx950.s "${FILTER:-ts_}" 2>/dev/null
done
