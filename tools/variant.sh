#!/bin/bash
# Builds a variant of libtsamd.so whose K-specialised kernels for one K are compiled with extra
# flags (ablations / tuning experiments):  tools/variant.sh <name> <K> <hipcc flags...>
# -> terastructure_amd/lib/variants/libtsamd_<name>.so ; select it with TSAMD_LIB=<path>.
# UNIT=sched builds the variant of the whole-schedule kernel's unit (csrc/tsamd_sched.hip) instead.
set -e
cd "$(dirname "$0")/.."
NAME=$1; K=$2; shift 2
python -m terastructure_amd.build >/dev/null
D=terastructure_amd/lib/variants; mkdir -p $D
U=${UNIT:-inst}; X=""; [ "$U" = sched ] && X="-mllvm -disable-machine-licm"
hipcc -c --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=16 $X -Iinclude -Iterastructure_amd/csrc \
  -DTSAMD_K=$K "$@" -o $D/inst_k${K}_$NAME.o terastructure_amd/csrc/tsamd_$U.hip
OBJS=$(ls terastructure_amd/lib/obj/*.o | grep -v "${U}_k${K}\.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libtsamd_$NAME.so $OBJS $D/inst_k${K}_$NAME.o -ldl
rm -f $D/inst_k${K}_$NAME.o
echo $D/libtsamd_$NAME.so
