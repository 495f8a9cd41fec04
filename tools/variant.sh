#!/bin/bash
# Builds a variant of libtsamd.so whose K-specialised kernels for one K are compiled with extra
# flags (ablations / tuning experiments):  tools/variant.sh <name> <K> <hipcc flags...>
# -> terastructure_amd/lib/variants/libtsamd_<name>.so ; select it with TSAMD_LIB=<path>.
# UNIT=sched builds the variant of the whole-schedule kernel's unit (csrc/tsamd_sched.hip) instead; UNIT=hol / UNIT=hyb that
# of ts_holblock / ts_hybrid (-DTSAMD_SCHED_TIME: their in-kernel timers);
# UNIT=all recompiles the per-K unit, the whole-schedule unit AND the host (csrc/tsamd.hip) with the flags
# (variants that change the resident kernels' geometry: -DTSAMD_RES_VEC / -DTSAMD_RES_ITEMS).
set -e
cd "$(dirname "$0")/.."
NAME=$1; K=$2; shift 2
python -m terastructure_amd.build >/dev/null 2>&1
D=terastructure_amd/lib/variants; mkdir -p $D
U=${UNIT:-inst}
CC="hipcc -c --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=16 -Iinclude -Iterastructure_amd/csrc"
NEW=""; SKIP="__none__"
if [ "$U" = inst ] || [ "$U" = all ]; then
  $CC -DTSAMD_K=$K "$@" -o $D/v_inst_k${K}_$NAME.o terastructure_amd/csrc/tsamd_inst.hip &
  NEW="$NEW $D/v_inst_k${K}_$NAME.o"; SKIP="$SKIP|/inst_k${K}\.o"
fi
if [ "$U" = sched ] || [ "$U" = all ]; then
  $CC -mllvm -disable-machine-licm -DTSAMD_K=$K "$@" -o $D/v_sched_k${K}_$NAME.o terastructure_amd/csrc/tsamd_sched.hip &
  NEW="$NEW $D/v_sched_k${K}_$NAME.o"; SKIP="$SKIP|/sched_k${K}\.o"
fi
if [ "$U" = hol ] || [ "$U" = hyb ] || [ "$U" = hhol ]; then   # the batched validation kernel / the above-capacity kernel (csrc/tsamd_hol.hip, tsamd_hyb.hip)
  $CC -mllvm -disable-machine-licm -DTSAMD_K=$K "$@" -o $D/v_${U}_k${K}_$NAME.o terastructure_amd/csrc/tsamd_$U.hip &
  NEW="$NEW $D/v_${U}_k${K}_$NAME.o"; SKIP="$SKIP|/${U}_k${K}\.o"
fi
if [ "$U" = all ]; then
  $CC "$@" -o $D/v_main_$NAME.o terastructure_amd/csrc/tsamd.hip &
  NEW="$NEW $D/v_main_$NAME.o"; SKIP="$SKIP|/tsamd\.o"
fi
wait
OBJS=$(ls terastructure_amd/lib/obj/*.o | grep -Ev "$SKIP")
hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libtsamd_$NAME.so $OBJS $NEW -ldl
rm -f $NEW
echo $D/libtsamd_$NAME.so
