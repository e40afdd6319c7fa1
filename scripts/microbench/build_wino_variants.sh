#!/bin/bash
# Variant builds of libmp_engine.so that differ ONLY in conv_wino_bf16.o (the other objects are the product build's), for the round-6
# root-cause experiment on the bf16x9 Winograd kernel and for tests/test_gpu_wino_permuted.py:
#   _build/wperm3, wperm8   product source + -DMP_WINO_PERMUTE=3 / 8 (deliberately different vector-register assignment in the K loop)
#   _build/pk1, phases, exp      (argument `ab`) A/B / profiling builds of round 6
#   _build/r5bad            the archived round-5 "L2 prefetch" source exactly as it failed (profiles/r05_wino_prefetch_ab.hip.txt)
#   _build/r5badfix         the same source + the two wait states in front of the accumulator-reset MFMAs (the round-6 fix), nothing else
# Usage: bash scripts/microbench/build_wino_variants.sh   (after make -C megapose6d_amd/csrc)
set -e
MODE="$1"
cd "$(dirname "$0")/../.."
C=megapose6d_amd/csrc
B=scripts/microbench/_build
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Iinclude -Wall -Wno-unused-function"
OTHERS=$(ls $C/*.o | grep -v conv_wino_bf16.o)
link() { mkdir -p $B/$1; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/$1/libmp_engine.so $OTHERS $B/$1/conv_wino_bf16.o; }
for n in 3 8; do
  mkdir -p $B/wperm$n
  /opt/rocm/bin/hipcc $F -DMP_WINO_PERMUTE=$n -c $C/conv_wino_bf16.hip -o $B/wperm$n/conv_wino_bf16.o
  link wperm$n
done
if [ "$MODE" = "ab" ]; then   # A/B and profiling builds of the round: phases = cycle stamps in prologue / epilogue, exp = the DIAG instances
  for v in "phases -DMP_WINO_PHASES" "exp -DMP_CONV_EXPERIMENTS" $EXTRA_VARIANTS; do
    set -- $v; n=$1; shift
    mkdir -p $B/$n
    /opt/rocm/bin/hipcc $F "$@" -c $C/conv_wino_bf16.hip -o $B/$n/conv_wino_bf16.o
    link $n
  done
fi
if [ "$MODE" = "ab" ]; then   # base = the kernel at git revision $BASE_REV (default HEAD = before the edit being measured): same-box A/B
  REV=${BASE_REV:-HEAD}
  mkdir -p $B/base/src
  cp $C/*.h $B/base/src/
  for f in conv_wino_bf16.hip wino_common.h; do git show $REV:megapose6d_amd/csrc/$f > $B/base/src/$f; done
  git show $REV:megapose6d_amd/csrc/conv_wino_bf16_sched.h > $B/base/src/conv_wino_bf16_sched.h 2>/dev/null || rm -f $B/base/src/conv_wino_bf16_sched.h
  /opt/rocm/bin/hipcc $F -DMP_WINO_PK=0 -c $B/base/src/conv_wino_bf16.hip -o $B/base/conv_wino_bf16.o
  link base
fi
if [ "$MODE" = "r5" ]; then
  for v in r5bad r5badfix; do
    mkdir -p $B/$v/src
    cp $C/*.h $B/$v/src/
    sed -i 's/int telemetry; /int telemetry; int prefetch_stride; /' $B/$v/src/wino_common.h
    cp profiles/r05_wino_prefetch_ab.hip.txt $B/$v/src/conv_wino_bf16.hip
    if [ $v = r5badfix ]; then
      sed -i 's/asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, 0"/asm volatile("s_nop 1\\n\\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, 0"/' $B/$v/src/conv_wino_bf16.hip
      grep -c 's_nop 1\\n\\tv_mfma' $B/$v/src/conv_wino_bf16.hip
    fi
    /opt/rocm/bin/hipcc $F -c $B/$v/src/conv_wino_bf16.hip -o $B/$v/conv_wino_bf16.o
    link $v
  done
fi
ls -la $B/*/libmp_engine.so
