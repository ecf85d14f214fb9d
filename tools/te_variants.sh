#!/bin/bash
# Diagnostic builds of the tANS encoder (the call-to-call differences of round 3): each variant is the product
# library with mic_encode.hip compiled under another set of -D flags.  Built here or on the GPU box (hipcc cross-compiles):
#   tools/te_variants.sh build        -> build_variants/libmic_<name>.so (git-ignored, travels with gpurun)
#   tools/te_variants.sh run [flavours] [rounds]   (GPU box) -> gpurun_out/r4/te_variants.log
set -u
cd "$(dirname "$0")/.."
CS=medical-image-codec_amd/csrc
VAR=build_variants
names=(ship comb comb_inline comb_noinline comb_barvm comb_check comb_noipra comb_oldtr comb_oldtr_nohand comb_warm384 comb_blk1 comb_blk1_oldtr_nohand_warm)
flags=("" "-DTE_COMBINED" "-DTE_COMBINED -DTE_INLINE" "-DTE_COMBINED -DTE_NOINLINE" "-DTE_COMBINED -DTE_BAR_VM" "-DTE_COMBINED -DTE_CHECK" "-DTE_COMBINED -mllvm -enable-ipra=0" "-DTE_COMBINED -DTE_OLD_TRAILER" "-DTE_COMBINED -DTE_OLD_TRAILER -DTE_NO_HANDOFF" "-DTE_COMBINED -DTE_WARM_TOK=384" "-DTE_COMBINED -DTE_BLK1_64" "-DTE_COMBINED -DTE_BLK1_64 -DTE_OLD_TRAILER -DTE_NO_HANDOFF -DTE_WARM_TOK=384")
if [ "${1:-build}" = build ]; then
  bash $CS/build.sh > /dev/null || exit 1
  mkdir -p $VAR
  OBJS=""; for o in $CS/build/*.o; do case $o in *mic_encode.o) ;; *) OBJS="$OBJS $o";; esac; done
  for i in "${!names[@]}"; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${flags[$i]} -Rpass-analysis=kernel-resource-usage \
        -c $CS/mic_encode.hip -o $VAR/enc_${names[$i]}.o 2> $VAR/remarks_${names[$i]}.txt \
      && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $VAR/libmic_${names[$i]}.so $OBJS $VAR/enc_${names[$i]}.o -lpthread ) &
    if (( (i + 1) % 4 == 0 )); then wait; fi
  done
  wait
  ls -la $VAR/*.so
else
  shift
  mkdir -p gpurun_out/r4
  for n in "${names[@]}" old; do
    echo "== $n" >> gpurun_out/r4/te_variants.log
    MIC_HIP_LIB=$PWD/$VAR/libmic_$n.so python tools/dbg_fse1.py "$@" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4/te_variants.log
  done
  cat gpurun_out/r4/te_variants.log
fi
