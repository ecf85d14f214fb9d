#!/bin/bash
# A / B builds of the library: one source file compiled under other -D flags, the rest of the objects shared.
#   tools/variants.sh build <file.hip> name1="-DX=1" name2="-DX=2 -DY" ...   -> build_variants/libmic_<name>.so (travels with gpurun)
#   tools/variants.sh run "<command>" name1 name2 ...                         (GPU box) the command once per library, MIC_HIP_LIB set
set -u
cd "$(dirname "$0")/.."
CS=medical-image-codec_amd/csrc
VAR=build_variants
mode=$1; shift
if [ "$mode" = build ]; then
  src=$1; shift
  bash $CS/build.sh > /dev/null || exit 1
  mkdir -p $VAR
  obj=${src%.hip}.o
  OBJS=""; for o in $CS/build/*.o; do case $o in */$obj) ;; *) OBJS="$OBJS $o";; esac; done
  n=0
  for spec in "$@"; do
    name=${spec%%=*}; fl=${spec#*=}
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $fl -c $CS/$src -o $VAR/${name}_$obj \
      && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $VAR/libmic_$name.so $OBJS $VAR/${name}_$obj -lpthread ) &
    n=$((n + 1)); if (( n % 4 == 0 )); then wait; fi
  done
  wait
  ls -la $VAR/*.so
else
  cmd=$1; shift
  for name in "$@"; do
    echo "== $name"
    MIC_HIP_LIB=$PWD/$VAR/libmic_$name.so $cmd 2>&1 | grep -v amdgpu.ids
  done
fi
