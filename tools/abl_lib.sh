#!/bin/bash
# builds the library with each flag set in turn and runs a diagnostic on it (GPU box)
# usage: FLAGSETS="-DX=1 -DX=2" CMD="python tools/dbg_fse1.py" tools/abl_lib.sh
mkdir -p gpurun_out/r3
for a in $FLAGSETS; do
  EXTRA_FLAGS=$a bash medical-image-codec_amd/csrc/build.sh > /dev/null 2>&1
  echo "== $a" >> gpurun_out/r3/abl_lib.log
  $CMD 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3/abl_lib.log
done
cat gpurun_out/r3/abl_lib.log
