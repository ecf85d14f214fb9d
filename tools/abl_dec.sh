#!/bin/bash
# builds of the library with each flag set in turn, decode kernel times of the bench batch (GPU box)
mkdir -p gpurun_out/r3
for a in $FLAGSETS; do
  EXTRA_FLAGS=$a bash medical-image-codec_amd/csrc/build.sh > /dev/null 2>&1
  echo "$a" >> gpurun_out/r3/abl_dec.log
  python tools/time_dec.py 288 2 2>/dev/null | grep -E "k_dec|equal" >> gpurun_out/r3/abl_dec.log
done
cat gpurun_out/r3/abl_dec.log
