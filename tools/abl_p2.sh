#!/bin/bash
# ablation builds of k_dec_predict2 on the GPU box: which side of the two-wave pipeline bounds a step
mkdir -p gpurun_out/r3
for a in 0 1 2 4 6 7; do
  EXTRA_FLAGS=-DP2_ABL=$a bash medical-image-codec_amd/csrc/build.sh > /dev/null 2>&1
  echo "P2_ABL=$a" >> gpurun_out/r3/abl_p2.log
  python tools/time_dec.py 288 2 2>/dev/null | grep "k_dec" >> gpurun_out/r3/abl_p2.log
done
cat gpurun_out/r3/abl_p2.log
