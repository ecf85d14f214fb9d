#!/bin/bash
# ablation builds of the tokeniser on the GPU box (timing only: the output of a TK_ABL build is wrong on purpose)
# usage: FLAGSETS="-DTK_ABL=0 -DTK_WPS=4 ..." tools/abl_tok.sh
mkdir -p gpurun_out/r3
for a in ${FLAGSETS:--DTK_ABL=0 -DTK_ABL=2 -DTK_ABL=4}; do
  EXTRA_FLAGS=$a bash medical-image-codec_amd/csrc/build.sh > /dev/null 2>&1
  echo "$a" >> gpurun_out/r3/abl_tok.log
  python tools/abl_tokens.py 2>/dev/null | tail -1 >> gpurun_out/r3/abl_tok.log
done
cat gpurun_out/r3/abl_tok.log
