#!/bin/bash
# Builds libmic_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libmic_hip.so
SRCS="mic_api.hip mic_api_ext.hip mic_pica.hip mic_encode.hip mic_decode.hip mic_decode_ls.hip mic_decode_px.hip mic_tables.hip mic_wavelet.hip mic_temporal.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fgpu-rdc-off -Wall -Wno-unused-function"
# -fgpu-rdc-off is not a real flag on every hipcc; fall back silently
if ! $HIPCC --offload-arch=gfx950 -fPIC -shared -x hip /dev/null -o /dev/null -fgpu-rdc-off 2>/dev/null; then
  FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function"
fi
$HIPCC $FLAGS ${EXTRA_FLAGS:-} -o $OUT $SRCS
echo "built $(realpath $OUT)"
