#!/bin/bash
# Builds libmic_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
# One object per source, compiled side by side and only when the source (or a header) is newer; then one link.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libmic_hip.so
OBJ=build
SRCS="mic_api.hip mic_host_io.hip mic_api_ext.hip mic_pica.hip mic_encode.hip mic_decode.hip mic_decode_ls.hip mic_decode_px.hip mic_decode_rows.hip mic_decode_fused.hip mic_tables.hip mic_wavelet.hip mic_temporal.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${EXTRA_FLAGS:-}"
mkdir -p $OBJ
# a change of flags rebuilds everything
echo "$FLAGS" > $OBJ/flags.new
if ! cmp -s $OBJ/flags.new $OBJ/flags 2>/dev/null; then rm -f $OBJ/*.o; mv $OBJ/flags.new $OBJ/flags; else rm -f $OBJ/flags.new; fi
newest_hdr=$(ls -t *.h ../../include/*.h | head -1)
pids=()
for s in $SRCS; do
  o=$OBJ/${s%.hip}.o
  if [ ! -f $o ] || [ $s -nt $o ] || [ $newest_hdr -nt $o ]; then
    ( $HIPCC $FLAGS -c $s -o $o.tmp && mv $o.tmp $o ) &
    pids+=($!)
  fi
done
rc=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait $p || rc=1; }; done
[ $rc -eq 0 ] || { echo "compile failed" >&2; exit 1; }
OBJS=""
for s in $SRCS; do OBJS="$OBJS $OBJ/${s%.hip}.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT.tmp $OBJS -lpthread
mv $OUT.tmp $OUT
echo "built $(realpath $OUT)"
