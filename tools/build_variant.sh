#!/bin/bash
# Build another copy of the library from a source snapshot, for same-box A/B runs through AVAMD_LIB:
#   [EXTRA_FLAGS=-D...] tools/build_variant.sh <dir holding pkg/csrc/ and include/> <name>  ->  tools/_bin/libavhip_<name>.so
set -e
SRC=$1; NAME=$2
OUT=$(dirname "$0")/_bin; mkdir -p "$OUT/obj_$NAME"
for f in "$SRC"/pkg/csrc/*.hip; do
  b=$(basename "$f" .hip)
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-value $EXTRA_FLAGS -c "$f" -o "$OUT/obj_$NAME/$b.o" &
  while [ "$(jobs -r | wc -l)" -ge 6 ]; do sleep 0.2; done
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libavhip_$NAME.so" "$OUT/obj_$NAME"/*.o
rm -rf "$OUT/obj_$NAME"
echo "$OUT/libavhip_$NAME.so"
