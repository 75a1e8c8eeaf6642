#!/bin/bash
# Build libqea_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libqea_hip.so
mkdir -p obj
pids=()
for f in *.hip; do
  o=obj/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || [ ../../include/qea_hip.h -nt "$o" ]; then
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" obj/*.o
echo "built $OUT"
