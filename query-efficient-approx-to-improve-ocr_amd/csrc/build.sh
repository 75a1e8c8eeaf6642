#!/bin/bash
# Build libqea_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# Every translation unit is compiled with -Rpass-analysis=kernel-resource-usage; the remarks are kept in obj/<file>.res and
# resource_report.py fails the build when a kernel spills more than QEA_MAX_SCRATCH bytes per lane (default 512: a register
# spill inside an MFMA loop of that size is a scheduling accident, e.g. partial tiles kept alive across a whole K chunk —
# seen once at 8 KB per lane; the few dozen bytes some tuned kernels carry are listed, not refused).
set -e
cd "$(dirname "$0")"
OUT=../libqea_hip.so
mkdir -p obj
pids=()
for f in *.hip; do
  o=obj/${f%.hip}.o
  if [ ! -f "$o" ] || [ ! -f "${o%.o}.res" ] || [ "$f" -nt "$o" ] || [ -n "$(find . -maxdepth 1 -name '*.h' -newer "$o")" ] || [ ../../include/qea_hip.h -nt "$o" ]; then
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Rpass-analysis=kernel-resource-usage $QEA_EXTRA_HIPCC_FLAGS \
      -c "$f" -o "$o" 2> "${o%.o}.res.tmp" && mv "${o%.o}.res.tmp" "${o%.o}.res" &
    pids+=($!)
  fi
done
fail=0
for p in "${pids[@]}"; do wait "$p" || fail=1; done
if [ $fail -ne 0 ]; then
  grep -h -B2 -A6 "error:" obj/*.res.tmp >&2 || cat obj/*.res.tmp >&2
  exit 1
fi
python3 resource_report.py obj/*.res
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" obj/*.o
echo "built $OUT"
