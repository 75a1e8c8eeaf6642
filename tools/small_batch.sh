#!/bin/bash
# Developer tool (GPU box): the Phase-B step at the reference's real batch sizes, eager and as a hipGraph replay, in the fp16 and
# the bf16 split forms on the same box.  -> gpurun_out/small_batch.txt
set -e
mkdir -p gpurun_out
out=gpurun_out/small_batch.txt
: > $out
for b in 8 32 128; do
  for mode in f16 bf16; do
    for g in "" "--graph"; do
      QEA_SPLIT=$mode python bench.py --batch $b --phase-b-only $g --steps 50 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null \
        | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$b', '$mode', '${g:-eager}', d['ms_per_step'], 'ms', round(d['value']), 'img/s')" >> $out
    done
  done
done
cat $out
