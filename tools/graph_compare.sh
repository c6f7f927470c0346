#!/bin/bash
# The train step at the reference's own batch sizes (trainer/task.py:186: 16, 8, 4, 2, 2, 2, 2 for stages 0-6):
# eager against HIP-graph replay (bench.py --graph, what train_step does by itself for batches of 16 and fewer),
# fp32 and bf16; prints stage, batch, precision, mode, ms per step, samples/s.  MGX_CFGS="6 2,5 2" picks (stage batch) pairs.
set -e -o pipefail
IFS=","
for cfg in ${MGX_CFGS:-0 16,1 8,2 4,3 2,4 2,5 2,6 2}; do
  IFS=" "
  set -- $cfg
  extra=""; [ "$1" = "0" ] && extra="--straight"
  for prec in float32 bfloat16; do
    for mode in "" "--graph"; do
      [ "$1" = "6" ] && [ -n "$mode" ] && continue      # the full-resolution step is not replayed (memory)
      timeout -k 10 300 python bench.py --stage $1 --batch $2 --steps 10 --warmup 3 --no-cpu-baseline --no-extras \
          --precision $prec $mode $extra 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('stage $1 batch $2 %-8s %-6s %9.2f ms/step %9.2f samples/s' % ('$prec', 'graph' if '$mode' else 'eager', d['ms_per_step'], d['value']), flush=True)"
    done
  done
done
