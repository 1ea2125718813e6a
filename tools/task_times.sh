#!/bin/bash
# fwd+loss+bwd step time of the larger task configurations, fp32 and bf16 math modes (bench.py, hipGraph)
# usage: tools/task_times.sh [extra env assignments...]
for spec in "vorticity 20" "elasticity 10" "naca 20" "burgers 8" "darcy 256"; do
  set -- $spec
  for m in fp32 bf16; do
    python bench.py --task $1 --batch $2 --math $m --no-extras --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1 b=$2 $m', d['ms_per_step'], 'ms', d['value'], 'samples/s', d.get('step_tflops', {}).get('achieved'))"
  done
done
