#!/bin/bash
mkdir -p gpurun_out/r6d
for rows in 4611686018427387904 1; do
  for b in 64 128 256; do
    PIT_FOLD_EDGE_ROWS=$rows timeout 300 python bench.py --task darcy --batch $b --math fp32 --no-extras --no-cpu-baseline --steps 100 --warmup 10 2>gpurun_out/r6d/err_$b.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('fold_edge_rows=$rows darcy b=$b', d['ms_per_step'], 'ms', d['value'], 'samples/s', (d.get('parity') or {}).get('rel_l2_out'), (d.get('parity') or {}).get('rel_l2_weight_grad_worst'))" 2>&1 | tee -a gpurun_out/r6d/times.txt
  done
done
