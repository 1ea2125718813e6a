#!/bin/bash
# union-tile attention for wide masked layers on shared meshes (pit_union_att_*): step times with and without (PIT_UNION_ATT=0)
for spec in "vorticity 20 fp32" "vorticity 20 bf16" "cylinder 200 fp32"; do
  set -- $spec
  for u in 1 0; do
    PIT_UNION_ATT=$u python bench.py --task $1 --batch $2 --math $3 --no-extras --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d.get('parity') or {}
print('$1 b=$2 $3 union_att=$u', d['ms_per_step'], 'ms', 'rel_l2_out', p.get('rel_l2_out'), 'wgrad', p.get('rel_l2_weight_grad_worst'), 'dlmda', p.get('rel_l2_dlmda_all_layers'))"
  done
done
