#!/bin/bash
# round 6, first GPU pass: new tests, then A/B of the folded decoder on the two configurations it targets
mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > gpurun_out/r6a/tests.txt 2>&1
tail -30 gpurun_out/r6a/tests.txt
for fold in 1 0; do
  for spec in "vorticity 20" "naca 20"; do
    set -- $spec
    for m in fp32 bf16; do
      PIT_FOLD_DECODER=$fold timeout 300 python bench.py --task $1 --batch $2 --math $m --no-extras --no-cpu-baseline --steps 100 --warmup 10 2>gpurun_out/r6a/err_$1_$m_$fold.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('fold=$fold $1 b=$2 $m', d['ms_per_step'], 'ms', d['value'], 'samples/s', d.get('parity'))" 2>&1 | tee -a gpurun_out/r6a/times.txt
    done
  done
done
