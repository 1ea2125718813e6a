#!/bin/bash
# round 6 quick pass: the round's tests (full failure text), then step times of the configurations given as "task batch" pairs
mkdir -p gpurun_out/$1; out=gpurun_out/$1; shift
timeout 900 python -m pytest tests/test_gpu_round6.py -q > $out/tests.txt 2>&1
grep -E "^(E  |FAILED|[0-9]+ (passed|failed))" $out/tests.txt | head -40
while [ $# -gt 1 ]; do
  task=$1; batch=$2; shift 2
  for m in fp32 bf16; do
    timeout 300 python bench.py --task $task --batch $batch --math $m --no-extras --no-cpu-baseline --no-parity --steps 100 --warmup 10 2>$out/err_${task}_$m.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$task b=$batch $m', d['ms_per_step'], 'ms', d['value'], 'samples/s')" 2>&1 | tee -a $out/times.txt
  done
done
