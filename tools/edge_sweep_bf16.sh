for spec in "32" "64" "256"; do
  for e in 1 0; do
    PIT_EDGE_FUSION=$e python bench.py --batch $spec --math bf16 --no-cpu-baseline --no-extras --no-parity --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('darcy bf16 b=$spec edge=$e', d['ms_per_step'], 'ms', round(d['value']), 'samples/s')"
  done
done
python bench.py --batch 256 --no-cpu-baseline --no-extras --no-parity --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('darcy fp32 b=256', d['ms_per_step'], 'ms', round(d['value']), 'samples/s')"
