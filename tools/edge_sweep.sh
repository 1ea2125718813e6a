for spec in "darcy 8" "darcy 16" "darcy 32" "darcy 64" "darcy 128" "burgers 8" "sod 8"; do
  set -- $spec
  for e in 1 0; do
    PIT_EDGE_FUSION=$e python bench.py --task $1 --batch $2 --no-cpu-baseline --no-extras --no-parity --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 b=$2 edge=$e', d['ms_per_step'], 'ms', round(d['value']), 'samples/s')"
  done
done
