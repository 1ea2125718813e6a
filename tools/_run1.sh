python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "carried_by or postponed_weight or raises_midway or inplace" 2>&1 | tail -5
python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rider on ', d['ms_per_step'])"
PIT_DW_RIDER=0 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rider off', d['ms_per_step'])"
done
