for i in 1 2 3; do
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair rider on ', d['ms_per_step'])"
PIT_NO_LIST_PAIR_RIDER=1 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair rider off', d['ms_per_step'])"
done
