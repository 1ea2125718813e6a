python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "carried_by or postponed or raises_midway or vorticity or rollout_step" 2>&1 | tail -2
python bench.py --task vorticity --batch 20 --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vorticity', d['ms_per_step'])"
python bench.py --task vorticity --batch 20 --math bf16 --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vorticity bf16', d['ms_per_step'])"
for i in 1 2; do python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('darcy b=8', d['ms_per_step'])"; done
