for i in 1 2 3; do
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('now ', d['ms_per_step'])"
PIT_LIB_PATH=$PWD/position_induced_transformer_amd/csrc/libpit_hip_prev.so python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prev', d['ms_per_step'])"
done
