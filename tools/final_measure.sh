#!/bin/bash
# Round-end measurement on the GPU box: default bench line, task times (fp32 / bf16 mode), Cylinder, the 20-step rollout,
# Sod, and the per-configuration rocprofv3 passes (tools/profile_round.sh).  Outputs under gpurun_out/; copy what is to be
# judged into profiles/ (tools/make_pmc_json.py for the traffic JSON).
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
bash tools/task_times.sh > gpurun_out/final/task_times.txt 2>&1
python bench.py --task cylinder --batch 200 --no-extras --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cylinder b=200 fp32', d['ms_per_step'], 'ms', d['value'], 'samples/s')" >> gpurun_out/final/task_times.txt
python bench.py --task vorticity --batch 20 --rollout 20 --no-extras --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('vorticity rollout-20 b=20 fp32', d['ms_per_step'], 'ms/optimiser step', d.get('peak_memory_GB'), 'GB peak')" >> gpurun_out/final/task_times.txt
python bench.py --task sod --batch 8 --no-extras --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('sod b=8 fp32', d['ms_per_step'], 'ms')" >> gpurun_out/final/task_times.txt
# (per-configuration rocprofv3 passes: run tools/profile_round.sh <outdir> separately)
cat gpurun_out/final/task_times.txt; tail -c 1500 gpurun_out/final/bench_default.json
