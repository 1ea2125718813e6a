#!/bin/bash
# rocprofv3 kernel shares + PMC passes for the configurations given (tools/profile_round.sh), output under gpurun_out/$1
out=$1; shift
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q 2>&1 | tail -3
bash tools/profile_round.sh gpurun_out/$out "$@" 2>&1 | tail -5
for c in "$@"; do cat gpurun_out/$out/$c.summary.txt | cut -c1-190; done
