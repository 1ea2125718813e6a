#!/bin/bash
# the whole GPU suite (timed), output under gpurun_out/$1
mkdir -p gpurun_out/$1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/$1/suite.txt 2>&1
grep -E "^(E  |FAILED|ERROR|[0-9]+ (passed|failed)|=+ .* in )" gpurun_out/$1/suite.txt | head -40

grep -A18 "slowest 15" gpurun_out/$1/suite.txt
