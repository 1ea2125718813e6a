#!/bin/bash
# HBM traffic of the bench's kernels with rocprofv3 PMC counters (separate passes, kernel-trace
# only, as MI355X_MICROARCH.md prescribes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).
# Usage (on the GPU box): tools/pmc_traffic.sh <outdir> [bench args...]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extras "$@" > "$OUT/$c.log" 2>&1 || echo "pass $c failed"
done
