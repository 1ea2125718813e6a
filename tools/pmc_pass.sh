#!/bin/bash
# rocprofv3 PMC passes (each counter group in its own run, kernel-trace only) over a bench.py
# invocation; per-kernel averages printed by tools/pmc_summary.py.
# Usage (GPU box): tools/pmc_pass.sh <outdir> "<bench args>" "<counters group 1>" ["<group 2>" ...]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(mkdir -p "$1" && cd "$1" && pwd); ARGS=$2; shift 2
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 "$R/bench.py" --no-cpu-baseline --no-extras $ARGS > "$OUT/g$i.log" 2>&1 || echo "pass $i failed"
  find "$OUT/g$i" -name "*kernel_trace.csv" -delete
done
