#!/bin/bash
# per-kernel times of tools/dw_bench.py (GPU box):  tools/dw_prof.sh <outdir> [shape ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(mkdir -p "$1" && cd "$1" && pwd); shift
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$R/tools/dw_bench.py" "$@" > "$OUT/bench.txt" 2> "$OUT/trace.log"
python3 "$R/tools/kstats.py" $(ls "$OUT"/trace/*/*kernel_stats.csv "$OUT"/trace/*kernel_stats.csv 2>/dev/null | head -1) 8
