#!/bin/bash
# All PMC passes behind profiles/r01_pmc_{traffic,mfma}.json (GPU box): each counter group in its own
# rocprofv3 run with --kernel-trace only.   tools/pmc_refresh.sh <outdir>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(mkdir -p "$1" && cd "$1" && pwd)
cd /tmp; export TMPDIR=/tmp
for b in 8 256; do
  for grp in "FETCH_SIZE" "WRITE_SIZE" "MFMA1:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "MFMA2:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES"; do
    name=${grp%%:*}; ctrs=${grp#*:}
    timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/b${b}_$name" -- python3 "$R/bench.py" --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/b${b}_$name.log" 2>&1 || echo "pass b$b $name failed"
    find "$OUT/b${b}_$name" -name "*kernel_trace.csv" -delete
  done
done
