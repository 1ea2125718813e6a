#!/bin/bash
# Kernel-time breakdown of one bench.py configuration (GPU box): rocprofv3 --kernel-trace --stats, top kernels by total time.
#   tools/quick_prof.sh <outdir> <bench args...>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(mkdir -p "$1" && cd "$1" && pwd); shift
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$R/bench.py" "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-parity > "$OUT/bench.json" 2> "$OUT/trace.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
b = json.load(open(out + "/bench.json"))
print(f"{b['ms_per_step']} ms/step under the profiler; kernel time shares:")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}%  n={int(r['Calls'])//23:3d}/step  mean {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
