#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per access shape against a known byte count (tools/micro/fetch_calib.hip), separate PMC passes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(mkdir -p "$1" && cd "$1" && pwd)
cd /tmp; export TMPDIR=/tmp
for mib in 64 512; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c.$mib" -o p -- "$R/tools/micro/fetch_calib" $mib > "$OUT/$c.$mib.log" 2>&1
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for mib in (64, 512):
    print(f"buffer {mib} MiB = {mib*1024} KiB")
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"{out}/{c}.{mib}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            m = sum(v) / len(v)
            print(f"  {c:10s} {k:16s} {m:12.1f} KiB   ratio to buffer {m / (mib * 1024):.3f}")
PY
