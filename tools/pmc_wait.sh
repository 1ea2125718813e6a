#!/bin/bash
# Where do the waves of a configuration's kernels spend their cycles?  One rocprofv3 PMC pass (kernel-trace only):
#   tools/pmc_wait.sh "<bench args>" [kernel-name filter ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS=$1; shift
OUT=$R/gpurun_out/pmc_wait_tmp
rm -rf $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p -- python3 $R/bench.py $ARGS --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - "$OUT" "$@" <<'PY'
import csv, glob, collections, sys
out, flt = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r["Counter_Name"] == "SQ_WAVE_CYCLES":
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    if flt and not any(x in k for x in flt):
        continue
    w = sum(cs["SQ_WAVE_CYCLES"]) / len(cs["SQ_WAVE_CYCLES"])
    if w <= 0:
        continue
    line = f"{k:70s}"
    if dur[k]:
        us = sum(dur[k]) / len(dur[k])
        ghz = sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"]) / 8 / (us * 1e3)
        line += f" {us:8.1f} us  clk~{ghz:4.2f} GHz"
    print(line)
    for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY"):
        if c in cs:
            print(f"      {c:22s} {100 * sum(cs[c]) / len(cs[c]) / w:5.1f} % of wave cycles")
PY
rm -rf $OUT
