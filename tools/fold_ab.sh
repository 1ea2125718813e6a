#!/bin/bash
# kernel times of the fold launches for the production library and the diagnostic variants (_diag/libpit_vfold<mask>.so)
#   tools/fold_ab.sh <task> <batch> <math> [mask ...]
task=${1:-darcy}; batch=${2:-256}; math=${3:-fp32}; shift 3
for lib in prod "$@"; do
  if [ $lib = prod ]; then unset PIT_LIB_PATH; else export PIT_LIB_PATH=$PWD/_diag/libpit_vfold$lib.so; fi
  rm -rf gpurun_out/foldab/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/foldab/$lib -o t -- python3 bench.py --task $task --batch $batch --math $math --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-parity > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/foldab/$lib/**/t_kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("$lib", " ".join(r["Name"].split("(")[0]+"="+str(round(float(r["AverageNs"])/1e3,1)) for r in rows if "fold_" in r["Name"]))
PY
done
