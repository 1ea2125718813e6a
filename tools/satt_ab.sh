#!/bin/bash
# kernel times of the satt launches for the production library and the diagnostic variants (_diag/libpit_vsatt<mask>.so)
task=${1:-elasticity}; batch=${2:-10}; shift 2
for lib in prod "$@"; do
  if [ $lib = prod ]; then unset PIT_LIB_PATH; else export PIT_LIB_PATH=$PWD/_diag/libpit_vsatt$lib.so; fi
  rm -rf gpurun_out/sattab/$lib
  PIT_SATT=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sattab/$lib -o t -- python3 bench.py --task $task --batch $batch --math bf16 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-parity > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/sattab/$lib/**/t_kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("$lib", " ".join(r["Name"][28:58].split("(")[0]+"="+str(round(float(r["AverageNs"])/1e3,1)) for r in rows if "satt_kernel" in r["Name"]))
PY
done
