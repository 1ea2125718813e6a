#!/bin/bash
# kernel times of the MLP chain launches for the production library and the diagnostic variants (_diag/libpit_vchain<mask>.so)
#   tools/chain_ab.sh <task> <batch> [mask ...]
task=${1:-vorticity}; batch=${2:-20}; shift 2
for lib in prod "$@"; do
  if [ $lib = prod ]; then unset PIT_LIB_PATH; else export PIT_LIB_PATH=$PWD/_diag/libpit_vchain$lib.so; fi
  rm -rf gpurun_out/chainab/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/chainab/$lib -o t -- python3 bench.py --task $task --batch $batch --math bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-parity > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/chainab/$lib/**/t_kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("$lib", " ".join(r["Name"].split("(anonymous namespace)::")[1].split("(")[0]+"="+str(round(float(r["AverageNs"])/1e3,1)) for r in rows if "mlp_chain" in r["Name"]))
PY
done
