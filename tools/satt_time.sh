#!/bin/bash
# step time and the satt kernels' mean times for the three mid-sized tasks in bf16 mode (PIT_SATT=1 forces the path)
for t in "elasticity 10" "naca 20" "vorticity 20"; do set -- $t
  rm -rf gpurun_out/sattab/prod
  PIT_SATT=${PIT_SATT:-1} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sattab/prod -o t -- python3 bench.py --task $1 --batch $2 --math bf16 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d[\"config\"][\"workload\"][:40], d[\"ms_per_step\"])"
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/sattab/prod/**/t_kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print(" ".join(r["Name"][28:58].split("(")[0]+"="+str(round(float(r["AverageNs"])/1e3,1)) for r in rows if "satt_" in r["Name"]))
PY
done
