#!/bin/bash
# Per-kernel times (rocprofv3 --kernel-trace --stats over 20 bench steps) of the fused encoder- / decoder-side launches for the
# production library and every _diag/libpit_v*.so variant (tools/edge_variants.sh).  GPU box:  tools/edge_ab.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for lib in "" $(ls $R/_diag/libpit_v*.so 2>/dev/null); do
  name=${lib:-production}; name=$(basename $name .so)
  rm -rf /tmp/edge_ab_$name
  PIT_LIB_PATH=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/edge_ab_$name -o t -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-parity --steps 20 --warmup 3 "$@" > /tmp/edge_ab_$name.json 2>/dev/null
  python3 - "$name" /tmp/edge_ab_$name <<'PY'
import csv, glob, sys, json
name, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
pick = {}
for r in rows:
    for k in ("decoder_fwd", "decoder_bwd", "encoder_fwd", "encoder_bwd", "block_fwd", "block_bwd", "mlp_bwd16", "dhead_finish", "gemm_rd_pair", "mlp_fwd64", "mlp_bwd64", "gemm_rr", "thin_dw", "posatt_cols_tiles", "posatt_rows_tiles"):
        if k in r["Name"] and int(r["Calls"]) > 100:
            pick[k] = float(r["AverageNs"]) / 1e3
try:
    ms = json.loads(open(d + ".json").read())["ms_per_step"]
except Exception:
    ms = None
print(f"{name:14s} ms/step {ms}  " + "  ".join(f"{k} {v:.1f}" for k, v in pick.items()))
PY
done
