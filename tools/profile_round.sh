#!/bin/bash
# Per-configuration evidence for profiles/ (GPU box): for every configuration one rocprofv3 kernel-trace
# run (--stats) and four PMC passes, EACH IN ITS OWN RUN with --kernel-trace only (MI355X guide:
# FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2; never combined with other trace domains).
#   tools/profile_round.sh <outdir> [config ...]      configs: darcy8 darcy256 vort vort_bf16 elast elast_bf16 naca naca_bf16
#                                                               rollout20 cyl200 zssr421 (summarised over the whole run)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(mkdir -p "$1" && cd "$1" && pwd); shift
CONFIGS=${@:-darcy8 darcy256 vort vort_bf16 elast elast_bf16 naca naca_bf16}
cd /tmp; export TMPDIR=/tmp
for cfg in $CONFIGS; do
  case $cfg in
    darcy8)    ARGS="--task darcy --batch 8" ;;
    darcy256)  ARGS="--task darcy --batch 256" ;;
    vort)      ARGS="--task vorticity --batch 20" ;;
    vort_bf16) ARGS="--task vorticity --batch 20 --math bf16" ;;
    elast)     ARGS="--task elasticity --batch 10" ;;
    elast_bf16) ARGS="--task elasticity --batch 10 --math bf16" ;;
    naca)      ARGS="--task naca --batch 20" ;;
    naca_bf16) ARGS="--task naca --batch 20 --math bf16" ;;
    rollout20) ARGS="--task vorticity --batch 20 --rollout 20"; STEPS="--steps 4 --warmup 1"; MODE=whole ;;
    cyl200)    ARGS="--task cylinder --batch 200"; STEPS="--steps 6 --warmup 2"; MODE=whole ;;
    zssr421)   PROG="$R/tools/zssr_probe.py"; ARGS=""; STEPS=""; MODE=whole ;;
    *) echo "unknown config $cfg"; continue ;;
  esac
  PROG=${PROG:-$R/bench.py}; STEPS=${STEPS---steps 20 --warmup 3}; MODE=${MODE:-step}
  BARGS="$ARGS $STEPS --no-cpu-baseline --no-extras --no-parity"
  [ "$PROG" != "$R/bench.py" ] && BARGS=""
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$cfg/trace" -o t -- python3 "$PROG" $BARGS > "$OUT/$cfg.bench.json" 2> "$OUT/$cfg.trace.log" || echo "$cfg trace failed"
  for grp in "FETCH:FETCH_SIZE" "WRITE:WRITE_SIZE" "MFMA1:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "MFMA2:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES"; do
    name=${grp%%:*}; ctrs=${grp#*:}
    timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/$cfg/$name" -o p -- python3 "$PROG" $BARGS > /dev/null 2> "$OUT/$cfg.$name.log" || echo "$cfg $name failed"
    find "$OUT/$cfg/$name" -name "*kernel_trace.csv" -delete
  done
  python3 "$R/tools/profile_summary.py" "$OUT" $cfg $MODE
  unset PROG STEPS MODE
  # keep only what the summary needs out of the (large) raw output
  find "$OUT/$cfg" -name "*kernel_trace.csv" -delete
  find "$OUT/$cfg" -name "*counter_collection.csv" -delete
done
