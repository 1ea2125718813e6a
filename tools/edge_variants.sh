#!/bin/bash
# Diagnostic builds of csrc/pit_edge.hip with parts of the fused encoder- / decoder-side launches switched off (PIT_EDGE_DBG bits:
# 1 decoder fwd without its W1 loads, 2 without the union gather (also the backward's), 4 without the saved-activation stores,
# 8 decoder bwd without its W1 loads, 16 without the d(values) atomics, 64 encoder without its value gathers), each linked with the
# production objects into _diag/libpit_v<mask>.so; time them with PIT_LIB_PATH=... python bench.py (results are void, times are not).
#   tools/edge_variants.sh 1 2 4 ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/position_induced_transformer_amd/csrc
mkdir -p $R/_diag
OBJS=$(python $R/tools/prod_objects.py pit_edge)
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -DPIT_EDGE_DBG=$v -c $C/pit_edge.hip -o $R/_diag/edge_v$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $R/_diag/libpit_v$v.so $OBJS $R/_diag/edge_v$v.o
  echo built _diag/libpit_v$v.so
done
