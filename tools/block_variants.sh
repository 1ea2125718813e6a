#!/bin/bash
# Diagnostic builds of csrc/pit_block.hip with part of the fused processor kernels' matrix work switched off (PIT_BLOCK_EXP bits:
# 1 forward contraction issues half its MFMAs, 2 forward GEMM1 half its k-steps, 4 backward contraction half its MFMAs, 8 backward
# dX phase one tile per wave; all loads kept), each linked with the production objects into _diag/libpit_vb<mask>.so; time them with
# tools/edge_ab.sh (results are void, times are not).   tools/block_variants.sh 1 2 3 4 ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/position_induced_transformer_amd/csrc
mkdir -p $R/_diag
OBJS=$(python $R/tools/prod_objects.py pit_block)
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -DPIT_BLOCK_EXP=$v -c $C/pit_block.hip -o $R/_diag/block_v$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $R/_diag/libpit_vb$v.so $OBJS $R/_diag/block_v$v.o
  echo built _diag/libpit_vb$v.so
done
