#!/bin/bash
# One diagnostic library: csrc/<file>.hip rebuilt with extra -D flags and linked with the production objects into
# _diag/libpit_v<name>.so (time it with tools/edge_ab.sh; results of such a build are void, times are not).
#   tools/variant_build.sh <name> <file.hip> -DFLAG[=v] ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/position_induced_transformer_amd/csrc
name=$1; src=$2; shift 2
mkdir -p $R/_diag
base=$(basename $src .hip)
OBJS=$(python $R/tools/prod_objects.py $base)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 "$@" -c $C/$src -o $R/_diag/${base}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $R/_diag/libpit_v$name.so $OBJS $R/_diag/${base}_$name.o
echo built _diag/libpit_v$name.so
