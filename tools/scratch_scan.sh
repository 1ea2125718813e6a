#!/bin/bash
# private_segment (scratch) bytes per lane and VGPR count of every kernel of the library: kernels that keep part of their
# working set in scratch memory.  tools/scratch_scan.sh > profiles/r06_scratch_scan.txt   (CPU only: hipcc cross-compiles)
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/position_induced_transformer_amd/csrc
echo "# hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Rpass-analysis=kernel-resource-usage, per translation unit"
echo "# columns: scratch bytes/lane, VGPRs, kernel (only kernels with scratch > 0 are listed; the count of all kernels follows)"
for f in $C/*.hip; do
  out=$(/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -c $f -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
        | grep -E "Function Name|VGPRs:|ScratchSize" | paste - - - | sed 's/\[-Rpass[^ ]*//g')
  total=$(echo "$out" | grep -c "Function Name")
  echo "== $(basename $f): $total kernels"
  echo "$out" | awk '{ n=""; v=""; s=""; for (i=1;i<=NF;i++) { if ($i=="Name:") n=$(i+1); if ($i=="VGPRs:") v=$(i+1); if ($i=="[bytes/lane]:") s=$(i+1) } if (s+0 > 0) printf "%6d %4d  %s\n", s, v, n }' | sort -rn | c++filt | cut -c1-170
done
