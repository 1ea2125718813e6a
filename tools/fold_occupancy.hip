// Diagnostic (GPU box): workgroups per CU the runtime grants the fold kernels at their launch shapes.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I position_induced_transformer_amd/csrc -I include tools/fold_occupancy.hip -o _diag/fold_occupancy
#include "pit_fold.hip"
#include <cstdio>
template <typename K> static void show(const char* name, K k, int threads, size_t smem) {
    int n = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, threads, smem);
    printf("%-28s threads %d  dynamic LDS %6zu B  -> %d workgroups per CU (%s)\n", name, threads, smem, n, hipGetErrorString(e));
}
int main() {
    show("fold_bwd<2, fp32> um 64", fold_bwd_kernel<2, false>, NTB, fold_smem(2, false, 64, true));
    show("fold_bwd<2, bf16> um 64", fold_bwd_kernel<2, true>, NTB, fold_smem(2, true, 64, true));
    show("fold_bwd<1, fp32> um 64", fold_bwd_kernel<1, false>, NTB, fold_smem(1, false, 64, true));
    show("fold_fwd<2, fp32> um 64", fold_fwd_kernel<2, false, NTB>, NTB, fold_smem(2, false, 64, false));
    show("fold_fwd<2, bf16> um 64", fold_fwd_kernel<2, true, 256>, 256, fold_smem(2, true, 64, false));
    return 0;
}
