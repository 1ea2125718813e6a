/* pit_hip.h -- C ABI of the MI355X-native PiT position-attention hot path.
 *
 * The reference (junfeng-chen/position_induced_transformer) has no FFI: its hot path
 * is a sequence of torch ops inside pit.py.  Each entry point below replaces one such
 * sequence (cited as pit.py:line) and is what a ctypes / pybind / cgo binding for this
 * path would bind.  Conventions:
 *   - plain `extern "C"`, raw DEVICE pointers (fp32 unless stated), int sizes, strides
 *     in ELEMENTS; `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - every function returns 0 on success, a negative PIT_ERR_* for an argument error,
 *     or a positive hipError_t; nothing allocates, frees or synchronises, so every call can
 *     be captured into a hipGraph; there is no MUTABLE process-global state (the math mode, the head-scale convention and
 *     the storage formats are arguments).  What the library does read once per process: diagnostic environment switches
 *     (PIT_NO_* / PIT_LDS_* / PIT_RR_*: force a slower kernel family for A/B measurements, listed in DESIGN.md section 4;
 *     never needed for correct results), and per-kernel function attributes (dynamic LDS limits) set on first use;
 *   - outputs and workspaces are caller-owned (the PyTorch host code allocates them).
 *
 * Mesh conventions: `mesh_batch` = 1 for the batch-free (fixed) meshes of
 * posatt_fixed / _periodic1d / _periodic2d (pit.py:129-144,186-200,243-258) where
 * the attention weights are shared by the whole batch, and = b for the per-sample
 * meshes of posatt (pit.py:46-52).  Meshes are (mesh_batch, n, space_dim) contiguous,
 * space_dim in {1,2,3}.
 */
#ifndef PIT_HIP_H
#define PIT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PIT_ABI_VERSION 23
#define PIT_DSCALE_SLOTS 1024 /* fp64 accumulators per head in pit_posatt_bwd's workspace */

/* distance metric (dist2att variants) */
#define PIT_METRIC_EUCLID     0   /* pit.py:47,134  sum_c (xo_c-xi_c)^2                         */
#define PIT_METRIC_PERIODIC1D 1   /* pit.py:190-194 wrap |dx| to min(|dx|, l-|dx|), coordinate 0 */
#define PIT_METRIC_PERIODIC2D 2   /* pit.py:248-253 per-coordinate wrap, then sum of squares     */

/* argument errors */
#define PIT_ERR_NULL     -1
#define PIT_ERR_SIZE     -2
#define PIT_ERR_METRIC   -3
#define PIT_ERR_UNSUPPORTED -4

int pit_version(void);
/* human-readable text for a return code of any function below */
const char* pit_error_string(int code);

/* Math mode of the MFMA contractions (attention forward and d(values), MLP GEMMs): the `math_mode`
 * ARGUMENT of every call that contracts - the library keeps no process-wide mode (thread-safe; a
 * captured hipGraph keeps the mode its launches were captured with).  The reference computes in
 * fp32 (pit.py has no autocast), so
 *   PIT_MATH_FP32: v_mfma_f32_32x32x2_f32, exact fp32 products - the parity mode;
 *   PIT_MATH_BF16: operands rounded to bf16 (RNE), bf16 MFMA with fp32 accumulation.  Distances,
 *     head scale, quantile thresholds, mask, softmax weights, the d(scale) reduction, loss and
 *     optimiser stay fp32 in both modes, so the kept sets are identical; outputs agree with the
 *     fp32 mode to ~1e-2 relative L2 (tests/test_gpu_bf16.py states the tolerance).
 *     Exception (stated, not silent): the small-regime fused kernels - pit_mlp_fwd / pit_mlp_bwd_data on 16-row slabs
 *     (mlp_fwd16_kernel, mlp_bwd16_kernel) and the fused processor blocks (pit_block_*: PIT_MATH_FP32 only, they
 *     return PIT_ERR_UNSUPPORTED otherwise) - are latency-bound, not MFMA-bound, and contract in fp32 in BOTH modes.
 * Any other value returns PIT_ERR_UNSUPPORTED. */
#define PIT_MATH_FP32 0
#define PIT_MATH_BF16 1
/* bf16 STORAGE of the large decoder-side tensors, OR-ed into the `math_mode` argument (PIT_MATH_BF16 only; ABI 11).
 * BASELINE configs 3 and 5 (Vorticity, NACA) are bound by the fp32 traffic of the decoder tail - the up-projection's
 * output (rows x H*hid), the decoder MLP's saved pre-activations and their gradients: 81 920 / 225 420 rows - not by
 * the matrix pipe; with these flags those tensors live in memory as bf16 (widened exactly on load, RNE on store), every
 * accumulation stays fp32.  Distances, mask, softmax weights, the d(scale) reduction, parameters and their gradients
 * are untouched.  A call whose shape does not take the kernels that implement a flag returns PIT_ERR_UNSUPPORTED
 * (ask pit_mlp_bf16_io_supported first).
 *   pit_posatt_fwd   PIT_IO_OUT_BF16   `out` is bf16 (candidate-list layers, no input copy)
 *   pit_posatt_bwd   PIT_IO_DOUT_BF16  `d_out` is bf16 (candidate-list layers)
 *   pit_mlp_fwd      PIT_IO_X_BF16     `x` is bf16;  PIT_IO_SAVE_BF16: z1, h are written as bf16
 *   pit_mlp_bwd*     PIT_IO_X_BF16, PIT_IO_SAVE_BF16 (z1, h and the dZ1 scratch - rows*n1 bf16 - are bf16),
 *                    PIT_IO_DX_BF16    `d_x` is written as bf16 */
#define PIT_IO_X_BF16    0x100
#define PIT_IO_SAVE_BF16 0x200
#define PIT_IO_DX_BF16   0x400
#define PIT_IO_OUT_BF16  0x800
#define PIT_IO_DOUT_BF16 0x1000
/* pit_posatt_fwd / pit_posatt_bwd, masked layers on candidate lists (round 4): OR-ed into math_mode, PIT_ATT_UNION asks for the
 * UNION-TILE kernels - 16 consecutive rows (one wavefront) contract against the union of their candidate keys (every key's value
 * row fetched once per tile, v_mfma_f32_16x16x4_f32) instead of one wavefront per row gathering its own keys.  Exact for any
 * input, fast when consecutive rows share their keys (grids, body-fitted meshes: a 16-row tile's union is 25-60 keys); the
 * caller decides per mesh plan.  Ignored where it does not apply: coordinate channels, self-attention concat, n_in > 4096,
 * n_head > 2, list capacity > 64, dim not a multiple of 8 or rows not 16-byte aligned (the per-row kernels run instead).
 * pit_posatt_bwd with this flag: d(scale) from the tiles; d(values) from the transposed lists when rev_ptr / rev_row are
 * given, else from the tiles (256-row blocks contracted on MFMA, their sums ADDED to memory with fp32 atomics - the call
 * zeroes d_values first; needs a dense d_values: ld_dvalues = dim, dvalues_bstride = n_in*dim; sums differ in the last
 * bits from run to run). */
#define PIT_ATT_UNION    0x2000
/* 1 if pit_mlp_fwd / pit_mlp_bwd* of this shape accept PIT_IO_X_BF16 | PIT_IO_SAVE_BF16 | PIT_IO_DX_BF16 (thin output
 * layer n2 <= 4 without trailing gelu, large regime, widths multiples of 8) */
int pit_mlp_bf16_io_supported(int rows, int n0, int n1, int n2, int out_gelu);

/* pit.py:48 (and :135,:196,:254): c_h = tan(0.25*pi*(1-1e-7)*(1+sin(lmda_h))).
 * Evaluated through fp64 with the reference's fp32 intermediate roundings. */
int pit_head_scale(const float* lmda, int n_head, float* scale_out, void* stream);

/* Selection pre-pass replacing the sort inside torch.quantile (pit.py:49,136,197,255).
 * For every row (mesh_batch*n_out rows of length n_in) of the UNSCALED squared distance
 * writes stats[0][row] = m_(k), stats[1][row] = m_(k+1) (k+1 clipped to n_in-1) and
 * stats[2][row] = min_j m.  rank_k = floor(fl32(q)*fl32(n_in-1)) (0-based).  With
 * need_kth = 0 only the row minimum is computed (locality 1.0: nothing is masked).
 * stats is 3*mesh_batch*n_out floats. */
int pit_select_fwd(const float* mesh_out, const float* mesh_in, int mesh_batch, int n_out, int n_in,
                   int space_dim, int metric, float period, int rank_k, int need_kth,
                   float* stats, void* stream);

/* The transposed lists (key -> listing rows) of candidate lists that were built WITHOUT them (rev_ptr = NULL in pit_plan_fwd /
 * pit_neighbors_fwd): round 4 - per-sample plans are rebuilt every step and only a backward that walks the lists by key
 * (d(values) of the candidate-list kernels) needs the transpose; the forward and d(scale) do not.  rev_ptr (mesh_batch, n_in+1),
 * rev_row (mesh_batch, n_out*cap), workspace 2*mesh_batch*n_in ints - as in pit_neighbors_fwd.  n_in <= 4096
 * (PIT_ERR_UNSUPPORTED otherwise: rebuild the plan with rev_ptr). */
int pit_lists_transpose(const int* nbr_idx, const int* nbr_cnt, int mesh_batch, int n_out, int n_in, int cap,
                        int* rev_ptr, int* rev_row, int* workspace, void* stream);

/* pit_select_fwd (need_kth = 1) and pit_neighbors_fwd in ONE pass over the rows: the distances of
 * a row stay in registers, the order statistics are searched on a narrowed candidate set and the
 * lists are emitted from the same registers (rows longer than 4096 keys fall back to the two
 * streaming passes).  Arguments as in those two functions; stats is written, then used.
 * flags (0 = let the library choose; ABI 17 - these were environment reads per call): PIT_PLAN_WAVE_PER_ROW keeps the
 * wave-per-row kernel where the one-row-per-lane kernel (per-sample meshes from 32 768 rows) would run, PIT_PLAN_TWO_PASSES
 * forces the two streaming passes.  Same results either way (tests compare them). */
#define PIT_PLAN_WAVE_PER_ROW 1
#define PIT_PLAN_TWO_PASSES   2
int pit_plan_fwd(const float* mesh_out, const float* mesh_in, int mesh_batch, int n_out, int n_in,
                 int space_dim, int metric, float period, int rank_k, float* stats, int cap,
                 int* nbr_idx, int* nbr_cnt, int* rev_ptr, int* rev_row, int* workspace, int flags, void* stream);

/* Candidate lists for the masked layers (sparse path).  For every row: the keys with
 * m <= m_(k+1)*(1+2^-21) - a superset of the kept set of pit.py:50 for ANY head scale (k+2 keys
 * plus ties), so it depends on the meshes only.  nbr_idx (rows, cap) int32, nbr_cnt (rows) int32
 * holds the TRUE count (a row with count > cap is truncated and consumers scan all keys for it).
 * stats: from pit_select_fwd with need_kth=1.
 * Optionally (rev_ptr != NULL) also the transposed lists (key -> rows listing it) as CSR per mesh
 * sample, for d(values): rev_ptr (mesh_batch, n_in+1), rev_row (mesh_batch, n_out*cap) row
 * indices local to the sample, -1 = unused slot; workspace: 2*mesh_batch*n_in ints.  Rows with
 * count > cap are left out of the transpose (pit_posatt_bwd adds them densely). */
int pit_neighbors_fwd(const float* mesh_out, const float* mesh_in, int mesh_batch, int n_out, int n_in,
                      int space_dim, int metric, float period, const float* stats, int cap,
                      int* nbr_idx, int* nbr_cnt, int* rev_ptr, int* rev_row, int* workspace, void* stream);

/* Fused dist2att + convolution forward (pit.py:46-57 / 133-144 and the periodic
 * variants; posatt.forward :37-44 with copy_inputs, posatt_cross*.forward :63-71).
 *   values   (batch, n_in, dim)   rows ld_values apart, samples values_bstride apart
 *   head     n_head floats: lmda (head_is_scale=0) or the scale c itself (=1)
 *   stats    from pit_select_fwd (may be NULL when masked=0 and self_attn=1: min is 0)
 *   rank_w   fractional part of the quantile rank (fp32), used when masked=1
 *   out      (batch, n_out, ...) rows ld_out apart; head h / channel d is written at
 *            column out_col0 + h*dim + d; with copy_inputs=1 (self attention, n_out ==
 *            n_in) values[b,n,:] is also copied to columns [0,dim) -> torch.cat of :44
 *   rowstat  (mesh_batch, n_head, n_out, 4) = {T, S_min, 1/rowsum, sum_j P*m} saved for
 *            the backward; scale_out (n_head) receives the c that was used.
 *   nbr_idx/nbr_cnt/nbr_cap: candidate lists from pit_neighbors_fwd (masked layers); NULL = the
 *            dense MFMA kernel.  With lists the layer costs O(n_out * k * columns).
 *   coord_dims: > 0 fuses the coordinate concat of the task forwards (train_darcy.py:51-55:
 *            func_in = cat((tile(mesh_in), func_in), -1)): value channels [0, coord_dims) are the key
 *            coordinates, read from mesh_in; `values` then holds the other dim - coord_dims channels (and
 *            d_values of pit_posatt_bwd has that many).  Candidate-list kernels only
 *            (PIT_ERR_UNSUPPORTED otherwise: the caller materialises the concat). */
int pit_posatt_fwd(const float* mesh_out, const float* mesh_in, int mesh_batch, int n_out, int n_in,
                   int space_dim, int metric, float period,
                   const float* values, int batch, int dim, long ld_values, long values_bstride,
                   const float* head, int n_head, int head_is_scale,
                   const float* stats, float rank_w, int masked, int self_attn,
                   float* out, long ld_out, long out_bstride, int out_col0, int copy_inputs,
                   float* rowstat, float* scale_out,
                   const int* nbr_idx, const int* nbr_cnt, int nbr_cap, int coord_dims, int math_mode, void* stream);

struct pit_mlp_params_job;

/* pit_posatt_fwd with a RIDER (round 4): the processor's block weights (pit_block_weights, declared below - they depend on
 * the latent mesh and the lmda's only, not on this layer's data) formed by extra workgroups of the SAME launch when this
 * layer runs on the small candidate-list kernel (the down-projection of the small regime), by a launch of their own
 * right after it otherwise.  job == NULL: exactly pit_posatt_fwd.  The job's fields are pit_block_weights' arguments. */
struct pit_block_weights_job {
    const float* mesh; int n_pts, space_dim, metric; float period;
    int n_layers; const float* const* heads; int head_is_scale, n_head;
    float *e, *q, *inv, *rowstat, *scale_out;
};
int pit_posatt_fwd_job(const float* mesh_out, const float* mesh_in, int mesh_batch, int n_out, int n_in,
                       int space_dim, int metric, float period,
                       const float* values, int batch, int dim, long ld_values, long values_bstride,
                       const float* head, int n_head, int head_is_scale,
                       const float* stats, float rank_w, int masked, int self_attn,
                       float* out, long ld_out, long out_bstride, int out_col0, int copy_inputs,
                       float* rowstat, float* scale_out,
                       const int* nbr_idx, const int* nbr_cnt, int nbr_cap, int coord_dims, int math_mode, void* stream,
                       const struct pit_block_weights_job* job);

/* Backward of pit_posatt_fwd (closed form; the reference uses
 * autograd).  d_out has the layout of `out` (columns out_col0 + h*dim + d).
 *   d_values (batch, n_in, dim): written (NULL = not needed).  With add_residual=1 (self
 *            attention) the gradient of the copied inputs, d_out[b,j,0:dim], is added.
 *   scale    the c written to scale_out by the forward (NULL = recompute from head)
 *   d_head   n_head floats (NULL = not needed): gradient w.r.t. lmda (head_is_scale=0) or
 *            w.r.t. c (=1); accumulate_head is a bit set: PIT_HEAD_ACCUMULATE adds to the current
 *            contents instead of writing; PIT_HEAD_DEFER leaves the fp64 accumulators LOADED and
 *            skips the finishing kernel - the caller later drains several layers at once with
 *            pit_posatt_dhead_finish (each deferred layer then needs its own workspace).
 *   workspace: n_head*PIT_DSCALE_SLOTS doubles: fp64 accumulators for d c.  They must be ZERO
 *            on entry and are left zero on exit (the finishing kernel drains them with atomic
 *            exchanges, applies d c/d lmda and writes d_head), so a caller allocates and zeroes
 *            them once; no per-call memset.
 *   nbr_complete: 1 = the caller knows that no row's candidate list overflowed (nbr_cnt <= nbr_cap
 *            everywhere, e.g. checked once for a cached fixed-mesh plan): the pass that adds the
 *            overflowed rows to d_values is not launched.  0 = unknown (always correct).
 * d_values and d_head are computed by independent kernels: a caller may issue two calls (one
 * with d_values == NULL, one with d_head == NULL) on different streams to overlap them.
 *   rider    NULL, or the arguments of a pit_mlp_bwd_params call the caller has postponed (the weight-gradient
 *            reductions of the MLP whose d_x is this call's d_out: pit.py:116-121 runs mlp -> attention, so the
 *            backward runs them in this order and nothing downstream needs d_w*).  The call performs it: inside
 *            the attention launch when both are small (latency-bound grids sharing the chip), otherwise as the
 *            separate launches pit_mlp_bwd_params would have made.  Read during the call only. */
typedef struct pit_mlp_params_job {          /* = the argument list of pit_mlp_bwd_params */
    const float* x; long ldx; int rows, n0, n1, n2; const float* h; int out_gelu;
    const float* d_y; long ld_dy;
    float *d_w1, *d_b1, *d_w2, *d_b2; int accumulate; const float* scratch; int math_mode;
} pit_mlp_params_job;
#define PIT_HEAD_ACCUMULATE 1
#define PIT_HEAD_DEFER      2
#define PIT_HEAD_IS_SCALE   4   /* pit_posatt_dhead_finish only: d_head is w.r.t. c, no chain rule */
int pit_posatt_bwd(const float* mesh_out, const float* mesh_in, int mesh_batch, int n_out, int n_in,
                   int space_dim, int metric, float period,
                   const float* values, int batch, int dim, long ld_values, long values_bstride,
                   const float* head, int n_head, int head_is_scale, const float* scale,
                   const float* rowstat, int masked,
                   const float* d_out, long ld_dout, long dout_bstride, int out_col0,
                   float* d_values, long ld_dvalues, long dvalues_bstride, int add_residual,
                   float* d_head, int accumulate_head, double* workspace,
                   const int* nbr_idx, const int* nbr_cnt, int nbr_cap, int nbr_complete,
                   const int* rev_ptr, const int* rev_row, const pit_mlp_params_job* rider,
                   int coord_dims, int math_mode, void* stream);

/* Finishing step of n_layers (<= 32) pit_posatt_bwd calls issued with PIT_HEAD_DEFER, in ONE
 * launch: per layer l drains workspaces[l] (n_heads[l]*PIT_DSCALE_SLOTS doubles, left zero), applies
 * d c / d lmda (heads[l] = lmda, scales[l] = the forward's c or NULL to recompute; with
 * PIT_HEAD_IS_SCALE in flags[l] the result is d c itself) and writes or (PIT_HEAD_ACCUMULATE) adds to
 * d_heads[l].  The arrays are HOST arrays of device pointers / ints, read during the call.
 * rider (ABI 17): NULL, or a postponed pit_mlp_bwd_params the pass has no later launch for (the encoder MLP's, when its fused
 * backward is the pass's last launch): performed by extra workgroups of this launch when small, else by its own launches. */
int pit_posatt_dhead_finish(int n_layers, double* const* workspaces, float* const* d_heads,
                            const float* const* heads, const float* const* scales, const int* n_heads,
                            const int* flags, const pit_mlp_params_job* rider, void* stream);

/* ---- Fused processor blocks (batch-free meshes, small regime; csrc/pit_block.hip) ------------------------------
 * pit.processor (pit.py:114-122) is n_blocks x [posatt.forward (self attention on the latent mesh, locality 1.0:
 * pit.py:102) -> kaiming_mlp -> gelu].  For the batch-free operators the softmax weights of every block depend on
 * (mesh_ltt, lmda_l) only, so ONE launch forms them for all blocks of the step and each block's forward / backward
 * chain becomes one launch (attention as plain MFMA contractions + the MLP phases on the same 16-row slab).
 *
 * pit_block_supported: 1 when this shape takes the fused path (dim = hid_dim = 64, n_head in {1,2}, n_pts a multiple
 * of 256 and <= 2048 (E and Q are n_layers*n_head*n_pts^2 floats each), 256 <= batch*n_pts <= 16384 rows (measured on MI355X: +27 % per step at
 * Darcy batch 16, +19 % at 32, +2 % at 64 over the unfused kernels)); callers fall back to pit_posatt_* + pit_mlp_* otherwise. */
int pit_block_supported(int n_pts, int n_head, int dim, int batch);

/* Weights of n_layers (<= 16) self-attention layers on one batch-free mesh (n_pts, space_dim), nothing masked:
 *   e   (n_layers, n_head, n_pts, n_pts)  exp(-c_lh m[n,j])  - un-normalised and symmetric (S_min = 0: every row holds
 *                                          its own point), so e serves the forward (rows) and d(values) (columns)
 *   q   (same shape, or NULL: forward only) e (m - mbar_n) / rowsum_n : the weights of the d(scale) contraction
 *   inv (n_layers, n_head, n_pts)          1 / rowsum
 *   rowstat (n_layers, n_head, n_pts, 4)   {T = +inf, S_min = 0, 1/rowsum, mbar} exactly as pit_posatt_fwd saves it, so
 *                                          pit_posatt_bwd can serve any layer of the stack as well
 *   scale_out (n_layers, n_head)           the c that was used
 * heads: HOST array of n_layers device pointers (n_head floats each: lmda, or c itself with head_is_scale = 1). */
int pit_block_weights(const float* mesh, int n_pts, int space_dim, int metric, float period, int n_layers,
                      const float* const* heads, int head_is_scale, int n_head, float* e, float* q,
                      float* inv, float* rowstat, float* scale_out, void* stream);

/* One processor block forward: xcat (batch*n_pts, (1+n_head)*dim) holds the block's input in columns [0, dim); the
 * attention output of head h is written to columns [(1+h)*dim, (2+h)*dim) (torch.cat((inputs, conv), -1), pit.py:44)
 * and the block's MLP ((1+n_head)*dim -> dim -> dim, arguments as pit_mlp_fwd) runs on the slab in the same launch.
 * e / inv: this layer's slices of pit_block_weights' outputs. */
int pit_block_fwd(const float* e, const float* inv, int n_pts, int n_head, int dim, int batch, float* xcat,
                  const float* w1, const float* b1, const float* w2, const float* b2, int out_gelu,
                  float* z1, float* h, float* z2, float* y, long ldy, int math_mode, void* stream);

/* One block of the backward chain.  Given d_xcat (gradient of this block's concat tensor, complete):
 *   d(values) = d_xcat[:, 0:dim] + sum_h e_h^T (d_xcat[:, head h] / rowsum_h)   per 16-point slab, then
 *   - with the PREVIOUS block's MLP (w1 (dim, n0_prev), w2 (dim, dim), its saved z1 / z2, out_gelu): the data path of
 *     that MLP's backward on the slab - dZ2, dZ1 into scratch_prev (rows*(2*dim) floats, the layout of
 *     pit_mlp_bwd_data) and dX into d_xprev (rows, n0_prev) (NULL = not needed);
 *   - with w1 == NULL: d(values) is written to d_values (rows, dim).
 *   dscale != NULL: this layer's d(scale) partial sums are ADDED to these accumulators (n_head*PIT_DSCALE_SLOTS
 *     doubles, the PIT_HEAD_DEFER convention of pit_posatt_bwd: drain with pit_posatt_dhead_finish); needs qw and
 *     xcat (the block's concat tensor, whose first dim columns are the attention's values).
 *   rider: as in pit_posatt_bwd (the weight-gradient reductions of this block's own MLP).
 *   rider2: a second postponed job carried the same way - typically a SLICE of rows of a larger one (the decoder MLP's
 *     reductions are row sums: any partition of the rows, each slice with accumulate = 1, gives the same gradient), so
 *     that a job too big for one launch is spread over the block launches of the backward chain. */
int pit_block_bwd(const float* e, const float* inv, const float* qw, int n_pts, int n_head, int dim, int batch,
                  const float* d_xcat, const float* xcat, double* dscale,
                  const float* w1, const float* w2, const float* z1, const float* z2, int out_gelu, int n0_prev,
                  float* d_xprev, long ld_dxprev, float* scratch_prev,
                  float* d_values, long ld_dvalues,
                  const struct pit_mlp_params_job* rider, const struct pit_mlp_params_job* rider2,
                  int math_mode, void* stream);

/* ---- Fused encoder-side and decoder-side launches (batch-free meshes, small regime; round 5; csrc/pit_edge.hip) ------------
 * pit.encoder (pit.py:108-112) = cross attention mesh_in -> mesh_ltt + kaiming_mlp + gelu, pit.decoder (pit.py:124-127) =
 * cross attention mesh_ltt -> mesh_out + kaiming_mlp, each as ONE launch per direction on 16-row slabs of one sample: the
 * attention output of a slab stays in LDS as the A operand of the MLP, and in the backward the gradient of that tensor
 * ((batch, n_out, n_head*dim): 7.6 MB at Darcy b=8) never exists in memory.
 *
 * pit_slab_plan: the lmda-independent part of a masked cross-attention layer on a FIXED mesh pair, per 16-row slab - built once
 * (pit_slab_plan_build) from the candidate lists of pit_plan_fwd, which must be complete (no row's count above cap; report[1]
 * says so) and, for the decoder, have unions of at most PIT_SLAB_UNION_MAX keys (report[0] = the largest union):
 *   m     (n_slabs*16, cap) squared distance of every candidate (rows beyond n_out: unused), the fp32 expression of pit.py:47 /
 *         :192-194 / :251-253; slot (n_slabs*16, cap) the candidate's position in the sorted union of its slab's keys;
 *   keys  (n_slabs, PIT_SLAB_UNION_MAX) that union, nkeys (n_slabs) its size.  stats / rank_w / idx / cnt / cap as pit_posatt_fwd.
 * All pointers are device memory owned by the caller; the struct is passed by pointer and read during the call. */
#define PIT_SLAB_UNION_MAX 64
typedef struct pit_slab_plan {
    int n_out, n_in, cap, n_slabs, umax;
    const float* stats; float rank_w;
    const int* idx; const int* cnt;
    const float* m; const unsigned short* slot; const int* keys; const int* nkeys;
    int rows;       /* rows per slab (ABI 19): 16 for the fused launches and pit_union_att_*, 64 / 128 / 256 for pit_fold_* */
} pit_slab_plan;
/* report: 3 ints, ZERO on entry (largest union, 1 if a list overflowed, longest list).  n_in <= 16384.  rows_per_slab: 16, 64, 128
 * or 256; m / slot hold n_slabs*rows_per_slab rows, n_slabs = ceil(n_out / rows_per_slab). */
int pit_slab_plan_build(const float* mesh_out, const float* mesh_in, int n_out, int n_in, int space_dim, int metric, float period,
                        const int* nbr_idx, const int* nbr_cnt, int cap, int rows_per_slab, float* m, unsigned short* slot,
                        int* keys, int* nkeys, int* report, void* stream);
/* 1 when the fused launches cover this shape: n_head 1 or 2, dim (= the MLP's hidden width) 32 or 64, 256 <= batch*rows <= 2^20 */
int pit_edge_supported(int n_head, int dim, int batch, int rows_per_sample);
/* The up-projection's softmax weights for one step: they depend on (mesh pair, lmda) only, so they are formed ONCE (one workgroup
 * per slab) and every (sample, slab) workgroup of pit_decoder_fwd / _bwd reads its tile as an MFMA operand:
 *   pw (n_slabs, n_head, 16, um) = P (normalised), qw (same shape; NULL: forward only) = P (m - mbar), zeros where a row does not
 *   list a slot; um = 32, 48 or 64 >= max_union (report[0] of pit_slab_plan_build), max_count = report[2]; scale_out (n_head) = c.
 * As a launch of its own, or - the job struct - as extra workgroups of pit_encoder_fwd's launch. */
int pit_decoder_weights(const pit_slab_plan* plan, const float* head, int head_is_scale, int n_head, int max_union, int max_count,
                        float* pw, float* qw, float* scale_out, const float* w1, float* w1f, int dim, void* stream);
/* w1 / w1f (both or neither; ABI 18): the decoder MLP's first weight w1 (dim, n_head*dim), copied to w1f in the order the lanes of
 * pit_decoder_fwd consume it as MFMA operand fragments (the copy is part of the step because w1 changes once per step). */
typedef struct pit_decoder_weights_job {
    const pit_slab_plan* plan; const float* head; int head_is_scale, n_head, max_union, max_count;
    float *pw, *qw, *scale_out;
    const float* w1; float* w1f; int dim;
} pit_decoder_weights_job;
/* pit.decoder forward.  values (batch, n_in, dim) rows ld_values apart; pw from pit_decoder_weights; the MLP is
 * (n_head*dim -> dim -> n2), n2 <= 4, no trailing gelu; y (batch*n_out, n2).  Saved for the backward when given (all or none):
 * x (batch*n_out, n_head*dim) the attention's output, z1 / h (batch*n_out, dim).  zero_buf: zero_n floats cleared on the way (the
 * d_values buffer pit_decoder_bwd adds to).  loss_part != NULL: the slab's partial sums of RelLpNorm(true, y*scale + shift)
 * (utils.py:86-98, p = 1 or 2) as (batch, n2, n_slabs, 2) doubles - plain stores, nothing to zero, the same bits on every run.
 * max_union: report[0] of pit_slab_plan_build: sizes the launch's LDS tiles (as in pit_decoder_weights).
 * w1f: the fragment-order copy of w1 formed by pit_decoder_weights for THIS w1, or NULL (the launch then reads w1's rows). */
int pit_decoder_fwd(const pit_slab_plan* plan, const float* values, long ld_values, long values_bstride, int batch,
                    int n_head, int dim, const float* pw,
                    const float* w1, const float* w1f, const float* b1, const float* w2, const float* b2, int n2,
                    float* x, float* z1, float* h, float* y, float* zero_buf, long zero_n,
                    const float* loss_true, const float* loss_scale, const float* loss_shift, int loss_p, double* loss_part,
                    int max_union, void* stream);
/* pit.decoder backward: d_y (batch*n_out, n2) -> dz1 (batch*n_out, dim: the scratch of the MLP's weight-gradient reductions,
 * pit_mlp_bwd_params with d_y and this scratch), d_values (batch, n_in, dim) ADDED to (fp32 atomics; zero on entry), the layer's
 * d(scale) accumulators (PIT_HEAD_DEFER convention; finish with the scale_out of pit_decoder_weights).  d_y == NULL: the loss
 * inside - d(pred) is formed from loss_part (the forward's partial sums), multiplied by *loss_seed when given, written to d_pred
 * (batch*n_out, n2); *loss_out receives sum_b mean_c ||true - pred'|| / ||true||, norms_out (batch, n2, 2) the norms (may be NULL). */
int pit_decoder_bwd(const pit_slab_plan* plan, const float* values, long ld_values, long values_bstride, int batch,
                    int n_head, int dim, const float* pw, const float* qw,
                    const float* w1, const float* w2, int n2, const float* z1,
                    const float* d_y, long ld_dy, float* dz1, float* d_values, long dvalues_bstride, double* dscale,
                    const float* loss_pred, const float* loss_true, const float* loss_scale, const float* loss_shift,
                    const float* loss_seed, int loss_p, const double* loss_part, float* d_pred, float* loss_out,
                    float* norms_out, int max_union, void* stream);
/* The same union-tile contraction WITHOUT the MLP, for any value width that is a multiple of 64 (masked cross attention on a
 * batch-free mesh pair with a slab plan: the up-projections of Vorticity / Cylinder at hid 256): a workgroup per (sample, slab,
 * 64-column chunk).  out[b, n, h*dim + d] (no input copy, column offset 0) fp32 or - out_bf16 - bf16; the backward reads d_out of the
 * same layout ONCE, ADDS d(values) (fp32 atomics; zero on entry; NULL: not needed) and / or the d(scale) accumulators (PIT_HEAD_DEFER
 * convention; NULL: not needed).  pw / qw from pit_decoder_weights. */
int pit_union_att_supported(int n_head, int dim, int batch, int rows_per_sample);
int pit_union_att_fwd(const pit_slab_plan* plan, const float* values, long ld_values, long values_bstride, int batch,
                      int n_head, int dim, const float* pw, void* out, long ld_out, long out_bstride, int out_bf16,
                      int max_union, void* stream);
int pit_union_att_bwd(const pit_slab_plan* plan, const float* values, long ld_values, long values_bstride, int batch,
                      int n_head, int dim, const float* pw, const float* qw,
                      const void* d_out, long ld_dout, long dout_bstride, int dout_bf16,
                      float* d_values, long ld_dvalues, long dvalues_bstride, double* dscale, int max_union, void* stream);
/* ---- The decoder side with the decoder MLP's first layer folded into the values (round 6; csrc/pit_fold.hip) -------------------
 * pit.decoder (pit.py:124-127) = de(up(values)) with nothing non-linear between the up-projection and de.mlp1, so
 *     Z1 = sum_h P_h (V W1_h^T) + b1,   W1_h = W1[:, h*dim : (h+1)*dim]
 * - the first Linear runs on the n_in LATENT points instead of the n_out output points, and the (batch, n_out, n_head*dim) tensor of
 * pit.py:125 never exists.  vw (batch, n_in, dim*n_head) is HEAD-INTERLEAVED: vw[b, j, n*n_head + h] = sum_d values[b, j, d] W1[n][h*dim + d]
 * = pit_linear_fwd(values, w = W1's memory read as an (n_head*dim, dim) row-major matrix) - no permuted copy of W1, and the weight
 * gradient of that Linear has W1.grad's layout.  For batch-free mesh pairs:
 *   pit_slab_plan_build(..., rows_per_slab = 64 | 128 | 256)  the plan (tall slabs: their unions must fit PIT_SLAB_UNION_MAX)
 *   pit_fold_weights   pw / qw (n_slabs*n_head, rows, um) for one step, as pit_decoder_weights; pw16 / qw16 (optional): the same tiles
 *                      rounded to bf16 - what pit_fold_att_fwd / _bwd take as pw / qw in the bf16 math mode
 *   pit_fold_att_fwd   z[b, n, c] = sum_h sum_j P_h[n][j] vw[b, j, c*n_head + h]        (fp32, or bf16 with PIT_IO_OUT_BF16)
 *   pit_fold_att_bwd   d_vw[b, j, c*n_head + h] += sum_n P_h[n][j] dz[b, n, c] (fp32 atomics, ZERO on entry; NULL: not needed),
 *                      d(scale) accumulators (PIT_HEAD_DEFER convention; NULL: not needed); dz bf16 with PIT_IO_DOUT_BF16
 * math_mode PIT_MATH_FP32: v_mfma_f32_16x16x4_f32 on the fp32 pw / qw; PIT_MATH_BF16: pw / qw ARE pw16 / qw16 (a workgroup reads half the
 * bytes per pass and rounds nothing), the value rows rounded to bf16 in LDS, v_mfma_f32_16x16x32_bf16.
 * dim a multiple of 64, n_head 1 or 2, every tensor below 2 GiB (pit_fold_supported). */
int pit_fold_supported(int n_head, int dim, int batch, int rows_per_sample, int n_in);
int pit_fold_weights(const pit_slab_plan* plan, const float* head, int head_is_scale, int n_head, int max_union, int max_count,
                     float* pw, float* qw, float* scale_out, unsigned short* pw16, unsigned short* qw16, void* stream);
int pit_fold_att_fwd(const pit_slab_plan* plan, const float* vw, long ld_vw, long vw_bstride, int batch, int n_head, int dim,
                     const void* pw, void* z, long ld_z, long z_bstride, int max_union, int math_mode, void* stream);
int pit_fold_att_bwd(const pit_slab_plan* plan, const float* vw, long ld_vw, long vw_bstride, int batch, int n_head, int dim,
                     const void* pw, const void* qw, const void* dz, long ld_dz, long dz_bstride,
                     float* d_vw, long ld_dvw, long dvw_bstride, double* dscale,
                     float* tiles, const int* rev_ptr, const int* rev_ent, int max_union, int math_mode, void* stream);
/* tiles != NULL (with d_vw): NO atomics - every (sample, slab) writes its sums into its own tile of `tiles` (batch, n_slabs,
 * PIT_SLAB_UNION_MAX, dim*n_head floats) and a second launch writes d_vw[b, j, :] = the sum of the tile rows that hold key j, in the
 * fixed order of the CSR lists rev_ptr (n_in + 1) / rev_ent (entries slab*PIT_SLAB_UNION_MAX + slot, grouped by key): d_vw need not
 * be zero on entry and is the same bits on every run. */
/* The rest of `de` (pit.py:21-26 after the first Linear) for out_dim = n2 <= 4 and n1 in {64, 128, 256}:
 *   pit_thin_tail_fwd   y[m][o] = sum_n gelu_erf(z[m][n] + b1[n]) w2[o][n] + b2[o]                      (nothing is saved)
 *   pit_thin_tail_bwd   dz[m][n] = (sum_o d_y[m][o] w2[o][n]) gelu'(z[m][n] + b1[n]), and ADDS d_b1[n] = sum_m dz[m][n],
 *                       d_w2[o][n] = sum_m d_y[m][o] gelu(z[m][n] + b1[n]), d_b2[o] = sum_m d_y[m][o] to the gradients (partial sums meet
 *                       in `scratch` - pit_thin_tail_scratch_floats() floats, ZERO on first use, left zero - and a finishing launch adds
 *                       them: the gradients are complete when the call's launches are).
 * math_mode: PIT_MATH_FP32 - libm's erff; PIT_MATH_BF16 - the polynomial normal CDF of pit_common.h (|error| 1.4e-8); PIT_IO_X_BF16:
 * z (and dz) are bf16 in memory. */
int pit_thin_tail_scratch_floats(void);
int pit_thin_tail_fwd(const void* z, long ldz, int rows, int n1, int n2, const float* b1, const float* w2, const float* b2,
                      float* y, long ldy, int math_mode, void* stream);
int pit_thin_tail_bwd(const void* z, long ldz, int rows, int n1, int n2, const float* b1, const float* w2,
                      const float* d_y, long ld_dy, void* dz, long ld_dz, float* d_b1, float* d_w2, float* d_b2,
                      float* scratch, int math_mode, void* stream);
/* y = x w^T without bias (w (n_out, n_in) contiguous; zero_bias: n_out zeros) and its backward d_x = d_y w (NULL: not needed),
 * d_w (+)= d_y^T x (NULL: not needed; accumulate = 0 zeroes it first) - the GEMM launchers of pit_mlp_fwd / _bwd. */
int pit_linear_fwd(const float* x, long ldx, int rows, int n_in, int n_out, const float* w, const float* zero_bias,
                   float* y, long ldy, int math_mode, void* stream);
int pit_linear_bwd(const float* x, long ldx, int rows, int n_in, int n_out, const float* w, const float* d_y, long ld_dy,
                   float* d_x, long ld_dx, float* d_w, int accumulate, int math_mode, void* stream);

/* pit.encoder forward.  Value channels [0, coord_dims) are the key coordinates (mesh_in; train_darcy.py:51-55), the other
 * value_dim channels come from values (batch, n_in, value_dim); n_head*(coord_dims + value_dim) <= 16.  The MLP is
 * (n_head*(coord_dims+value_dim) -> dim -> dim) followed by gelu; y rows ldy apart (the first columns of the processor's concat
 * buffer).  clear_buf: clear_n floats zeroed on the way (the step's flat gradient buffer).  weights: pit_block_weights' arguments,
 * dec_weights: pit_decoder_weights' - each performed by extra workgroups of this launch (dim 64) or a launch of its own right after. */
struct pit_block_weights_job;
int pit_encoder_fwd(const pit_slab_plan* plan, const float* mesh_in, int space_dim, int coord_dims,
                    const float* values, long ld_values, long values_bstride, int value_dim, int batch,
                    int n_head, int dim, const float* head, int head_is_scale,
                    const float* w1, const float* b1, const float* w2, const float* b2,
                    float* x, float* z1, float* h, float* z2, float* y, long ldy, float* rowstat, float* scale_out,
                    float* clear_buf, long clear_n, const struct pit_block_weights_job* weights,
                    const pit_decoder_weights_job* dec_weights, void* stream);
/* pit.encoder backward: d_y (batch*n_out, dim) rows ld_dy apart -> scratch (dZ1 | dZ2: the layout of pit_mlp_bwd_data) and the
 * down-projection's d(scale) accumulators (required).  The inputs get no gradient (data). */
int pit_encoder_bwd(const pit_slab_plan* plan, const float* mesh_in, int space_dim, int coord_dims,
                    const float* values, long ld_values, long values_bstride, int value_dim, int batch,
                    int n_head, int dim, const float* scale, const float* rowstat,
                    const float* w1, const float* w2, const float* z1, const float* z2,
                    const float* d_y, long ld_dy, float* scratch, double* dscale, void* stream);

/* ---- Batch-free self-attention on PRECOMPUTED weights, large regime (round 4; pit.py:133-144 with locality 1.0) --------
 * The large-regime attention kernels re-formed exp(-c m) in every workgroup (posatt_rows_tiles / posatt_cols_tiles: the weight
 * phase and the MFMA phase of a workgroup do not overlap, 47-51 % MFMA busy).  With the weights of pit_block_weights in
 * memory (n_pts <= 2048) the weight phase is a coalesced copy requested one pass ahead:
 *   pit_posatt_pre_fwd   out[b, n, out_col0 + h*dim + d] = (1/rowsum_h[n]) sum_j e_h[n][j] values[b, j, d]  (e symmetric)
 *   pit_posatt_pre_bwd   d_values[b, j, :] = (add_residual ? d_out[b, j, 0:dim] : 0) + sum_h sum_n (e_h[n][j]/rowsum_h[n]) d_out[b, n, head h]
 *                        and, with workspace != NULL, the layer's d(scale) partial sums ADDED to workspace (PIT_HEAD_DEFER
 *                        convention: drain with pit_posatt_dhead_finish) from q = e (m - mbar)/rowsum.
 * e / q: this layer's (n_head, n_pts, n_pts) slices, rowstat its (n_head, n_pts, 4) slice of pit_block_weights' outputs.
 * pit_posatt_pre_supported: 1 when the shape is in the regime these launches cover (else use pit_posatt_fwd / _bwd). */
int pit_posatt_pre_supported(int n_pts, int n_head, int dim, int batch);
int pit_posatt_pre_fwd(const float* e, const float* rowstat, int n_pts, int n_head, int dim, int batch,
                       const float* values, long ld_values, long values_bstride,
                       float* out, long ld_out, long out_bstride, int out_col0, int copy_inputs, int math_mode, void* stream);
int pit_posatt_pre_bwd(const float* e, const float* q, const float* rowstat, int n_pts, int n_head, int dim, int batch,
                       const float* values, long ld_values, long values_bstride,
                       const float* d_out, long ld_dout, long dout_bstride, int out_col0,
                       float* d_values, long ld_dvalues, long dvalues_bstride, int add_residual,
                       double* workspace, int math_mode, void* stream);

/* ---- Dense self-attention of the processor on bf16 MFMA (round 6; csrc/pit_satt.hip; bf16 math mode) ---------------------------
 * posatt.forward (pit.py:37-57) with locality 1.0 on one mesh (per-sample or batch-free): 1-2 heads, dim 128 / 256, 64..2048 points.
 *   pit_satt_fwd  out[b, n, out_col0 + h*dim + d] = sum_j softmax_j(-c_h m[n, j]) values[b, j, d]; copy_inputs: out[b, n, 0:dim] = values.
 *                 x16: batch*n_pts*dim bf16 (the rounded values; keep it for the backward); rowstat (mesh_batch, n_head, n_pts, 4) =
 *                 {-, 0, 1/rowsum, mbar}, scale_out (n_head) = c, as pit_posatt_fwd.
 *   pit_satt_bwd  d_values[b, j, :] = (add_residual ? d_out[b, j, 0:dim] : 0) + sum_h sum_n P_h[n, j] d_out[b, n, out_col0 + h*dim + :] (NULL: not
 *                 needed); the layer's d(scale) accumulators (PIT_HEAD_DEFER convention; NULL: not needed).  scale = the forward's c;
 *                 g16: scratch of batch*n_head*n_pts*dim bf16.
 *   x16_ready / g16_ready: the caller's x16 / g16 already hold bf16(values) / bf16(d_out_h / rowsum_h) - written by the producing
 *                 pit_mlp_chain_fwd (y16) / pit_mlp_chain_bwd (g16): no prep launch (x16_ready needs copy_inputs = 0).
 *   rider (pit_satt_bwd, optional): the postponed weight-gradient reductions of the MLP whose backward produced d_out (pit_posatt_bwd's
 *                 convention): carried as extra workgroups of the one backward launch, or run as pit_mlp_bwd_params would.
 *   e_tiles (optional, NULL: none): (mesh_batch, n_head, pit_satt_tiles_elems(n_pts)) bf16 - the forward leaves its rounded weights there
 *                 in MFMA A-fragment order and pit_satt_bwd's d(values) (the same, symmetric, matrix) reads them instead of forming
 *                 every weight again.
 * Weights, row sums and the d(scale) reduction are fp32 / fp64; the MFMA operands are bf16 (v_mfma_f32_16x16x32_bf16). */
int pit_satt_supported(int n_pts, int n_head, int dim, int batch, int mesh_batch);
int pit_satt_fwd(const float* mesh, int mesh_batch, int n_pts, int space_dim, int metric, float period,
                 const float* values, long ld_values, long values_bstride, int batch, int dim,
                 const float* head, int n_head, int head_is_scale, unsigned short* x16,
                 float* out, long ld_out, long out_bstride, int out_col0, int copy_inputs,
                 float* rowstat, float* scale_out, unsigned short* e_tiles, int x16_ready, void* stream);
int pit_satt_bwd(const float* mesh, int mesh_batch, int n_pts, int space_dim, int metric, float period,
                 int batch, int dim, const float* scale, int n_head, const float* rowstat,
                 const unsigned short* x16, unsigned short* g16,
                 const float* d_out, long ld_dout, long dout_bstride, int out_col0,
                 float* d_values, long ld_dvalues, long dvalues_bstride, int add_residual, double* dscale,
                 const unsigned short* e_tiles, int g16_ready, const pit_mlp_params_job* rider, void* stream);
long pit_satt_tiles_elems(int n_pts);

/* kaiming_mlp.forward (pit.py:21-26): y = W2 * gelu_erf(W1 x + b1) + b2, optionally
 * followed by the trailing gelu of pit.py:111,121 (out_gelu=1).
 *   x (rows, n0) rows ldx apart; w1 (n1,n0), b1 (n1), w2 (n2,n1), b2 (n2) contiguous;
 *   z1 (rows,n1) and h (rows,n1) are saved for the backward (pre-activation and
 *   gelu(z1)); z2 (rows,n2) is the pre-activation of the trailing gelu (out_gelu=1,
 *   else may be NULL); y (rows, n2) rows ldy apart. */
int pit_mlp_fwd(const float* x, long ldx, int rows, int n0, int n1, int n2,
                const float* w1, const float* b1, const float* w2, const float* b2, int out_gelu,
                float* z1, float* h, float* z2, float* y, long ldy, int math_mode, void* stream);

/* One-launch kaiming_mlp + gelu chains of the bf16 math mode for hid 128 / 256 on a few thousand rows (round 6; csrc/pit_chain.hip):
 * pit_mlp_fwd(..., out_gelu = 1) / pit_mlp_bwd_data as ONE launch each on 32-row slabs, the weights streamed through LDS as bf16
 * panels.  w1_bf16 (n1, n0) / w2_bf16 (n1, n1): row-major bf16 copies of the weights (pit_cast_bf16_multi); n2 == n1 in {128, 256},
 * n0 a multiple of n1 (the processor's (1 + H) hid), <= 1024.  z1 / h / z2 (rows, n1) fp32 as pit_mlp_fwd saves them; y rows ldy apart.
 * Backward: scratch = dZ1 | dZ2 (rows*n1 floats each: what pit_mlp_bwd_params reads), d_x (rows, n0) rows ld_dx apart or NULL.
 * gelu / gelu' are the polynomial CDF of the bf16 mode (pit_thin_tail_*). */
int pit_mlp_chain_supported(int rows, int n0, int n1, int n2);
int pit_mlp_chain_fwd(const float* x, long ldx, int rows, int n0, int n1, const unsigned short* w1_bf16, const float* b1,
                      const unsigned short* w2_bf16, const float* b2, float* z1, float* h, float* z2, float* y, long ldy,
                      unsigned short* y16 /* optional: bf16(y), (rows, n1) - the x16 of the next layer's pit_satt_fwd */, void* stream);
int pit_mlp_chain_bwd(int rows, int n0, int n1, const unsigned short* w1_bf16, const unsigned short* w2_bf16,
                      const float* z1, const float* z2, const float* d_y, long ld_dy, float* d_x, long ld_dx, float* scratch,
                      unsigned short* g16 /* optional: the g16 of the producing layer's pit_satt_bwd: bf16(d_x[:, (1+h) n1 + :] / rowsum_h), (rows / pts, n0 / n1 - 1, pts, n1) */,
                      const float* rowstat /* that layer's */, int pts, int mesh_batch, void* stream);
/* dst[i] = bf16(src[i]) (RNE) for n <= 32 tensors of count[i] floats (multiples of 4; 16-byte aligned sources): one launch. */
int pit_cast_bf16_multi(int n, const float* const* src, unsigned short* const* dst, const long* count, void* stream);

/* Backward of pit_mlp_fwd.  d_y (rows,n2) rows ld_dy apart (ld_dy == n2 when out_gelu).
 * d_x (rows,n0) rows ld_dx apart (NULL = not needed).  d_w1,d_b1,d_w2,d_b2 receive the
 * parameter gradients through fp32 atomics over row slabs: accumulate=0 zeroes them first
 * (plain gradient), accumulate=1 adds into their current contents (e.g. the parameters'
 * .grad inside a flat gradient buffer - no memset, no separate add).
 * scratch: rows*(n1+n2) floats. */
int pit_mlp_bwd(const float* x, long ldx, int rows, int n0, int n1, int n2,
                const float* w1, const float* w2, const float* z1, const float* h, const float* z2,
                int out_gelu, const float* d_y, long ld_dy,
                float* d_x, long ld_dx, float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                int accumulate, float* scratch, int math_mode, void* stream);

/* The two halves of pit_mlp_bwd as separate entry points, so a caller can put the
 * parameter-gradient GEMMs (which nothing downstream in the backward pass depends on) on a
 * second stream: _data computes dZ1 (and dZ2 when out_gelu) into scratch and d_x; _params
 * consumes scratch and must be ordered after _data (event / same stream). */
int pit_mlp_bwd_data(int rows, int n0, int n1, int n2, const float* w1, const float* w2,
                     const float* z1, const float* z2, int out_gelu, const float* d_y, long ld_dy,
                     float* d_x, long ld_dx, float* scratch, int math_mode, void* stream);
int pit_mlp_bwd_params(const float* x, long ldx, int rows, int n0, int n1, int n2, const float* h,
                       int out_gelu, const float* d_y, long ld_dy,
                       float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                       int accumulate, const float* scratch, int math_mode, void* stream);
/* 1 if postponing pit_mlp_bwd_params of this shape and handing it to the next pit_posatt_bwd as `rider` costs
 * nothing when the attention call cannot merge it (i.e. pit_mlp_bwd would have issued _data and _params as
 * separate launches anyway), else 0 (pit_mlp_bwd merges d_x with the reductions: keep the single call). */
int pit_mlp_bwd_params_deferrable(int rows, int n0, int n1, int n2, int out_gelu, long ld_dy);

/* RelLpNorm (utils.py:80-98): loss = sum_b mean_c ||true - pred'||_p / ||true||_p with norms
 * over the point axis of (batch, npts, nch) contiguous tensors, pred' = pred*scale + shift when
 * the optional per-pixel (npts, nch) affine of PixelWiseNormalization.denormalize
 * (utils.py:25-34, train_darcy.py:129) is given (both NULL = identity).
 * norms (batch, nch, 2) = {||true-pred'||_p, ||true||_p} is saved for the backward; loss is one
 * float.  workspace: PIT_REL_LP_WS_FLOATS(batch, nch) floats, 16-byte aligned (loss accumulator + arrival
 * counter, and per (sample, channel) two fp64 partial sums + a counter: a series is split over several
 * workgroups), zero before the first call and left zero by every call.  grad_loss: device pointer to the upstream scalar (NULL = 1).
 * The backward writes d loss/d pred and/or d loss/d true (either may be NULL): the scripts pass
 * the model output as `pred` (train_darcy.py:130) or as `true` (train_vorticity.py:124). */
#define PIT_REL_LP_WS_FLOATS(batch, nch) (4 + 5 * (batch) * (nch))
int pit_rel_lp_loss_fwd(const float* tru, const float* pred, const float* pred_scale,
                        const float* pred_shift, int batch, int npts, int nch, int p,
                        float* norms, float* loss, float* workspace, void* stream);
int pit_rel_lp_loss_bwd(const float* tru, const float* pred, const float* pred_scale,
                        const float* pred_shift, int batch, int npts, int nch, int p,
                        const float* norms, const float* grad_loss, float* d_pred, float* d_true, void* stream);

/* pit_rel_lp_loss_fwd that ALSO writes, in the same launch, the gradients for an upstream gradient
 * of 1 - d_pred_unit and/or d_true_unit, (batch,npts,nch), NULL = not wanted - and clears clear_n
 * floats at clear_buf (e.g. the flat gradient accumulators of the step whose backward pass follows;
 * clear_n = 0: nothing).  A training step seeded with d loss = 1 then needs neither the loss
 * backward launch nor a memset. */
int pit_rel_lp_loss_fwd_grad(const float* tru, const float* pred, const float* pred_scale,
                             const float* pred_shift, int batch, int npts, int nch, int p,
                             float* norms, float* loss, float* workspace, float* d_pred_unit,
                             float* d_true_unit, float* clear_buf, long clear_n, void* stream);

/* RelMaxNorm (utils.py:59-77), the evaluation metric of train_burgers.py:80 / train_sod.py:84 /
 * train_elasticity.py:118 / train_naca.py:132:  out = sum_b mean_c max_l|true - pred| / max_l|true|
 * over (batch, npts, nch) contiguous tensors.  Forward only (the scripts never differentiate it).
 * workspace: 2 doubles (accumulator + arrival counter), zero before the first call, left zero. */
int pit_rel_max_norm(const float* tru, const float* pred, int batch, int npts, int nch, float* out,
                     double* workspace, void* stream);

/* nn.InstanceNorm1d over the point axis as train_vorticity.py:43,56,59 applies it
 * (norm(x.permute(0,2,1)).permute(0,2,1); no affine, no running statistics, biased variance),
 * computed directly on the (batch, points, channels) layout:
 *   y[b,l,c] = (x[b,l,c] - mean_l x[b,:,c]) * rstd[b,c],  rstd = 1/sqrt(var_l + eps).
 * x rows ldx apart, samples x_bstride apart; y (batch,npts,nch) and rstd (batch,nch) contiguous. */
int pit_instance_norm_fwd(const float* x, long ldx, long x_bstride, int batch, int npts, int nch, float eps,
                          float* y, float* rstd, void* stream);
/* d_x = rstd * (d_y - mean_l d_y - y * mean_l(d_y * y)); all (batch,npts,nch) contiguous. */
int pit_instance_norm_bwd(const float* d_y, const float* y, const float* rstd, int batch, int npts, int nch,
                          float* d_x, void* stream);

/* torch.optim.Adam step (no amsgrad) over FLAT fp32 buffers of n elements, learning rate following
 * CosineAnnealingLR(T_max=cosine_t_max, eta_min) when cosine_t_max > 0 (train_darcy.py:115-116),
 * else constant lr0.  `step` (device int64, starts at 0) is incremented here; `scalars` is 4
 * floats of device scratch, ZERO before the first call (slot 3 is an arrival ticket; slots 0-2
 * receive lr_t and the bias corrections of the step taken).  With zero_grads=1 the gradients are
 * cleared as they are consumed (the next backward pass accumulates into zeros: no memset launch).
 * One launch, no host sync: replayable from a hipGraph. */
int pit_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, long n,
                  long long* step, float lr0, float eta_min, int cosine_t_max, float beta1, float beta2,
                  float eps, float weight_decay, int zero_grads, float* scalars, void* stream);

/* Layout probe used by the tests: D = A(32x8) * B(8x32) through the same
 * v_mfma_f32_32x32x2_f32 fragment maps the kernels use. */
int pit_debug_mfma_tile(const float* a, const float* b, float* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PIT_HIP_H */
