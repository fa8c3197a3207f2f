/*
 * svnet_hip.h — C ABI of libsvnet_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * SVNet hot path.  The reference (hellozhuo/svnet) is pure Python/PyTorch and has no FFI of its
 * own; each entry point below names the reference op chain it replaces (file:line relative to the
 * reference repository root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated otherwise; tensors are dense row-major fp32,
 *     indices int64; "rows" M is the product of all leading dimensions
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library never allocates or
 *     frees device memory and keeps no device state between calls
 *   - `stream` is a hipStream_t (passed as void*); calls are asynchronous and ordered on it
 *   - return value: 0 = ok, <0 = error (SVNET_E_*); svnet_last_error() gives a message for the
 *     calling thread.  Nothing throws across the ABI.
 */
#ifndef SVNET_HIP_H
#define SVNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVNET_OK 0
#define SVNET_E_ARG (-1)       /* null pointer / negative size / inconsistent arguments */
#define SVNET_E_UNSUPPORTED (-2) /* shape outside what the kernels were built for */
#define SVNET_E_WORKSPACE (-3) /* workspace too small */
#define SVNET_E_LAUNCH (-4)    /* HIP reported an error at launch */

/* Grid-wide sums are spread over slices (workgroup w adds to slice w % SVNET_RED_SLICES): a thousand float / double atomics onto one
 * address are served one after the other at the memory side and set the pace of the reduction kernels.                            */
#define SVNET_RED_SLICES 16
/* Length (elements) of a SLICED accumulator of L sums (svnet_colstats_f64, svnet_bn_act_bwd_reduce_f32, svnet_vbn_bwd_reduce_f32,
 * svnet_bn_pool_bwd_f32, the col_sums of svnet_binlinear_i8_fwd_f32): [L totals | SVNET_RED_SLICES x L slices | 2 spare], zero-filled by
 * the caller.  The REDUCING entry point only fills the slices; the totals (first L elements) are written by the entry point that
 * CONSUMES the accumulator (svnet_bn_finalize_f32, svnet_bn_act_bwd_apply_f32, svnet_vbn_bwd_apply_f32, the apply pass of
 * svnet_bn_pool_bwd_f32; svnet_vbn_fwd_stats_f32 reads the slices and writes no totals) or by svnet_slices_sum_f32 / _f64 - i.e. the
 * slices are handed from their adders to their reader across a kernel boundary of the stream, never inside a launch.               */
#define SVNET_SLICED_LEN(L) ((SVNET_RED_SLICES + 1) * (L) + 2)
/* buf[0:L] = sum over the slices of a sliced accumulator (for a caller that reads the totals itself). */
int svnet_slices_sum_f32(float* buf, int64_t L, void* stream);
int svnet_slices_sum_f64(double* buf, int64_t L, void* stream);

/* ABI version = 100 * round-of-change + serial.  It changes whenever an entry point gains / loses an argument or a caller-owned buffer
 * changes its required length (200: sliced accumulators, SVNET_SLICED_LEN; 400: this header; 401: the totals of a sliced accumulator are
 * written by its consumer, svnet_slices_sum_*; 402: GX of svnet_v2s_bwd_*, gw of svnet_xyzblock_bwd_f32 and col_sum of svnet_gemm_f32 are sliced accumulators; 403: svnet_binweight_grad_f32 takes sliced inputs).  svnet_version() returns the value the
 * library was BUILT with: a caller compiled against another header must refuse to run (svnet_amd/_lib.py does).                   */
#define SVNET_ABI_VERSION 417
int svnet_version(void);
const char* svnet_last_error(void);

/* A gate MLP run by extra workgroups of a coefficient launch (svnet_edgeblock_coeffs_f32, svnet_xyzblock_coeffs_f32,
 * svnet_edgeblock_bwd_coeffs_f32: their last argument, NULL = none): the arguments of svnet_gate_mlp_fwd_f32 / svnet_gate_mlp_bwd_f32
 * below.  The gate and the coefficients only share their inputs; as two launches they were two latency-bound links of every fused
 * layer's critical path. */
typedef struct svnet_gate_fwd_job {
    const float* gin; const double* gin_f64; float* gin_out; float in_scale; const float* W0; const float* W2;
    int64_t B, Cin, H, Ov; float* h; float* gate;
    const float* rows; int64_t R;   /* optional third source of the MLP's input (gin and gin_f64 NULL): gin_out[b,:] = mean over the R
                                       rows of cloud b of rows [B*R, Cin] (sv_layers.py:179: s.mean over the points), kept for the backward */
} svnet_gate_fwd_job;
typedef struct svnet_gate_bwd_job {
    const float* dgate; const float* gate; const float* h; const float* gin; float in_scale; const float* W0; const float* W2;
    int64_t B, Cin, H, Ov; float out_scale; float* dgin; float* dW0; float* dW2;
} svnet_gate_bwd_job;

/* ------------------------------------------------------------------ k-NN  (models/utils/sv_util.py:19-25, knn)
 * x is addressed as x[b*sb + n*sn + c*sc] (the [B,C,N] tensor the reference passes, any strides).
 * xx_mode: 0 = ||x||^2 summed the way ATen reduces an OUTER dim (contiguous [B,C,N]),
 *          1 = the way ATen reduces the CONTIGUOUS dim (transposed view of [B,N,C])   
 * idx_out: [B,N,k] int64, cloud-local neighbour ids, nearest first (self first), ties -> lowest id.
 * Bit-exact against the reference's torch-CPU result for C <= 384, N <= 4096, k <= 64.           */
size_t svnet_knn_workspace_bytes(int64_t B, int64_t N, int64_t C);
int svnet_knn_f32(const float* x, int64_t B, int64_t N, int64_t C, int64_t sb, int64_t sn, int64_t sc,
                  int xx_mode, int k, int64_t* idx_out, void* workspace, size_t workspace_bytes, void* stream);
/* The feature-space graph of get_graph_feature_sv (sv_util.py:100-101): k-NN on the rows cat[s, v.view(B,N,3Cv)] read from
 * s [B,N,Cs] and v [B,N,Cv3] where they lie (no concatenated copy); same arithmetic as svnet_knn_f32 with xx_mode 1.
 * Workspace: svnet_knn_workspace_bytes(B, N, Cs + Cv3).                                               */
int svnet_knn_sv_f32(const float* s, int64_t Cs, const float* v, int64_t Cv3, int64_t B, int64_t N, int k, int64_t* idx_out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* The candidate table of a k-NN call prepared by the PRODUCER of the features (the apply pass of the previous fused level): the k-NN is
 * on the forward's critical path and its first kernel only re-reads, squares and transposes what that pass has just written
 * (sv_dgcnn_cls.py:55-65: svpool(conv_l(..)) -> get_graph_feature_sv -> knn).  svnet_knn_table_fusable: 1 when the table of (B, N, C)
 * is the channel-major one the producers write and N % 32 == 0.  svnet_edgeblock_apply_knn_f32 / svnet_xyzblock_apply_knn_f32: the apply
 * pass of svnet_edgeblock_apply_f32 / svnet_xyzblock_apply_f32 (same arguments, bit-identical s_out / v_out / slices) that also fills
 * knn_workspace (svnet_knn_workspace_bytes(P / N, N, Os + 3 Ov)) with the table and the ||x||^2 of the rows cat[s, v.view(3 Ov)]
 * (sv_util.py:100; ATen's contiguous-row recipe, bit for bit what svnet_knn_sv_f32 computes).  svnet_knn_from_table_f32: the k-NN
 * proper on such a workspace - idx_out as svnet_knn_f32.                                                                            */
int svnet_knn_table_fusable(int64_t B, int64_t N, int64_t C);
int svnet_knn_from_table_f32(const void* workspace, size_t workspace_bytes, int64_t B, int64_t N, int64_t C, int k, int64_t* idx_out,
                             void* stream);
int svnet_edgeblock_apply_knn_f32(const int32_t* n_max, const int32_t* n_min, const float* mv, const float* mvn, const float* coef,
                                  const float* gate, int64_t P, int64_t N, int64_t Os, int64_t Ov, float slope, float* s_out, float* v_out,
                                  float* s_cat, int64_t s_ld, float* v_cat, int64_t v_ld, void* knn_workspace, size_t knn_workspace_bytes,
                                  void* stream);
int svnet_xyzblock_apply_knn_f32(const float* y_max, const float* y_min, const float* mv, const float* mvn, const float* coef,
                                 const float* gate, int64_t P, int64_t N, int64_t Os, int64_t Ov, float slope, float* s_out, float* v_out,
                                 float* s_cat, int64_t s_ld, float* v_cat, int64_t v_ld, void* knn_workspace, size_t knn_workspace_bytes,
                                 void* stream);

/* ------------------------------------------------------------------ edge features from xyz
 * (sv_util.py:28-62 get_graph_feature, :64-88 get_graph_feature_cross)
 * x: contiguous [B,3m,N] (channel = m*3+d).  idx: [B,N,k] cloud-local.
 * mode 0: out[b,n,j,d,:] = [x_j-x_i (m), x_i (m)]            -> [B,N,k,3,2m]
 * mode 1: (first=True)    [x_j-x_i, mean_j(x_j-x_i)]          -> [B,N,k,3,2m]
 * mode 2: (cross)         [x_j-x_i, x_i, x_j x x_i]           -> [B,N,k,3,3m]                   */
int svnet_edge_xyz_f32(const float* x, const int64_t* idx, int64_t B, int64_t N, int64_t k, int64_t m, int mode,
                       float* out, void* stream);

/* ------------------------------------------------------------------ edge features from (s,v) tables
 * (sv_util.py:90-116 get_graph_feature_sv; backward = autograd of :106-114, an index_put accumulate)
 * table: [B*N, G, F]; out: [B*N*k, G, 2F] with out[e,g,:F] = t[j,g,:]-t[i,g,:], out[e,g,F:] = t[i,g,:].
 * (G,F) = (1,Cs) for scalars, (3,Cv) for vectors.  idx_is_global: idx already holds b*N+j row ids. */
int svnet_edge_diffcat_fwd_f32(const float* table, const int64_t* idx, int idx_is_global, int64_t B, int64_t N,
                               int64_t k, int64_t G, int64_t F, float* out, void* stream);
/* d_table must be zero-filled by the caller; gradients are accumulated with float atomics.          */
int svnet_edge_diffcat_bwd_f32(const float* d_out, const int64_t* idx, int idx_is_global, int64_t B, int64_t N,
                               int64_t k, int64_t G, int64_t F, float* d_table, void* stream);

/* ------------------------------------------------------------------ generic fp32 GEMM  C[M,N] = epi(sum_k A(i,k) B(k,j))
 * Used for every dense contraction of the path: F.linear of sv_layers.py:31,49 and the autograd
 * products of its backward.  A(i,k) = A[i*a_rs + k*a_cs] * a_scale[k], B(k,j) = B[k*b_rs + j*b_cs],
 * C(i,j) = C[i*ldc + j*c_cs].
 * Bit-planes are "row-sliced": word [(r >> 6) * W + c] holds bit (r & 63) of column c for rows 64*(r>>6)..+63
 * (W = number of columns; rows beyond the tensor are 0).  They appear in two places:
 *   - a_sign/a_nz != NULL: the A operand is ternary, A(i,k) = nz ? (sign ? +1 : -1) : 0, sliced over the
 *     REDUCTION index k with W = M columns i (this is x_b^T of a binarized layer: weight-gradient products);
 *   - mask: STE mask of the output, sliced over the output rows i with W = N columns.
 * b_exact != 0 promises that every B value is exactly representable in bf16 (sign weights: -1, 0, +1), which
 * lets the tall-and-skinny product run on bf16 MFMA with an exact 3-way split of A.
 * Epilogue, in this order: *alpha, *col_scale[j], +bias[j], *mask(i,j), then col_sum[j] += sum_i C(i,j) (float atomics into
 * the slices of a SLICED accumulator of N floats, SVNET_SLICED_LEN(N): caller zero-fills, totals by svnet_slices_sum_f32), then
 * store (or accumulate when accumulate != 0).                                                                 */
typedef struct svnet_gemm_desc {
    int64_t M, N, K;
    const float* A; int64_t a_rs, a_cs;
    const float* a_scale;
    const uint64_t* a_sign; const uint64_t* a_nz;
    const float* B; int64_t b_rs, b_cs;
    int b_exact;
    float* C; int64_t ldc, c_cs;
    float alpha;
    const float* col_scale;
    const float* bias;
    const uint64_t* mask;
    float* col_sum;
    int split_k;       /* 0 = choose automatically (vector-ALU kernel only) */
    int accumulate;    /* C += result */
    uint32_t tern_tile_mask; /* ternary A only: bit t set = rows [32t, 32t+32) of A^T (output rows i) may be non-zero; 0 = all.
                                Cleared tiles are skipped: their outputs are left untouched (use with accumulate on zeros)       */
    void* workspace;   /* optional scratch; with b_exact, >= svnet_gemm_workspace_bytes(N, K) lets the MFMA path pack B once as */
    size_t workspace_bytes; /* bf16 [N][K] (k contiguous) so that its LDS staging is plain 16-byte copies                       */
} svnet_gemm_desc;
size_t svnet_gemm_workspace_bytes(int64_t N, int64_t K);
int svnet_gemm_f32(const svnet_gemm_desc* desc, void* stream);

/* ------------------------------------------------------------------ binarized layers (sv_layers.py:20-53 Linear, :55-78 Conv1d)
 * Weight preparation (sign(W) with sign(0)=0, clamp/STE mask |W|<=1.2):
 *   w_sign/w_nz: [O, Kw] uint64 bit-planes (Kw = ceil(K/64)), w_b: [O,K] fp32 in {-1,0,1},
 *   w_eff: [O,K] = scale[o]*w_b (NULL allowed), any output pointer may be NULL.                    */
int svnet_binweight_prepare_f32(const float* W, const float* scale, int64_t O, int64_t K, uint64_t* w_sign,
                                uint64_t* w_nz, float* w_b, float* w_eff, void* stream);
/* y[m,o] = scale[o] * sum_k sgn(x[m,k]+beta[k]) * sgn(W[o,k])  (+bias[o]) by XNOR/popcount on ternary
 * bit-planes.  x row stride ldx.  Optional outputs (NULL to skip), each a row-sliced plane [ceil(M/64), K]
 * uint64 (layout: see svnet_gemm_desc):  x_sign, x_nz (sign / non-zero planes of the binarized input),
 * x_ste (|x+beta| <= 1.2).  3 bits per input element replace the saved fp32 input in training.        */
int svnet_binlinear_fwd_f32(const float* x, int64_t ldx, const float* beta, const uint64_t* w_sign,
                            const uint64_t* w_nz, const float* scale, const float* bias, int64_t M, int64_t K,
                            int64_t O, float* y, uint64_t* x_sign, uint64_t* x_nz, uint64_t* x_ste, void* stream);
/* The same layer on the matrix cores, for many rows (v_mfma_i32_32x32x32_i8 on int8 ternary operands, exact int32 counts: outputs
 * and saved planes identical to svnet_binlinear_fwd_f32).  w_i8 = svnet_binweight_pack_i8(W): sign(W) as int8 [O][128*ceil(K/128)],
 * every 128-column chunk in the kernel's reduction order, zero padded (svnet_binweight_i8_bytes bytes).                           */
size_t svnet_binweight_i8_bytes(int64_t O, int64_t K);
int svnet_binweight_pack_i8(const float* W, int64_t O, int64_t K, int8_t* w_i8, void* stream);
/* col_sums (NULL to skip): a sliced accumulator of 2*O doubles (SVNET_SLICED_LEN(2*O), zero-filled): sum_m y[m,o] and sum_m y[m,o]^2 - the batch statistics of the BatchNorm that follows
 * (sv_layers.py:189), from the exact integer counts, so that the output is not read again for them (feed svnet_bn_finalize_f32).      */
int svnet_binlinear_i8_fwd_f32(const float* x, int64_t ldx, const float* beta, const int8_t* w_i8, const float* scale,
                               const float* bias, int64_t M, int64_t K, int64_t O, float* y, uint64_t* x_sign, uint64_t* x_nz,
                               uint64_t* x_ste, double* col_sums, void* stream);
/* The same layer when part of its input is CONSTANT over the rows of a cloud (the broadcast half of a concatenation:
 * sv_dgcnn_partseg.py:115-118 `repeat` + `cat`, sv_pointnet_cls.py:43-52 `expand_as` + `svcat`): x [M, K] holds only the per-point
 * columns, cloud_n [M / rows_per_cloud, O] the integer counts (as floats) of the per-cloud columns - svnet_binlinear_fwd_f32 over the
 * B per-cloud rows with scale = 1 -, and y[m,o] = scale[o] * (count(x[m]) + cloud_n[m / rows_per_cloud, o]) (+ bias).  The sum of the two
 * counts is the count over the full row: outputs and col_sums identical to the product over the materialised concatenation.      */
int svnet_binlinear_i8_cloud_fwd_f32(const float* x, int64_t ldx, const float* beta, const int8_t* w_i8, const float* scale,
                                     const float* bias, int64_t M, int64_t K, int64_t O, float* y, uint64_t* x_sign, uint64_t* x_nz,
                                     uint64_t* x_ste, double* col_sums, const float* cloud_n, int64_t rows_per_cloud, void* stream);
/* Chain rule from GX[o,k] = sum_m g[m,o] x_eff[m,k] to the parameters of a bw layer (App. C2):
 *   dW[o,k] = scale[o]*GX[o,k]*[|W[o,k]|<=1.2],  dscale[o] = sum_k w_b[o,k]*GX[o,k];  accumulate != 0 adds to dW/dscale. */
/* gx_sliced != 0: GX is a sliced accumulator of O*K floats (SVNET_SLICED_LEN) whose slices are added up on the way in.
 * sum_buf / sum_len (NULL / 0 to skip): another sliced accumulator whose totals this launch leaves in its first sum_len elements
 * (the dL/dbeta column sums of the same layer's input-gradient product: saves a svnet_slices_sum_f32 launch).                       */
int svnet_binweight_grad_f32(const float* GX, const float* W, const float* scale, int64_t O, int64_t K,
                             float* dW, float* dscale, int accumulate, int gx_sliced, float* sum_buf, int64_t sum_len, void* stream);

/* ------------------------------------------------------------------ fused edge block (tier 2)
 * One pass over the edges of a BINARIZED edge layer, never materialising an edge tensor:
 *   get_graph_feature_sv (sv_util.py:90-116) -> SVBlock (sv_layers.py:172-196) -> svpool (sv_util.py:118-132).
 * Inputs are point tables: s [P,Cs], v [P,3,Cv], idx [P,k] (cloud-local), and two small per-point products that
 * the linear maps on v_e = [v_j - v_i, v_i] collapse to:
 *   zz [P,3,6]    = [Zp | Zq],  Zp = v . (scale_z*sign(Wz[:, :Cv]))^T,  Zq = v . (scale_z*sign(Wz[:, Cv:]))^T   (v2s frame)
 *   ut [P,3,2Ov]  = [U  | T ],  U  = v . (scale2*sign(W2[:, :Cv]))^T,   T  = v . (scale2*sign(W2[:, Cv:]))^T    (linear2)
 * linear1's sign planes / beta are permuted once into the kernel's bit order (5 words: s_j-s_i | s_i | s_v[:,0] |
 * s_v[:,1] | s_v[:,2]) by svnet_edgeblock_prepare_f32.  Limits: Cs <= 64, 2*Cv <= 64, Os <= 128, Ov <= 64, k <= 64 (backward: 2 <= k).
 * Per-point outputs: n_max/n_min [P,Os] (extreme integer popcount sums over the k neighbours) with their slots,
 * mv/mvn [P,3,Ov] (mean_k v', mean_k v'/|v'|); batch statistics as exact integer sums stat_n [SVNET_RED_SLICES][2*Os] (sum n,
 * sum n^2; in slices that svnet_edgeblock_coeffs_f32 adds up) and fp64 sums stat_v [SVNET_RED_SLICES][2*Ov]; gate_sum [B,2Cs] = sum over the cloud's edges of [s_j-s_i, s_i] (caller zero-fills
 * stat_*, gate_sum; stat_* may both be NULL in eval mode).                                                      */
typedef struct svnet_edgeblock_desc {
    int64_t B, N, k;
    int Cs, Cv, Os, Ov;
    const float* s; const float* v; const int64_t* idx;
    const float* zz; const float* ut;
    const uint64_t* w_sign; const uint64_t* w_nz; const float* beta_perm;
    int32_t* n_max; int32_t* n_min; uint8_t* slot_max; uint8_t* slot_min;
    float* mv; float* mvn;
    int64_t* stat_n; double* stat_v; double* gate_sum;
    /* kept for the backward (both or none): n16 [E,Os] = the integer popcount sum of every edge row, planes [E,3,5] = the
     * sign | non-zero | STE (|x+beta| <= 1.2) bit planes of the binarized edge feature in the fused bit order            */
    int16_t* n16; uint64_t* planes;
    /* (may be NULL) the flag svnet_edgeblock_wbt_bf16 leaves: 1 = no weight of the layer is exactly 0, i.e. sign(W1) is +-1 everywhere and the
     * forward kernels may skip the weights' non-zero plane (same integer counts, 40 % fewer popcount instructions)                     */
    const uint32_t* w_dense;
} svnet_edgeblock_desc;
int svnet_edgeblock_prepare_f32(const float* W, const float* beta, int64_t Os, int64_t Cs, int64_t Cv, uint64_t* w_sign,
                                uint64_t* w_nz, float* beta_perm /*[5*64]*/, void* stream);
/* The binarized weights of the two per-point products in one table: wv [2Ov+6, Cv] = [sign(W2[:, :Cv]) ; sign(W2[:, Cv:]) ;
 * sign(Wz[:, :Cv]) ; sign(Wz[:, Cv:])], scv [2Ov+6] = [scale2, scale2, scalez, scalez]  (ut = v.wv[:2Ov]^T*scv, zz = v.wv[2Ov:]^T*scv). */
int svnet_edgeblock_prepare_vec_f32(const float* W2, const float* scale2, const float* Wz, const float* scalez, int64_t Ov,
                                    int64_t Cv, float* wv, float* scv, void* stream);
int svnet_edgeblock_fwd_f32(const svnet_edgeblock_desc* desc, void* stream);
/* coef [4*Os + 4*Ov] = [A1 | B1 | mean_y | invstd_y | Av | Bv | mean_n' | invstd_n']: BatchNorm folded into
 * y = A1*n + B1 and q = Av + Bv/n'; training != 0 uses the batch sums (E = B*N*k edges) and updates running_*.   */
int svnet_edgeblock_coeffs_f32(const int64_t* stat_n, const double* stat_v, int64_t E, int64_t Os, int64_t Ov,
                               const float* scale1, const float* gamma1, const float* beta1, float* running_mean1,
                               float* running_var1, const float* gamma2, const float* beta2, float* running_mean2,
                               float* running_var2, int training, float eps, float momentum, float* coef,
                               int64_t* num_batches_tracked1, int64_t* num_batches_tracked2 /* += 1 when training; may be NULL */,
                               const svnet_gate_fwd_job* gate_job /* may be NULL */, void* stream);
/* s_out[P,Os] = leaky_relu(A1*(A1>=0 ? n_max : n_min) + B1);  v_out[P,3,Ov] = gate[b]*(Av*mv + Bv*mvn).
 * s_cat / v_cat (each may be NULL): the same values written a second time as a column slice of wider rows - s_cat[p*s_ld + o],
 * v_cat[(p*3 + d)*v_ld + c] - i.e. straight into svcat([x1, x2, x3, x4]) of sv_dgcnn_cls.py:68 (no concatenation pass).        */
int svnet_edgeblock_apply_f32(const int32_t* n_max, const int32_t* n_min, const float* mv, const float* mvn,
                              const float* coef, const float* gate, int64_t P, int64_t N, int64_t Os, int64_t Ov,
                              float slope, float* s_out, float* v_out, float* s_cat, int64_t s_ld, float* v_cat, int64_t v_ld,
                              void* stream);

/* Everything between a fused level's edge pass and the next level's k-NN in ONE launch: the coefficients (svnet_*_coeffs_f32, gate MLP
 * included), the apply pass (svnet_*_apply_f32) and, with knn_workspace, the next k-NN's table (svnet_*_apply_knn_f32).  Three dependent
 * launches of latency-bound work per level sat on the forward's critical path; here every workgroup (32 points of one cloud) derives
 * the ~2 (Os + Ov) coefficients from the statistic slices and its cloud's gate itself - the same expressions as the separate kernels,
 * bit-identical outputs - and workgroup 0 alone writes coef, the running statistics and the counters.  Needs N % 32 == 0, Os <= 256,
 * Ov <= 256 (svnet_block_tail_supported); stat1 = the scalar path's sums: int64 slices of (sum n, sum n^2) for the edge block
 * (scale1 required), fp64 slices of (sum y, sum y^2) for the first level (scale1 NULL); hi / lo = n_max / n_min (int32) or y_max / y_min. */
typedef struct svnet_block_tail_desc {
    const void* stat1; const double* stat_v; int64_t E, Os, Ov;
    const float* scale1;
    const float* gamma1; const float* beta1; float* running_mean1; float* running_var1;
    const float* gamma2; const float* beta2; float* running_mean2; float* running_var2;
    int training; float eps, momentum;
    float* coef; int64_t* num_batches_tracked1; int64_t* num_batches_tracked2;
    svnet_gate_fwd_job gate;
    const void* hi; const void* lo; const float* mv; const float* mvn;
    int64_t P, N; float slope;
    float* s_out; float* v_out; float* s_cat; int64_t s_ld; float* v_cat; int64_t v_ld;
    void* knn_workspace; size_t knn_workspace_bytes;          /* NULL / 0: no table */
} svnet_block_tail_desc;
int svnet_block_tail_supported(int64_t P, int64_t N, int64_t Os, int64_t Ov, int with_knn_table);
int svnet_edgeblock_tail_f32(const svnet_block_tail_desc* desc, void* stream);
int svnet_xyzblock_tail_f32(const svnet_block_tail_desc* desc, void* stream);

/* Backward of the fused edge block: a point-level prelude reduces the batch-statistic terms of both BatchNorms,
 * then the edge pass produces all gradients from the point tables and the n16 / planes the forward kept
 * (csrc/edgeblock_bwd.hip).  Outputs of svnet_edgeblock_bwd_f32 (caller zero-fills every *_acc, dzc, dbeta_perm):
 *   dn_out [E,Os]            dL/d(scale*n) per edge          -> GX = dn_out^T . x_b via svnet_gemm_f32 (ternary A)
 *   x_sign32/x_nz32          row-sliced planes of x_b in fused column order, [ceil(E/64), 320] uint64 viewed as uint32
 *   msg [E,R]                per-edge contributions to the NEIGHBOUR j of each edge (summed by svnet_edgeblock_bwd_gather_f32)
 *   ub_tab, ge_tab [P,3,Ov]  per-point operands from which that kernel recomputes the neighbour's share of dL/dv'
 *   ds_acc [P,Cs], dv_acc [P,3,Cv]   centre parts of the gradients of the point tables
 *   dvc [P,3,Ov], dzc [P,3,3]  centre sums of dL/dv' and dL/dz   (dU = sum_j - dvc, dT = dvc; dZp = sum_j - dzc, dZq = dzc)
 *   dbeta_perm [SVNET_DBETA_SLICES][320]   dL/dbeta in fused column order, spread over slices that
 *                            svnet_edgeblock_bwd_gather_f32 sums (one set of 320 addresses was an atomic hot spot)        */
#define SVNET_DBETA_SLICES 64
typedef struct svnet_edgeblock_bwd_desc {
    int64_t B, N, k;
    int Cs, Cv, Os, Ov;
    const float* v; const int64_t* idx; const float* zz; const float* ut;
    const int16_t* n16; const uint64_t* planes;   /* kept by svnet_edgeblock_fwd_f32 */
    const uint16_t* w1bt;        /* svnet_edgeblock_wbt_bf16: sign(W1) as bf16 MFMA fragments, 320 * 16*ceil(Os/16) values */
    const float* scale1;
    const uint8_t* slot_max; const uint8_t* slot_min;
    const float* coef;           /* from svnet_edgeblock_coeffs_f32 */
    const float* gate;           /* [B,Ov] */
    const float* gy;             /* [P,Os] from the prelude */
    const float* bcoef;          /* [8*Os + 2*Ov + 4] from svnet_edgeblock_bwd_coeffs_f32 (with scale1) */
    const float* gv;             /* upstream gradient of v_out [P,3,Ov] */
    const float* gconst;         /* [B,2Cs]: dL/d(gate input) / (N*k) */
    float* dn_out; uint32_t* x_sign32; uint32_t* x_nz32;
    float* msg;                  /* [E, svnet_edgeblock_msg_stride]: per-edge neighbour contributions (written, not accumulated) */
    float* ub_tab; float* ge_tab; /* [P,3,Ov] each: T_i - U_i and gv_i*gate/k, written by the vector path for the gather kernel */
    float* ds_acc; float* dv_acc; float* dvc; float* dzc; float* dbeta_perm;   /* centre sums (atomics) / dvc written */
    int64_t* debug;              /* optional [4]: {count, first bad edge, its idx value, N}; edges with idx outside [0,N) are skipped */
    int parts;                   /* 0 or 3: both kernels; 1: vector path only (dvc, ub_tab, ge_tab); 2: scalar/tile path only.  The two
                                    are independent, so a caller may issue them on two streams                                   */
} svnet_edgeblock_bwd_desc;
/* wbt: 320 * 16*ceil(Os/16) bf16 values, [column tile (10)][k-step][lane (64)][8] (the tile kernel's B-fragment order) */
/* w_dense (may be NULL; then Cs / Cv are unused): one uint32, set to 1 when the planes show no zero weight among the columns in use
 * (Cs bits of words 0 - 1, 2 Cv bits of words 2 - 4), else 0 - see svnet_edgeblock_desc                                                   */
int svnet_edgeblock_wbt_bf16(const uint64_t* w_sign, const uint64_t* w_nz, int64_t Os, int64_t Cs, int64_t Cv, uint16_t* wbt,
                             uint32_t* w_dense, void* stream);
/* gy = Gs*lrelu'(y) at the pooled edge; red [SVNET_RED_SLICES][2*Os], redv [SVNET_RED_SLICES][2*Ov], dgate [B,Ov] accumulate (caller
 * zero-fills).  The batch sums are spread over slices (workgroup w adds to slice w % SVNET_RED_SLICES) that
 * svnet_edgeblock_bwd_coeffs_f32 adds up: 512 workgroups adding to ONE set of 2*Os + 2*Ov addresses spent more time in the memory
 * side's atomic queue (11 - 24 us of the kernel's 23 - 35, measured) than reading their rows.                                  */
int svnet_edgeblock_bwd_prelude_f32(const float* gs, const float* gv, const int32_t* n_max, const int32_t* n_min,
                                    const float* mv, const float* mvn, const float* coef, const float* scale1,
                                    const float* gate, int64_t P, int64_t N, int64_t Os, int64_t Ov, float slope,
                                    float* gy, float* red, float* redv, float* dgate,
                                    /* optional SECOND gradient source, added to the first: gs2[p*gs2_ld + o], gv2[(p*3+d)*gv2_ld + c] - column
                                       slices of the gradient of svcat([x1, .., xn]) read where they lie (gs / gv may then be NULL: the
                                       concatenation is the only consumer); with gv2 the summed vector gradient is written to gv_sum [P,3,Ov] */
                                    const float* gs2, int64_t gs2_ld, const float* gv2, int64_t gv2_ld, float* gv_sum, void* stream);
/* bcoef = [m1 | m2 | cs (Os each) | c0 | c1 (Ov each)], and when scale1 != NULL (binarized layer), from the next multiple of
 * 4 floats on, the tile kernel's per-channel table [cs | alpha | beta | scale | pooled-is-max] (5*Os): allocate
 * 8*Os + 2*Ov + 4 floats.  BatchNorm parameter gradients are written to dgamma*, dbeta*.                              */
/* red / redv: the SVNET_RED_SLICES slices a prelude kernel filled (svnet_edgeblock_bwd_prelude_f32, svnet_xyzblock_bwd_prelude_f32). */
int svnet_edgeblock_bwd_coeffs_f32(const float* red, const float* redv, const float* coef, const float* gamma1,
                                   const float* gamma2, int64_t E, int64_t Os, int64_t Ov, int training, const float* scale1,
                                   float* bcoef, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2,
                                   const svnet_gate_bwd_job* gate_job /* may be NULL */, void* stream);
int svnet_edgeblock_bwd_f32(const svnet_edgeblock_bwd_desc* desc, void* stream);
/* Reverse neighbour lists of a kNN graph (idx [B*N,k], cloud-local ids): the edges e = i*k + t that point at j are
 * rev_edge[rev_range[2j] .. rev_range[2j+1]), their source points i (global ids) rev_src[..].  rev_range [2*B*N], rev_edge and
 * rev_src [B*N*k]; N <= 8192; ids outside [0,N) are skipped.  Optional (all three or none): lists longer than `chunk` entries
 * are cut up for svnet_edgeblock_bwd_gather_f32 - every chunk after the first becomes an item (point, chunk number) of ovf_items
 * [2 * (2*B*N*k/chunk + 1)] int32, their number is ADDED to ovf_count[0] (caller zero-fills).                                   */
int svnet_knn_reverse_i32(const int64_t* idx, int64_t B, int64_t N, int64_t k, int32_t* rev_range, int32_t* rev_edge,
                          int32_t* rev_src, int64_t chunk, int32_t* ovf_items, int32_t* ovf_count, void* stream);
/* Row stride (floats) of the per-edge message rows msg[e] = [dL/ds_j (Cs) | dL/dv_j (3*Cv) | dL/dz (9) | pad]
 * that svnet_edgeblock_bwd_f32 writes instead of scattering with float atomics.                                       */
int64_t svnet_edgeblock_msg_stride(int64_t Cs, int64_t Cv, int64_t Ov);
/* Sums the message rows over the reverse lists (one wave per destination point, no atomics), recomputes the neighbour's share
 * of dL/dv' of every incoming edge from ut (U_j), ub_tab / ge_tab (source point) and the coefficients, and finishes the
 * point-level gradients: acat [3P, acat_ld] = [dU | dT | dZp | dZq | pad] (dU = sum - dvc, dT = dvc, ...), ds_acc [P,Cs] +=,
 * dv_acc [P,3,Cv] +=, dbeta1 [2Cs+6Cv] = the slices of dbeta_perm summed, in the reference's feature order.                                  */
int svnet_edgeblock_bwd_gather_f32(const float* msg, const int32_t* rev_range, const int32_t* rev_edge, const int32_t* rev_src,
                                   const float* ut, const float* ub_tab, const float* ge_tab, const float* coef, const float* bcoef,
                                   int64_t Os, const float* dvc, const float* dzc, int64_t P /* = B*N points */,
                                   int64_t N /* points per cloud (sets the XCD-aware point order) */, int64_t Cs, int64_t Cv,
                                   int64_t Ov, float* acat,
                                   int64_t acat_ld /* row stride of acat, >= 2Ov+6 (a multiple of 4 keeps the GEMM's loads 16-byte) */,
                                   float* ds_acc, float* dv_acc, const float* dbeta_perm, float* dbeta1,
                                   int64_t chunk, const int32_t* ovf_items, const int32_t* ovf_count /* as given to svnet_knn_reverse_i32, or 0 / NULL:
                                   a wave then walks a whole list, however long */, void* stream);
/* STE chain rule (svnet_binweight_grad_f32's formula, ASSIGNED) for linear1 from GXp [Os,320] (fused column order), for
 * linear2 from GXc[0:2Ov] and for the v2s frame from GXc[2Ov:2Ov+6]  (GXc [2Ov+6, Cv]).                              */
int svnet_edgeblock_bwd_params_f32(const float* GXp, const float* GXc, const float* W1, const float* scale1, const float* W2,
                                   const float* scale2, const float* Wz, const float* scalez, int64_t Os, int64_t Ov,
                                   int64_t Cs, int64_t Cv, float* dW1, float* dscale1, float* dW2, float* dscale2, float* dWz,
                                   float* dscalez, void* stream);

/* ------------------------------------------------------------------ fused FIRST edge layer (full precision)
 * get_graph_feature (sv_util.py:28-62) -> Vector2Scalar(2,3) (init_scalar, sv_layers.py:111-129)
 * -> SVBlock((6,2),(Os,Ov)) fp (sv_layers.py:172-196) -> svpool (sv_util.py:118-132) in one pass over the edges
 * (csrc/xyzblock.hip).  x: contiguous [B,3,N]; idx [B*N,k] cloud-local; w0/wz: [3,2] (init_scalar / block v2s),
 * w1: [Os,12], w2: [Ov,2].  Per-point outputs as for the binarized block, with fp32 y_max / y_min instead of
 * integer sums; stat_y [SVNET_RED_SLICES][2*Os], stat_v [SVNET_RED_SLICES][2*Ov] fp64 sums (slices, added up by
 * svnet_xyzblock_coeffs_f32) and gate_sum [B,6] accumulate (caller zero-fills).                                   */
typedef struct svnet_xyzblock_desc {
    int64_t B, N, k;
    int Os, Ov;
    const float* x; const int64_t* idx;
    const float* w0; const float* wz; const float* w1; const float* w2;
    float* y_max; float* y_min; uint8_t* slot_max; uint8_t* slot_min;
    float* mv; float* mvn;
    double* stat_y; double* stat_v; double* gate_sum;
    int64_t nc;        /* vector channels of the edge feature: 0 / 2 = [x_j - x_i | x_i] (get_graph_feature), 3 = + x_j x x_i          */
                       /* (get_graph_feature_cross); w0 / wz [3,nc], w1 [Os,6nc], w2 [Ov,nc], gate_sum [B,3nc]                    */
} svnet_xyzblock_desc;
int svnet_xyzblock_fwd_f32(const svnet_xyzblock_desc* desc, void* stream);
/* coef: same layout as svnet_edgeblock_coeffs_f32 (A1 = gamma*invstd, B1 = beta - gamma*mean*invstd, ...).        */
int svnet_xyzblock_coeffs_f32(const double* stat_y, const double* stat_v, int64_t E, int64_t Os, int64_t Ov,
                              const float* gamma1, const float* beta1, float* running_mean1, float* running_var1,
                              const float* gamma2, const float* beta2, float* running_mean2, float* running_var2,
                              int training, float eps, float momentum, float* coef, int64_t* num_batches_tracked1,
                              int64_t* num_batches_tracked2, const svnet_gate_fwd_job* gate_job /* may be NULL */, void* stream);
int svnet_xyzblock_apply_f32(const float* y_max, const float* y_min, const float* mv, const float* mvn, const float* coef,
                             const float* gate, int64_t P, int64_t N, int64_t Os, int64_t Ov, float slope, float* s_out,
                             float* v_out, float* s_cat, int64_t s_ld, float* v_cat, int64_t v_ld /* as svnet_edgeblock_apply_f32 */,
                             void* stream);
/* Backward: the coordinates need no gradient, so the edge pass only accumulates parameter gradients
 * gw = [dW1 (Os*12) | dW2 (Ov*2) | dW0 (6) | dWz (6)]: a SLICED accumulator of GW = Os*6nc + Ov*nc + 6nc floats (SVNET_SLICED_LEN(GW),
 * caller zero-fills, totals by svnet_slices_sum_f32).  bcoef comes from
 * svnet_edgeblock_bwd_coeffs_f32 (same coefficient layout); gconst [B,6] = dL/d(gate input) / (N*k).             */
int svnet_xyzblock_bwd_prelude_f32(const float* gs, const float* gv, const float* y_max, const float* y_min, const float* mv,
                                   const float* mvn, const float* coef, const float* gate, int64_t P, int64_t N, int64_t Os,
                                   int64_t Ov, float slope, float* gy, float* red, float* redv, float* dgate, const float* gs2,
                                   int64_t gs2_ld, const float* gv2, int64_t gv2_ld, float* gv_sum /* as svnet_edgeblock_bwd_prelude_f32 */,
                                   void* stream);
typedef struct svnet_xyzblock_bwd_desc {
    int64_t B, N, k;
    int Os, Ov;
    const float* x; const int64_t* idx;
    const float* w0; const float* wz; const float* w1; const float* w2;
    const uint8_t* slot_max; const uint8_t* slot_min;
    const float* coef; const float* bcoef; const float* gate; const float* gy; const float* gv; const float* gconst;
    float* gw;
    int64_t nc;        /* as in svnet_xyzblock_desc; gw = [dW1 (Os*6nc) | dW2 (Ov*nc) | dW0 (3nc) | dWz (3nc)], gconst [B,3nc]   */
} svnet_xyzblock_bwd_desc;
int svnet_xyzblock_bwd_f32(const svnet_xyzblock_bwd_desc* desc, void* stream);

/* Weight-gradient product of a fused edge layer: GX[o*320 + c] += sum_e dy[e,o] * x_b[e,c] over the E = B*N*k edge rows, with
 * dy[e,o] = dL/dy_pre RECOMPUTED from the forward's int16 sums: chc[o]*g - (chc[Os+o] + chc[2Os+o]*n16[e,o]), g = gy[p,o] when the
 * pooled slot of (p,o) (slot_max or slot_min by chc[4Os+o]) is e - p*k, else 0; chc = the per-channel table svnet_edgeblock_bwd_coeffs_f32
 * leaves at bcoef + ((3*Os + 2*Ov + 3) & ~3).  x_sign / x_nz: the row-sliced planes the tile kernel writes.  8 <= k <= 64.
 * GX accumulates (caller zero-fills); q_tile_mask as tern_tile_mask of svnet_gemm_desc.                                          */
int svnet_edgeblock_wgrad_f32(const int16_t* n16, const uint8_t* slot_max, const uint8_t* slot_min, const float* gy,
                              const float* chc, const uint64_t* x_sign, const uint64_t* x_nz, int64_t E, int64_t k, int64_t Os,
                              float* GX, uint32_t q_tile_mask, void* stream);

/* ------------------------------------------------------------------ Vector2Scalar (sv_layers.py:104-129)
 * v: [M,3,C]; w_eff: [J,C] effective weights (scale*sign(W) or W); z[m,i,j] = sum_c v[m,i,c] w_eff[j,c];
 * s[m, d*J+j] = sum_i v[m,i,d] z[m,i,j].  z_out optional ([M,3,J]).  J == 3, C <= 768.              */
int svnet_v2s_fwd_f32(const float* v, const float* w_eff, int64_t M, int64_t C, int64_t J, float* s, float* z_out,
                      void* stream);
/* ds: [M,C*J]; dz_in: optional gradient arriving at z ([M,3,J]); dv: [M,3,C] (written);
 * GX: a SLICED accumulator of J*C floats (SVNET_SLICED_LEN(J*C), caller zero-fills; totals by svnet_slices_sum_f32):
 * GX[j,c] = sum_m sum_i dz[m,i,j] v[m,i,c].  */
int svnet_v2s_bwd_f32(const float* v, const float* w_eff, const float* ds, const float* dz_in, int64_t M, int64_t C,
                      int64_t J, float* dv, float* GX, void* stream);
/* cat[pre, Vector2Scalar(v)] written in place (sv_layers.py:187-188: the input of an SVBlock's linear1): out [M, out_ld] rows =
 * [pre (pre_cols floats) | s (C*J floats)], out_ld >= pre_cols + C*J; and the backward with the gradient of s given as a column slice
 * of a wider gradient (rows of stride ds_ld).                                                                                       */
int svnet_v2s_cat_fwd_f32(const float* v, const float* w_eff, const float* pre, int64_t pre_cols, int64_t M, int64_t C, int64_t J,
                          float* out, int64_t out_ld, void* stream);
/* ... and, from the copy of `pre` it makes anyway, the per-cloud column sums of pre: pre_sum [M / rows_per_cloud, pre_cols] fp64 (caller
 * zero-fills) - the input of the SVBlock's gate MLP (sv_layers.py:179: s.mean over the points; svnet_gate_mlp_fwd_f32 with gin_f64 =
 * pre_sum, in_scale = 1 / rows_per_cloud), so that no pooling pass reads s a second time.  svnet_v2s_cat_sum_supported: 24 < C <= 192,
 * pre_cols <= 256 (C <= 96) / 512, whole clouds, rows_per_cloud a multiple of 32 (C <= 96) / 16.                                      */
int svnet_v2s_cat_sum_supported(int64_t M, int64_t C, int64_t pre_cols, int64_t rows_per_cloud);
int svnet_v2s_cat_sum_fwd_f32(const float* v, const float* w_eff, const float* pre, int64_t pre_cols, int64_t M, int64_t C, int64_t multi,
                              float* out, int64_t out_ld, double* pre_sum, int64_t rows_per_cloud, void* stream);
int svnet_v2s_bwd_ld_f32(const float* v, const float* w_eff, const float* ds, int64_t ds_ld, const float* dz_in, int64_t M, int64_t C,
                         int64_t J, float* dv, float* GX, void* stream);
/* Frame projection with a GIVEN per-row frame z [M,3,J] (the back-projection einsum 'bimj,bijk->bimk' of
 * sv_pointnet_partseg.py:89): s[m, c*J+j] = sum_i v[m,i,c] z[m,i,j].  Backward: dv [M,3,C] and dz [M,3,J], both written.  */
int svnet_vproject_fwd_f32(const float* v, const float* z, int64_t M, int64_t C, int64_t J, float* s, void* stream);
int svnet_vproject_bwd_f32(const float* v, const float* z, const float* ds, int64_t M, int64_t C, int64_t J, float* dv,
                           float* dz, void* stream);

/* ------------------------------------------------------------------ BatchNorm1d over rows (+ activation)
 * (sv_layers.py:166-167,189-190; nn.BatchNorm1d defaults eps=1e-5, momentum=0.1)
 * stats: sums[0:C] = sum_m x, sums[C:2C] = sum_m x^2 in fp64; `sums` is a sliced accumulator of 2*C doubles (SVNET_SLICED_LEN(2*C),
 * zero-filled by the caller).
 * kind 0: x is [M,C];  kind 1: x is [M,3,C] and the statistic is n = ||x[m,:,c]||_2 + 1e-6 (VectorBN, :94). */
int svnet_colstats_f64(const float* x, int64_t M, int64_t C, int kind, double* sums, void* stream);
/* The vector path's linear layer of an SVBlock on rows together with the batch statistics of the VectorBN behind it
 * (sv_layers.py:44-49 with bw only, :86-102, :192): v [P,3,K] rows, w_b [O,K] the +-1 / 0 values of sign(W), col_scale [O] (may be NULL);
 * y[p,a,o] = col_scale[o] * sum_k v[p,a,k] w_b[o,k];  `sums` as svnet_colstats_f64 kind 1 leaves it for y (a sliced accumulator of
 * 2*O doubles, zero-filled by the caller): the separate statistics pass over y is not needed.  K <= 96, O <= 256.                   */
int svnet_vlinear_stats_f32(const float* v, int64_t P, int64_t K, const float* w_b, const float* col_scale, int64_t O, float* y,
                            double* sums, void* stream);
/* mean/invstd from the sums (training: `sums` = the sliced accumulator svnet_colstats_f64 / svnet_binlinear_i8_fwd_f32 filled; its totals
 * are left in sums[0:2C]), running-stat update and num_batches_tracked += 1 (each may be NULL). */
int svnet_bn_finalize_f32(double* sums, int64_t M, int64_t C, float eps, float momentum, float* mean,
                          float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                          void* stream);
/* eval mode: mean = running_mean, invstd = 1/sqrt(running_var+eps).                                 */
int svnet_bn_eval_stats_f32(const float* running_mean, const float* running_var, int64_t C, float eps, float* mean,
                            float* invstd, void* stream);
/* y = act((x-mean)*invstd*gamma+beta); act: 0 none, 1 leaky-relu(slope), 2 relu.                    */
int svnet_bn_act_fwd_f32(const float* x, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, int64_t M, int64_t C, int act, float slope, float* y, void* stream);
/* backward, pass 1: red[0:C] = sum g', red[C:2C] = sum g'*xhat (`red`: a sliced accumulator of 2*C floats, SVNET_SLICED_LEN(2*C), zero-filled),
 * g' = g * act'(bn(x)).  pass 2: dx; train_stats=1 subtracts the batch-statistic terms; it adds the slices of `red` up and leaves
 * dgamma = red[C:2C], dbeta = red[0:C] (without pass 2: svnet_slices_sum_f32(red, 2C)).           */
int svnet_bn_act_bwd_reduce_f32(const float* g, const float* x, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, int64_t M, int64_t C, int act, float slope,
                                float* red, void* stream);
int svnet_bn_act_bwd_apply_f32(const float* g, const float* x, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, float* red, int64_t M, int64_t C, int act,
                               float slope, int train_stats, float* dx, void* stream);

/* ------------------------------------------------------------------ VectorBN (+ gate)  (sv_layers.py:81-102, :194)
 * v: [M,3,C]; n = ||v||+1e-6; out = v * (bn(n)/n) * gate[b,c]  (gate NULL = 1; b = m / rows_per_batch). */
int svnet_vbn_fwd_f32(const float* v, const float* mean, const float* invstd, const float* gamma, const float* beta,
                      const float* gate, int64_t rows_per_batch, int64_t M, int64_t C, float* out, void* stream);
/* The same forward with the statistics finalised inside (training): sums = svnet_colstats_f64(v, kind 1); every thread derives its
 * channel's mean / invstd (svnet_bn_finalize_f32's arithmetic), workgroup 0 writes them to mean / invstd [C] (the backward needs them)
 * and updates running_mean / running_var / num_batches_tracked (each may be NULL) - no one-workgroup launch between the two passes. */
int svnet_vbn_fwd_stats_f32(const float* v, const double* sums, float eps, float momentum, float* mean, float* invstd,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* gamma,
                            const float* beta, const float* gate, int64_t rows_per_batch, int64_t M, int64_t C, float* out,
                            void* stream);
/* pass 1: red[0:C] = sum dr, red[C:2C] = sum dr*nhat (`red`: a sliced accumulator of 2*C floats, SVNET_SLICED_LEN(2*C)),
 * dgate[b,c] += sum g*v*q  (caller zero-fills red, dgate) */
int svnet_vbn_bwd_reduce_f32(const float* g, const float* v, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, const float* gate, int64_t rows_per_batch,
                             int64_t M, int64_t C, float* red, float* dgate, void* stream);
int svnet_vbn_bwd_apply_f32(const float* g, const float* v, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, const float* gate, float* red,
                            int64_t rows_per_batch, int64_t M, int64_t C, int train_stats, float* dv, void* stream);

/* ------------------------------------------------------------------ pooling (sv_util.py:118-132 svpool; adaptive pools of sv_dgcnn_cls.py:72-73)
 * x: [outer, R, inner] -> out [outer, inner].  mode 0 = max (argmax int32 saved, first index on ties),
 * mode 1 = mean.  workspace (optional, svnet_pool_workspace_bytes): lets a long reduction with few outputs (pooling over
 * the N points) be split over workgroups; the partial results are combined in a fixed order (bit-reproducible, no float
 * atomics).  NaN inputs are not supported on the split max path.               */
size_t svnet_pool_workspace_bytes(int64_t outer, int64_t R, int64_t inner, int mode);
int svnet_pool_fwd_f32(const float* x, int64_t outer, int64_t R, int64_t inner, int mode, float* out, int64_t out_ld,
                       int32_t* argmax, void* workspace, size_t workspace_bytes, void* stream);
/* [max | mean] over the reduced axis in ONE pass over x (sv_dgcnn_cls.py:72-74: adaptive max pool and adaptive avg pool of the same
 * feature): out_max / out_mean [outer, inner] rows of stride out_ld, argmax as svnet_pool_fwd_f32; R >= 256; workspace =
 * svnet_pool_workspace_bytes(outer, R, inner, 0) + svnet_pool_workspace_bytes(outer, R, inner, 1) bytes.  The mean's partial sums
 * are added in a fixed order (bit-reproducible), like mode 1 of svnet_pool_fwd_f32.                                              */
int svnet_pool_maxmean_fwd_f32(const float* x, int64_t outer, int64_t R, int64_t inner, float* out_max, float* out_mean,
                               int64_t out_ld, int32_t* argmax, void* workspace, size_t workspace_bytes, void* stream);
/* conv5 of the classifier (sv_layers.py:189-190 bn1 + LeakyReLU, then sv_dgcnn_cls.py:72-74 max / avg pool over the points): BatchNorm
 * with the given per-channel statistics (+ activation: 0 none, 1 LeakyReLU(slope), 2 ReLU) of y [outer*R, inner], pooled
 * [max | mean] over R in the same pass - the activated tensor is never written.  Outputs / argmax / workspace as
 * svnet_pool_maxmean_fwd_f32 (R >= 256).  The values equal pooling svnet_bn_act_fwd_f32's output bit for bit.                     */
int svnet_bn_pool_fwd_f32(const float* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                          int64_t outer, int64_t R, int64_t inner, int act, float slope, float* out_max, float* out_mean,
                          int64_t out_ld, int32_t* argmax, void* workspace, size_t workspace_bytes,
                          int workspace_zeroed /* != 0: the first outer*inner*8 bytes of the workspace (the arg-max keys) are zero already */,
                          void* stream);
/* Its backward from the POOLED gradients (gmax, gmean: rows of stride g_ld): the gradient of the activated tensor,
 * (r == argmax ? gmax : 0) + gmean / R, is formed on the fly.  red [2*inner] (caller zero-fills) = [dbeta | dgamma];
 * dy [outer*R, inner] (may be NULL) the gradient of y, with the batch-statistic terms when train_stats.                           */
int svnet_bn_pool_bwd_f32(const float* gmax, const float* gmean, int64_t g_ld, const int32_t* argmax, const float* y,
                          const float* mean, const float* invstd, const float* gamma, const float* beta, int64_t outer, int64_t R,
                          int64_t inner, int act, float slope, int train_stats, float* red, float* dy, void* stream);   /* out[o*out_ld + i]; argmax [outer,inner] */
int svnet_pool_bwd_f32(const float* g, const int32_t* argmax, int64_t outer, int64_t R, int64_t inner, int mode,
                       float* dx, void* stream);
/* The tail of the classifier's vector path as one pass over linear2's product v [B*N, 3, C] (C <= 192): VectorBN + gate
 * (sv_layers.py:86-102,193-194) -> Vector2Scalar of svfuse (sv_layers.py:111-129,206-220; w_eff [3, C] = scale * sign(W)) -> [max | mean]
 * over each cloud's N points (sv_dgcnn_cls.py:70-74) - neither VectorBN's output nor the [B*N, 3C] scalars are written.
 * sums != NULL (training): the sliced fp64 accumulator svnet_colstats_f64(v, B*N, C, kind 1) filled; the batch statistics are
 * finalised inside (mean / invstd [C] written, running statistics and *nbt updated as svnet_vbn_fwd_stats_f32 does).  sums == NULL
 * (eval): mean / invstd are INPUTS (svnet_bn_eval_stats_f32).  gate [B, C] or NULL.  out_max / out_mean [B, 3C] rows of stride out_ld,
 * argmax [B, 3C] int32 (point index in the cloud, first index on ties); workspace = svnet_vtail_workspace_bytes(B, N, C) bytes.
 * The maxima equal pooling the layer-wise chain's output; the means add their partial sums in a fixed order (bit-reproducible).   */
size_t svnet_vtail_workspace_bytes(int64_t B, int64_t N, int64_t C);
int svnet_vtail_fwd_f32(const float* v, const double* sums, float eps, float momentum, float* mean, float* invstd,
                        float* running_mean, float* running_var, long long* nbt, const float* gamma, const float* beta,
                        const float* gate, const float* w_eff, int64_t B, int64_t N, int64_t C, float* out_max, float* out_mean,
                        int64_t out_ld, int32_t* argmax, void* workspace, size_t workspace_bytes,
                        int workspace_zeroed /* != 0: the first B*3C*8 bytes (the arg-max keys) are zero already */, void* stream);
/* Its backward up to VectorBN's batch sums, from the POOLED gradients (gmax / gmean: rows of stride g_ld): the gradient of the scalars,
 * (point == argmax ? gmax : 0) + gmean / N, is formed on the fly, Vector2Scalar's backward runs in registers.  Outputs: g5 [B*N, 3, C] =
 * dL/d(VectorBN's output) (may be NULL: see svnet_vtail_bwd_apply_f32), to be handed to svnet_vbn_bwd_apply_f32 as its `g` together with `red`; red = SLICED accumulator of 2C floats
 * (SVNET_SLICED_LEN(2C), zero-filled; the apply pass adds the slices up and leaves [dbeta | dgamma]); dgate [B, C] (zero-filled, += ; may
 * be NULL when gate is); GX = sliced accumulator of 3C floats: dL/d(w_eff) as [3][C] (svnet_binweight_grad_f32 with gx_sliced).        */
int svnet_vtail_bwd_f32(const float* v, const float* mean, const float* invstd, const float* gamma, const float* beta,
                        const float* gate, const float* w_eff, const float* gmax, const float* gmean, int64_t g_ld,
                        const int32_t* argmax, int64_t B, int64_t N, int64_t C, float* red, float* dgate, float* GX, float* g5,
                        void* stream);
/* The rest of that backward WITHOUT the stored g5 (pass NULL for it above): the same per-point recomputation of dL/d(VectorBN's output), then
 * VectorBN's apply pass on it (svnet_vbn_bwd_apply_f32's formulas) with the totals of `red`'s slices - dv [B*N, 3, C] = dL/dv, and
 * [dbeta | dgamma] left in the first 2C elements of red.  One read of v and one write instead of two reads, a write and a read.          */
int svnet_vtail_bwd_apply_f32(const float* v, const float* mean, const float* invstd, const float* gamma, const float* beta,
                              const float* gate, const float* w_eff, const float* gmax, const float* gmean, int64_t g_ld,
                              const int32_t* argmax, int64_t B, int64_t N, int64_t C, float* red, int train_stats, float* dv, void* stream);
/* dx[o,r,i] = add[(o*R + r)*add_ld + i] + gmean[o,i] / R: the backward of a mean over r (sv_layers.py:179: the gate's pooled input) added to
 * another gradient of the same tensor that arrives as rows of stride add_ld (the s columns of the gradient of cat[s, Vector2Scalar(v)],
 * sv_layers.py:187-188) - one pass instead of a broadcast pass and a strided elementwise add.                                       */
int svnet_pool_mean_bwd_add_f32(const float* gmean, const float* add, int64_t add_ld, int64_t outer, int64_t R, int64_t inner,
                                float* dx, void* stream);
/* Backward of cat(max, mean) over the same axis (the classifier's global pooling, sv_dgcnn_cls.py:72-74) in one pass:
 * gmax / gmean: dL/dmax and dL/dmean as rows of stride g_ld (column slices of the pooled feature's gradient),
 * dx[o,r,i] = (argmax[o,i] == r ? gmax[o,i] : 0) + gmean[o,i] / R.       */
int svnet_pool_maxmean_bwd_f32(const float* gmax, const float* gmean, int64_t g_ld, const int32_t* argmax, int64_t outer, int64_t R,
                               int64_t inner, float* dx, void* stream);

/* ------------------------------------------------------------------ element-wise activations of the gate (sv_layers.py:156-161)
 * kind 1 = relu, 2 = sigmoid, 3 = leaky-relu(0.2).  Backward uses the OUTPUT y.                      */
int svnet_act_fwd_f32(const float* x, int64_t n, int kind, float* y, void* stream);
int svnet_act_bwd_f32(const float* g, const float* y, int64_t n, int kind, float* dx, void* stream);

/* ------------------------------------------------------------------ gate MLP of an SVBlock (sv_layers.py:156-161,179-183)
 * gate[b,:] = sigmoid(W2 . relu(W0 . (in_scale*gin[b,:])));  W0 [H,Cin], W2 [Ov,H], no biases; h [B,H] is saved for the
 * backward.  The input is either gin (fp32) or gin_f64 (the fp64 gate_sum of a fused edge layer), which is rounded to
 * fp32 into gin_out [B,Cin] first (kept by the caller for the backward).
 * Backward: dgin = out_scale * dL/d(in_scale*gin) (may be NULL), dW0 / dW2 ACCUMULATE (float atomics).       */
int svnet_gate_mlp_fwd_f32(const float* gin, const double* gin_f64, float* gin_out, float in_scale, const float* W0,
                           const float* W2, int64_t B, int64_t Cin, int64_t H, int64_t Ov, float* h, float* gate,
                           const float* rows /* may be NULL */, int64_t R /* see svnet_gate_fwd_job */, void* stream);
int svnet_gate_mlp_bwd_f32(const float* dgate, const float* gate, const float* h, const float* gin, float in_scale,
                           const float* W0, const float* W2, int64_t B, int64_t Cin, int64_t H, int64_t Ov, float out_scale,
                           float* dgin, float* dW0, float* dW2, void* stream);

/* ------------------------------------------------------------------ classifier heads: dense layers over M = batch rows (M <= 64)
 * (models/sv_dgcnn_cls.py:76-80, models/sv_pointnet_cls.py:59-61:  act(bn(linear(x)))  with linear = sv_layers.Linear(bw, ba),
 *  sv_layers.py:35-51, bn = nn.BatchNorm1d, act = leaky_relu / relu; then nn.Linear)
 * One packing pass over the layer's input, ONE pass forward (binarized product + batch statistics + BatchNorm + activation) and
 * two passes backward (per output channel: BatchNorm backward, weight / scale gradients; per 64 input columns: dx, dbeta) instead
 * of ~12 launch-bound kernels per layer.  Integer counts, outputs and gradients follow svnet_binlinear_fwd_f32 +
 * svnet_colstats_f64 / svnet_bn_finalize_f32 / svnet_bn_act_* and their backward formulas; nothing is accumulated atomically.
 * pack: x [M,K] + beta [K] -> x_sign / x_nz / x_ste row-major [M][ceil(K/64)] (x_ste may be NULL) and the column words
 *       xc_sign / xc_nz [K] (bit m = row m; both NULL or both given - the backward needs them; xc_ste [K], optional, is the STE
 *       plane in the same form = the row-sliced layout of svnet_binlinear_fwd_f32's saved planes).                               */
typedef struct svnet_binhead_desc {
    int64_t M, K, O;                  /* rows (1..64), input columns, output channels */
    const float* W;                   /* [O,K] fp32 master weights (backward: STE mask |W| <= 1.2 and sign) */
    const uint64_t* w_sign;           /* [O][wld] plane words of sign(W) (svnet_binweight_prepare_f32) */
    const uint64_t* w_nz;
    int64_t wld;                      /* words per weight row (>= ceil(K/64)) */
    const float* w_b;                 /* [O,K] sign(W) as +-1 / 0 (backward) */
    const float* scale;               /* [O] */
    const float* gamma;               /* BatchNorm weight / bias [O] */
    const float* bn_beta;
    float* running_mean;              /* [O] or NULL; updated when training (eval: read) */
    float* running_var;
    long long* nbt;                   /* num_batches_tracked (+= 1 when training) or NULL */
    int training;                     /* 1 = batch statistics + STE backward; 0 = running statistics (forward only) */
    float eps, momentum;
    int act;                          /* 0 none, 1 LeakyReLU(slope), 2 ReLU */
    float slope;
    const uint64_t* x_sign;           /* packed input (svnet_binhead_pack_f32) */
    const uint64_t* x_nz;
    const uint64_t* x_ste;            /* backward */
    const uint64_t* xc_sign;          /* backward */
    const uint64_t* xc_nz;
    float* y;                         /* [M,O] pre-BatchNorm output n * scale (forward: written; backward: read) */
    float* mean;                      /* [O] statistics used (forward: written; backward: read) */
    float* invstd;
    float* out;                       /* [M,O] forward result */
    const float* g;                   /* backward: dL/dout [M,O] */
    float* dnT;                       /* backward scratch [O][64] */
    float* dW;                        /* [O,K] (may be NULL) */
    float* dscale;                    /* [O] */
    float* dgamma;                    /* [O] */
    float* dbn_beta;                  /* [O] */
    float* dx;                        /* [M,K] and dbeta_in [K]: both NULL or both given */
    float* dbeta_in;
} svnet_binhead_desc;
int svnet_binhead_pack_f32(const float* x, const float* beta, int64_t M, int64_t K, uint64_t* x_sign, uint64_t* x_nz,
                           uint64_t* x_ste, uint64_t* xc_sign, uint64_t* xc_nz, uint64_t* xc_ste, void* stream);
int svnet_binhead_fwd_f32(const svnet_binhead_desc* d, void* stream);
int svnet_binhead_bwd_f32(const svnet_binhead_desc* d, void* stream);
/* Backward of y = x W^T + b (nn.Linear, sv_dgcnn_cls.py:80) over M <= 64 rows in one launch: dx [M,K] (may be NULL), dW [O,K],
 * db [O] (may be NULL), all ASSIGNED.                                                                                             */
int svnet_fplinear_small_bwd_f32(const float* g, const float* x, const float* W, int64_t M, int64_t K, int64_t O, float* dx,
                                 float* dW, float* db, void* stream);

/* ------------------------------------------------------------------ label-smoothed cross entropy (utils.py:33-50 cal_loss)
 * logits [R,C], target [R] int64; loss = mean_r -(soft . log_softmax); dlogits = d loss / d logits.
 * workspace: >= 1024 floats (per-workgroup partial losses, added in a fixed order: bit-reproducible).   */
int svnet_smooth_ce_f32(const float* logits, const int64_t* target, int64_t R, int64_t C, float eps, float* loss,
                        float* dlogits, float* workspace, int64_t workspace_floats, void* stream);

/* ------------------------------------------------------------------ optimizer steps on flat buffers (main_cls_dgcnn.py:128-133)
 * p, g, m, v, buf: n floats each (all parameters / gradients of the model, flattened in model.parameters() order).
 * Adam = torch.optim.Adam(lr, betas, eps, weight_decay) at 1-based `step` (bias correction); SGD = torch.optim.SGD(lr,
 * momentum, weight_decay) with buf = g on the first step.                                                   */
int svnet_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int64_t step, void* stream);
int svnet_sgd_step_f32(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float weight_decay,
                       int first_step, void* stream);
/* The same steps with their step-dependent scalars read from DEVICE memory (a launch that can be part of a captured graph; the host
 * refreshes `hyper` before every replay): adam hyper = [lr, beta1, beta2, eps, weight_decay, 1 - beta1^t, 1 / sqrt(1 - beta2^t)],
 * sgd hyper = [lr, momentum, weight_decay, first_step (0 / 1)].                                                                    */
int svnet_adam_step_dev_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, void* stream);
int svnet_sgd_step_dev_f32(float* p, const float* g, float* buf, int64_t n, const float* hyper, void* stream);

/* ------------------------------------------------------------------ diagnostics (no reference counterpart)
 * One thread writes the constant-rate device clock (s_memrealtime: 100 MHz ticks) to *slot when the stream reaches it: the start
 * times of a captured step's launches WITHOUT a profiler (svnet_amd._lib.StepClock, tools/step_clock.py).                          */
int svnet_stamp_u64(uint64_t* slot, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVNET_HIP_H */
