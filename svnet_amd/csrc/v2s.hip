// Vector2Scalar: rotation-invariant scalars from vector channels.
//
// Replaces  models/sv_layers.py:111-129 (Vector2Scalar.forward):
//   z[m,i,j] = sum_c v[m,i,c] * w_eff[j,c]          (F.linear on the last dim, :116; w_eff = scale*sign(W) or W)
//   s[m,c*J+j] = sum_i v[m,i,c] * z[m,i,j]          (einsum, :119-123)
// One group of G lanes owns one row; lane g holds channels g, g+G, g+2G (<= 3 per lane) of the three
// spatial components in registers, the 3xJ matrix z is reduced across the group with xor-shuffles, and
// each lane writes the J contiguous outputs of its channels.  One pass over v, one pass over s: HBM-bound.
#include "common.h"

namespace {

constexpr int J = 3;    // `multi` is 3 in every SV model
// CPL (template): channels per lane, 3 for C <= 192, 6 for C <= 384 (PointNet conv_fuse: Cv = 340), 12 for C <= 768 (the
// 682 / 478 vector channels of sv_pointnet_partseg.py:27,84-89)

template <int G>
__device__ __forceinline__ float group_sum(float v) { return group_sum_dpp<G>(v); }   // (DPP / row swaps: no LDS crossbar)

constexpr int V2S_SUM_ROWS = 4, V2S_SUM_COLS = 8;      // SUM mode: rows per group and workgroup, copied columns per lane (pre_cols <= 8 G)
template <int G, int CPL, bool SUM = false>
__global__ __launch_bounds__(256) void v2s_fwd_kernel(const float* __restrict__ v, const float* __restrict__ w, int64_t M, int C,
                                                      float* __restrict__ s, int64_t s_ld, float* __restrict__ z_out,
                                                      const float* __restrict__ pre, int pre_cols, double* __restrict__ pre_sum = nullptr,
                                                      int64_t rows_per_cloud = 0) {
    // SUM (svnet_v2s_cat_sum_fwd_f32): also the per-cloud column sums of `pre` - the input of the SVBlock's gate MLP (sv_layers.py:179) -
    // from the copy this kernel makes of it anyway.  A workgroup then owns V2S_SUM_ROWS consecutive rows per group (one cloud: checked on
    // the host), a lane adds up the columns it copies, the groups meet in LDS and the workgroup adds ONE fp64 value per column.
    constexpr int GPB = 256 / G;
    const int g = threadIdx.x % G;
    const int64_t group = SUM ? (int64_t)blockIdx.x * (GPB * V2S_SUM_ROWS) + threadIdx.x / G : ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = SUM ? GPB : ((int64_t)gridDim.x * blockDim.x) / G;
    float psum[SUM ? V2S_SUM_COLS : 1];
#pragma unroll
    for (int q = 0; q < (SUM ? V2S_SUM_COLS : 1); ++q) psum[q] = 0.f;
    float wr[J][CPL];
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        const int c = g + G * t;
#pragma unroll
        for (int j = 0; j < J; ++j) wr[j][t] = (c < C) ? w[j * C + c] : 0.f;
    }
    const int64_t iters = SUM ? V2S_SUM_ROWS : (M + ngroups - 1) / ngroups;  // uniform trip count: shuffles need every lane
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t m = group + it * ngroups;
        const bool live = m < M;
        float x[3][CPL];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int t = 0; t < CPL; ++t) {
                const int c = g + G * t;
                x[i][t] = (live && c < C) ? v[(m * 3 + i) * C + c] : 0.f;
            }
        float z[3][J];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                float p = 0.f;
#pragma unroll
                for (int t = 0; t < CPL; ++t) p = fmaf(x[i][t], wr[j][t], p);
                z[i][j] = group_sum<G>(p);
            }
        if (live) {
#pragma unroll
            for (int t = 0; t < CPL; ++t) {
                const int c = g + G * t;
                if (c < C) {
#pragma unroll
                    for (int j = 0; j < J; ++j)
                        s[m * s_ld + c * J + j] = x[0][t] * z[0][j] + x[1][t] * z[1][j] + x[2][t] * z[2][j];
                }
            }
            // cat[pre, s] in place: the row's leading columns are copied by the same lanes (s points pre_cols floats into the row)
            if (SUM) {
#pragma unroll
                for (int q = 0; q < V2S_SUM_COLS; ++q) {
                    const int c = g + G * q;
                    if (c < pre_cols) { const float t = pre[m * pre_cols + c]; s[m * s_ld - pre_cols + c] = t; psum[q] += t; }
                }
            } else if (pre) for (int c = g; c < pre_cols; c += G) s[m * s_ld - pre_cols + c] = pre[m * pre_cols + c];
            if (z_out && g == 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < J; ++j) z_out[(m * 3 + i) * J + j] = z[i][j];
            }
        }
    }
    if constexpr (SUM) {
        __shared__ float part[GPB][G * V2S_SUM_COLS];
        const int gi = threadIdx.x / G;
#pragma unroll
        for (int q = 0; q < V2S_SUM_COLS; ++q) part[gi][g + G * q] = psum[q];
        __syncthreads();
        const int64_t cloud = ((int64_t)blockIdx.x * (GPB * V2S_SUM_ROWS)) / rows_per_cloud;
        for (int c = threadIdx.x; c < pre_cols; c += blockDim.x) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < GPB; ++q) t += part[q][c];
            atomicAdd(&pre_sum[cloud * pre_cols + c], (double)t);
        }
    }
}

template <int G, int CPL>
__global__ __launch_bounds__(256) void v2s_bwd_kernel(const float* __restrict__ v, const float* __restrict__ w,
                                                      const float* __restrict__ ds, int64_t ds_ld, const float* __restrict__ dz_in, int64_t M,
                                                      int C, float* __restrict__ dv, float* __restrict__ GX) {
    __shared__ float gx_lds[J * 768];
    const int g = threadIdx.x % G;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) / G;
    for (int e = threadIdx.x; e < J * C; e += blockDim.x) gx_lds[e] = 0.f;
    __syncthreads();
    float wr[J][CPL], gx[J][CPL];
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        const int c = g + G * t;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            wr[j][t] = (c < C) ? w[j * C + c] : 0.f;
            gx[j][t] = 0.f;
        }
    }
    const int64_t iters = (M + ngroups - 1) / ngroups;
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t m = group + it * ngroups;
        const bool live = m < M;
        float x[3][CPL], d[CPL][J];
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            const int c = g + G * t;
            const bool ok = live && c < C;
#pragma unroll
            for (int i = 0; i < 3; ++i) x[i][t] = ok ? v[(m * 3 + i) * C + c] : 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) d[t][j] = ok ? ds[m * ds_ld + c * J + j] : 0.f;
        }
        float z[3][J], dz[3][J];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                float pz = 0.f, pd = 0.f;
#pragma unroll
                for (int t = 0; t < CPL; ++t) {
                    pz = fmaf(x[i][t], wr[j][t], pz);
                    pd = fmaf(d[t][j], x[i][t], pd);
                }
                z[i][j] = group_sum<G>(pz);
                dz[i][j] = group_sum<G>(pd) + ((live && dz_in) ? dz_in[(m * 3 + i) * J + j] : 0.f);
            }
        if (live) {
#pragma unroll
            for (int t = 0; t < CPL; ++t) {
                const int c = g + G * t;
                if (c < C) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        float a = 0.f;
#pragma unroll
                        for (int j = 0; j < J; ++j) a += d[t][j] * z[i][j] + dz[i][j] * wr[j][t];
                        dv[(m * 3 + i) * C + c] = a;
                    }
#pragma unroll
                    for (int j = 0; j < J; ++j) gx[j][t] += dz[0][j] * x[0][t] + dz[1][j] * x[1][t] + dz[2][j] * x[2][t];
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        const int c = g + G * t;
        if (c < C) {
#pragma unroll
            for (int j = 0; j < J; ++j) atomicAdd(&gx_lds[j * C + c], gx[j][t]);
        }
    }
    __syncthreads();
    // GX: a SLICED accumulator of J * C floats (svnet_hip.h SVNET_SLICED_LEN): 2 048 workgroups adding to the same 3 C addresses were
    // served one after the other at the memory side - ~45 of this kernel's 92 us at C = 170 (conv5's svfuse); the caller adds the
    // slices up (svnet_slices_sum_f32)
    float* gsl = svnet_slice_ptr(GX, J * C);
    for (int e = threadIdx.x; e < J * C; e += blockDim.x) atomicAdd(&gsl[e], gx_lds[e]);
}

// Frame projection (the back-projection einsum of sv_pointnet_partseg.py:89): s[m,c*J+j] = sum_i v[m,i,c] * z[m,i,j] with a GIVEN
// per-row frame z [M,3,J] (Vector2Scalar's second stage on its own).  Backward: dv[m,i,c] = sum_j ds[m,c*J+j] z[m,i,j],
// dz[m,i,j] = sum_c ds[m,c*J+j] v[m,i,c].
template <int G, int CPL>
__global__ __launch_bounds__(256) void vproject_fwd_kernel(const float* __restrict__ v, const float* __restrict__ z, int64_t M, int C,
                                                           float* __restrict__ s) {
    const int g = threadIdx.x % G;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) / G;
    for (int64_t m = group; m < M; m += ngroups) {
        float zz[3][J];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) zz[i][j] = z[(m * 3 + i) * J + j];
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            const int c = g + G * t;
            if (c < C) {
                const float x0 = v[(m * 3 + 0) * C + c], x1 = v[(m * 3 + 1) * C + c], x2 = v[(m * 3 + 2) * C + c];
#pragma unroll
                for (int j = 0; j < J; ++j) s[m * C * J + c * J + j] = x0 * zz[0][j] + x1 * zz[1][j] + x2 * zz[2][j];
            }
        }
    }
}

template <int G, int CPL>
__global__ __launch_bounds__(256) void vproject_bwd_kernel(const float* __restrict__ v, const float* __restrict__ z,
                                                           const float* __restrict__ ds, int64_t M, int C, float* __restrict__ dv,
                                                           float* __restrict__ dz) {
    const int g = threadIdx.x % G;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) / G;
    const int64_t iters = (M + ngroups - 1) / ngroups;  // uniform trip count: shuffles need every lane
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t m = group + it * ngroups;
        const bool live = m < M;
        float zz[3][J], pd[3][J];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                zz[i][j] = live ? z[(m * 3 + i) * J + j] : 0.f;
                pd[i][j] = 0.f;
            }
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            const int c = g + G * t;
            const bool ok = live && c < C;
            float x[3], d[J];
#pragma unroll
            for (int i = 0; i < 3; ++i) x[i] = ok ? v[(m * 3 + i) * C + c] : 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) d[j] = ok ? ds[m * C * J + c * J + j] : 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    a = fmaf(d[j], zz[i][j], a);
                    pd[i][j] = fmaf(d[j], x[i], pd[i][j]);
                }
                if (ok) dv[(m * 3 + i) * C + c] = a;
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float tot = group_sum<G>(pd[i][j]);
                if (live && g == 0) dz[(m * 3 + i) * J + j] = tot;
            }
    }
}

inline unsigned v2s_grid(int64_t M, int G) {
    const int64_t groups_per_block = 256 / G;
    int64_t blocks = svnet_cdiv(M, groups_per_block * 4);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

static int v2s_fwd_launch(const float* v, const float* w_eff, int64_t M, int64_t C, int64_t Jn, float* s, int64_t s_ld, float* z_out,
                          const float* pre, int64_t pre_cols, void* stream, double* pre_sum = nullptr, int64_t rows_per_cloud = 0) {
    SVNET_REQUIRE(v && w_eff && s && M >= 0 && C > 0 && s_ld >= C * J + (pre ? pre_cols : 0) && (!pre || pre_cols > 0), SVNET_E_ARG,
                  "svnet_v2s_fwd_f32: bad arguments");
    SVNET_REQUIRE(Jn == J && C <= 768, SVNET_E_UNSUPPORTED, "svnet_v2s_fwd_f32: needs multi == 3 and C <= 768 (got %lld, %lld)",
                  (long long)Jn, (long long)C);
    if (M == 0) return SVNET_OK;
    hipStream_t st = (hipStream_t)stream;
#define SVNET_V2S(G, CPL) hipLaunchKernelGGL((v2s_fwd_kernel<G, CPL>), dim3(v2s_grid(M, G)), dim3(256), 0, st, v, w_eff, M, (int)C, s, s_ld, z_out, pre, (int)pre_cols)
    if (pre_sum) {
        // (the widths the callers have: conv5 of the DGCNN callers, C = 83 / 80 with 256 leading columns)
        const int G_ = C <= 96 ? 32 : 64;
        SVNET_REQUIRE(C > 24 && C <= 192 && pre && pre_cols <= (int64_t)V2S_SUM_COLS * G_ && rows_per_cloud > 0 && M % rows_per_cloud == 0 &&
                          rows_per_cloud % ((256 / G_) * V2S_SUM_ROWS) == 0, SVNET_E_UNSUPPORTED,
                      "svnet_v2s_cat_sum_fwd_f32: C=%lld, pre_cols=%lld, rows_per_cloud=%lld not supported", (long long)C, (long long)pre_cols,
                      (long long)rows_per_cloud);
        const unsigned blocks = (unsigned)(M / ((256 / G_) * V2S_SUM_ROWS));
        if (G_ == 32) hipLaunchKernelGGL((v2s_fwd_kernel<32, 3, true>), dim3(blocks), dim3(256), 0, st, v, w_eff, M, (int)C, s, s_ld, z_out, pre, (int)pre_cols, pre_sum, rows_per_cloud);
        else hipLaunchKernelGGL((v2s_fwd_kernel<64, 3, true>), dim3(blocks), dim3(256), 0, st, v, w_eff, M, (int)C, s, s_ld, z_out, pre, (int)pre_cols, pre_sum, rows_per_cloud);
        SVNET_CHECK_LAUNCH("v2s_fwd_kernel<sum>");
        return SVNET_OK;
    }
    if (C <= 3) SVNET_V2S(1, 3);
    else if (C <= 24) SVNET_V2S(8, 3);
    else if (C <= 96) SVNET_V2S(32, 3);
    else if (C <= 192) SVNET_V2S(64, 3);
    else if (C <= 384) SVNET_V2S(64, 6);
    else SVNET_V2S(64, 12);
#undef SVNET_V2S
    SVNET_CHECK_LAUNCH("v2s_fwd_kernel");
    return SVNET_OK;
}
extern "C" int svnet_v2s_fwd_f32(const float* v, const float* w_eff, int64_t M, int64_t C, int64_t Jn, float* s, float* z_out,
                                 void* stream) {
    return v2s_fwd_launch(v, w_eff, M, C, Jn, s, C * J, z_out, nullptr, 0, stream);
}
/* out [M, out_ld] = cat[pre (pre_cols), Vector2Scalar(v) (C*J)] written in place: the concatenation feeding an SVBlock's linear1
 * (sv_layers.py:187-188) without the intermediate tensor and the cat pass.                                                         */
extern "C" int svnet_v2s_cat_fwd_f32(const float* v, const float* w_eff, const float* pre, int64_t pre_cols, int64_t M, int64_t C,
                                     int64_t Jn, float* out, int64_t out_ld, void* stream) {
    SVNET_REQUIRE(pre && out && pre_cols > 0, SVNET_E_ARG, "svnet_v2s_cat_fwd_f32: bad arguments");
    return v2s_fwd_launch(v, w_eff, M, C, Jn, out + pre_cols, out_ld, nullptr, pre, pre_cols, stream);
}

extern "C" int svnet_v2s_cat_sum_supported(int64_t M, int64_t C, int64_t pre_cols, int64_t rows_per_cloud) {
    const int64_t G_ = C <= 96 ? 32 : 64;
    return (C > 24 && C <= 192 && pre_cols > 0 && pre_cols <= (int64_t)V2S_SUM_COLS * G_ && rows_per_cloud > 0 && M > 0 && M % rows_per_cloud == 0 &&
            rows_per_cloud % ((256 / G_) * V2S_SUM_ROWS) == 0) ? 1 : 0;
}
/* ... and pre_sum [M / rows_per_cloud, pre_cols] (fp64, caller zero-fills) += the column sums of pre over each cloud's rows: the gate MLP's
 * input (svnet_gate_mlp_fwd_f32 with gin_f64 = pre_sum, in_scale = 1 / rows_per_cloud), from the copy of pre this kernel makes anyway. */
extern "C" int svnet_v2s_cat_sum_fwd_f32(const float* v, const float* w_eff, const float* pre, int64_t pre_cols, int64_t M, int64_t C,
                                         int64_t Jn, float* out, int64_t out_ld, double* pre_sum, int64_t rows_per_cloud, void* stream) {
    SVNET_REQUIRE(pre && out && pre_cols > 0 && pre_sum, SVNET_E_ARG, "svnet_v2s_cat_sum_fwd_f32: bad arguments");
    return v2s_fwd_launch(v, w_eff, M, C, Jn, out + pre_cols, out_ld, nullptr, pre, pre_cols, stream, pre_sum, rows_per_cloud);
}

extern "C" int svnet_v2s_bwd_ld_f32(const float* v, const float* w_eff, const float* ds, int64_t ds_ld, const float* dz_in, int64_t M,
                                    int64_t C, int64_t Jn, float* dv, float* GX, void* stream);
extern "C" int svnet_v2s_bwd_f32(const float* v, const float* w_eff, const float* ds, const float* dz_in, int64_t M, int64_t C,
                                 int64_t Jn, float* dv, float* GX, void* stream) {
    return svnet_v2s_bwd_ld_f32(v, w_eff, ds, C * J, dz_in, M, C, Jn, dv, GX, stream);
}
/* svnet_v2s_bwd_f32 with the gradient rows ds at stride ds_ld (a column slice of a wider gradient: the cat of svnet_v2s_cat_fwd_f32) */
extern "C" int svnet_v2s_bwd_ld_f32(const float* v, const float* w_eff, const float* ds, int64_t ds_ld, const float* dz_in, int64_t M,
                                    int64_t C, int64_t Jn, float* dv, float* GX, void* stream) {
    SVNET_REQUIRE(v && w_eff && ds && dv && GX && M >= 0 && C > 0 && ds_ld >= C * J, SVNET_E_ARG, "svnet_v2s_bwd_f32: bad arguments");
    SVNET_REQUIRE(Jn == J && C <= 768, SVNET_E_UNSUPPORTED, "svnet_v2s_bwd_f32: needs multi == 3 and C <= 768 (got %lld, %lld)",
                  (long long)Jn, (long long)C);
    if (M == 0) return SVNET_OK;
    hipStream_t st = (hipStream_t)stream;
#define SVNET_V2S(G, CPL) hipLaunchKernelGGL((v2s_bwd_kernel<G, CPL>), dim3(v2s_grid(M, G)), dim3(256), 0, st, v, w_eff, ds, ds_ld, dz_in, M, (int)C, dv, GX)
    if (C <= 3) SVNET_V2S(1, 3);
    else if (C <= 24) SVNET_V2S(8, 3);
    else if (C <= 96) SVNET_V2S(32, 3);
    else if (C <= 192) SVNET_V2S(64, 3);
    else if (C <= 384) SVNET_V2S(64, 6);
    else SVNET_V2S(64, 12);
#undef SVNET_V2S
    SVNET_CHECK_LAUNCH("v2s_bwd_kernel");
    return SVNET_OK;
}

extern "C" int svnet_vproject_fwd_f32(const float* v, const float* z, int64_t M, int64_t C, int64_t Jn, float* s, void* stream) {
    SVNET_REQUIRE(v && z && s && M >= 0 && C > 0, SVNET_E_ARG, "svnet_vproject_fwd_f32: bad arguments");
    SVNET_REQUIRE(Jn == J && C <= 768, SVNET_E_UNSUPPORTED, "svnet_vproject_fwd_f32: needs multi == 3 and C <= 768 (got %lld, %lld)",
                  (long long)Jn, (long long)C);
    if (M == 0) return SVNET_OK;
    hipStream_t st = (hipStream_t)stream;
#define SVNET_VP(G, CPL) hipLaunchKernelGGL((vproject_fwd_kernel<G, CPL>), dim3(v2s_grid(M, G)), dim3(256), 0, st, v, z, M, (int)C, s)
    if (C <= 24) SVNET_VP(8, 3);
    else if (C <= 96) SVNET_VP(32, 3);
    else if (C <= 192) SVNET_VP(64, 3);
    else if (C <= 384) SVNET_VP(64, 6);
    else SVNET_VP(64, 12);
#undef SVNET_VP
    SVNET_CHECK_LAUNCH("vproject_fwd_kernel");
    return SVNET_OK;
}

extern "C" int svnet_vproject_bwd_f32(const float* v, const float* z, const float* ds, int64_t M, int64_t C, int64_t Jn, float* dv,
                                      float* dz, void* stream) {
    SVNET_REQUIRE(v && z && ds && dv && dz && M >= 0 && C > 0, SVNET_E_ARG, "svnet_vproject_bwd_f32: bad arguments");
    SVNET_REQUIRE(Jn == J && C <= 768, SVNET_E_UNSUPPORTED, "svnet_vproject_bwd_f32: needs multi == 3 and C <= 768 (got %lld, %lld)",
                  (long long)Jn, (long long)C);
    if (M == 0) return SVNET_OK;
    hipStream_t st = (hipStream_t)stream;
#define SVNET_VP(G, CPL) hipLaunchKernelGGL((vproject_bwd_kernel<G, CPL>), dim3(v2s_grid(M, G)), dim3(256), 0, st, v, z, ds, M, (int)C, dv, dz)
    if (C <= 24) SVNET_VP(8, 3);
    else if (C <= 96) SVNET_VP(32, 3);
    else if (C <= 192) SVNET_VP(64, 3);
    else if (C <= 384) SVNET_VP(64, 6);
    else SVNET_VP(64, 12);
#undef SVNET_VP
    SVNET_CHECK_LAUNCH("vproject_bwd_kernel");
    return SVNET_OK;
}
