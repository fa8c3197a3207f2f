// The tail of the classifier's vector path as ONE pass over linear2's product (conv5 of sv_dgcnn_cls, 32 768 points x 3 x 170):
//   VectorBN (+ gate)        models/sv_layers.py:86-102,193-194      v5 = v / (|v| + EPS) * BN(|v| + EPS) * gate
//   Vector2Scalar (svfuse)   models/sv_layers.py:111-129,206-220     s_v[c*3+j] = sum_i v5[i,c] * (v5 . w_eff^T)[i,j]
//   [max | mean] over points models/sv_dgcnn_cls.py:70-74            adaptive max / avg pool of the s_v half of the [B,N,1022] feature
// Layer by layer this was: VectorBN read + write (134 MB), Vector2Scalar read + write (134 MB), pooling read (67 MB); its backward a
// broadcast of the pooled gradient (67 MB written), Vector2Scalar's backward (200 MB), VectorBN's reduce pass (134 MB).  Neither v5 nor
// s_v [P,510] nor the gradient of s_v is needed by anything else (sv_dgcnn_cls.py:69-74 pools the feature at once), so here
//   forward : ONE read of the product; v5 and s_v live in registers; packed arg-max keys + ordered partial sums per (cloud, row chunk);
//   backward: ONE read of the product, v5 / z recomputed, the pooled gradient (point == arg-max ? g_max : 0) + g_mean / N formed on the
//             fly, Vector2Scalar's backward in registers, VectorBN's batch sums and the gate's gradient accumulated, dL/dv5 written once
//             for the (unchanged) VectorBN apply pass.
// A wave owns a point: lane g holds channels g, g + 64, g + 128 (C <= 192) of the three axes; the 3 x 3 frame z is nine wave sums.
// Same arithmetic, in the same order, as vbn_fwd_kernel (norm.hip) and v2s_fwd_kernel (v2s.hip): the pooled maxima equal the layer-wise chain's.
#include "common.h"

namespace {

constexpr int J = 3;
constexpr float VT_VEPS = 1e-6f;      // EPS of sv_layers.py:18, added to the norm before BatchNorm (:94)

__device__ __forceinline__ unsigned long long vt_pack_key(float v, uint32_t r) {     // pool.hip pack_key: larger value, then lower row
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - r);
}

struct VtStats {        // training: the sliced fp64 sums of svnet_colstats_f64(kind 1); every lane derives its channels' statistics itself
    const double* sums; float* mean_out; float* invstd_out; float* rmean; float* rvar; long long* nbt;
    const float* mean_in; const float* invstd_in;      // eval (sums == nullptr): the running statistics' mean / invstd
    float eps, momentum;
};

template <int CPL>
struct VtChan {          // per-lane constants of its CPL channels
    float mu[CPL], is[CPL], ga[CPL], be[CPL], gt[CPL], w[J][CPL];
    bool ok[CPL];
};

template <int CPL>
__device__ __forceinline__ void vt_load_chan(VtChan<CPL>& ch, const VtStats& st, const float* gamma, const float* beta, const float* gate,
                                             const float* w, int64_t cloud, int64_t M, int C, int lane, bool keeper) {
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        const int c = lane + 64 * t;
        ch.ok[t] = c < C;
        const int cc = ch.ok[t] ? c : 0;
        if (st.sums) {       // bn_finalize_kernel's arithmetic (norm.hip)
            const double m = svnet_slices_total(st.sums, 2 * C, cc) / (double)M;
            double var = svnet_slices_total(st.sums, 2 * C, C + cc) / (double)M - m * m;
            if (var < 0.0) var = 0.0;
            ch.mu[t] = (float)m;
            ch.is[t] = (float)(1.0 / sqrt(var + (double)st.eps));
            if (keeper && ch.ok[t]) {
                st.mean_out[c] = ch.mu[t];
                st.invstd_out[c] = ch.is[t];
                if (st.rmean) st.rmean[c] = (1.f - st.momentum) * st.rmean[c] + st.momentum * (float)m;
                if (st.rvar) {
                    const double unb = (M > 1) ? var * ((double)M / (double)(M - 1)) : var;
                    st.rvar[c] = (1.f - st.momentum) * st.rvar[c] + st.momentum * (float)unb;
                }
            }
        } else {
            ch.mu[t] = st.mean_in[cc];
            ch.is[t] = st.invstd_in[cc];
        }
        ch.ga[t] = gamma[cc];
        ch.be[t] = beta[cc];
        ch.gt[t] = gate ? gate[cloud * C + cc] : 1.f;
#pragma unroll
        for (int j = 0; j < J; ++j) ch.w[j][t] = ch.ok[t] ? w[j * C + c] : 0.f;
    }
}

// raw product rows of point m -> registers (dead channels read channel 0 and are masked where they are consumed)
template <int CPL>
__device__ __forceinline__ void vt_load_row(float (&a)[3][CPL], const float* __restrict__ v, int64_t m, int C, int lane) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            const int c = lane + 64 * t;
            a[i][t] = v[(m * 3 + i) * C + (c < C ? c : 0)];
        }
}

// VectorBN + gate of one point's channels (vbn_fwd_kernel's expressions) and the frame z of Vector2Scalar (v2s_fwd_kernel's)
template <int CPL>
__device__ __forceinline__ void vt_point(const float (&a)[3][CPL], const VtChan<CPL>& ch, float (&x)[3][CPL], float (&n)[CPL], float (&rr)[CPL],
                                         float (&z)[3][J]) {
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        n[t] = sqrtf(a[0][t] * a[0][t] + a[1][t] * a[1][t] + a[2][t] * a[2][t]) + VT_VEPS;
        rr[t] = (n[t] - ch.mu[t]) * ch.is[t] * ch.ga[t] + ch.be[t];
#pragma unroll
        for (int i = 0; i < 3; ++i) x[i][t] = ch.ok[t] ? a[i][t] / n[t] * rr[t] * ch.gt[t] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            float p = 0.f;
#pragma unroll
            for (int t = 0; t < CPL; ++t) p = fmaf(x[i][t], ch.w[j][t], p);
            z[i][j] = group_sum_dpp<64>(p);
        }
}

// ---- forward: grid (row chunks, clouds), 4 waves; wave w takes points r0 + w, r0 + w + 4, ... of the chunk
template <int CPL>
__global__ __launch_bounds__(256) void vtail_fwd_kernel(const float* __restrict__ v, VtStats st, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ gate,
                                                        const float* __restrict__ w, int N, int C, int rows_per_chunk,
                                                        unsigned long long* __restrict__ keys, float* __restrict__ part, int64_t total) {
    constexpr int NO = J * CPL;                          // outputs per lane
    __shared__ unsigned long long lkey[4][NO * 64];
    __shared__ float lsum[4][NO * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t cloud = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_chunk, r1 = min(N, r0 + rows_per_chunk);
    const int64_t M = (int64_t)gridDim.y * N;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && st.sums && st.nbt) *st.nbt += 1;
    VtChan<CPL> ch;
    vt_load_chan<CPL>(ch, st, gamma, beta, gate, w, cloud, M, C, lane, blockIdx.x == 0 && blockIdx.y == 0 && wave == 0);

    float best[CPL][J], sum[CPL][J];
    uint32_t bi[CPL][J];
#pragma unroll
    for (int t = 0; t < CPL; ++t)
#pragma unroll
        for (int j = 0; j < J; ++j) { best[t][j] = -INFINITY; sum[t][j] = 0.f; bi[t][j] = 0xFFFFFFFFu; }

    float a[3][CPL], an[3][CPL];
    int r = r0 + wave;
    if (r < r1) vt_load_row<CPL>(a, v, cloud * N + r, C, lane);
    for (; r < r1; r += 4) {
        const int rn = min(r + 4, N - 1);                  // next point of this wave, requested before the current one is consumed (clamped)
        vt_load_row<CPL>(an, v, cloud * N + rn, C, lane);
        __builtin_amdgcn_sched_barrier(0);
        float x[3][CPL], n[CPL], rr[CPL], z[3][J];
        vt_point<CPL>(a, ch, x, n, rr, z);
#pragma unroll
        for (int t = 0; t < CPL; ++t)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float s = x[0][t] * z[0][j] + x[1][t] * z[1][j] + x[2][t] * z[2][j];
                sum[t][j] += s;
                if (s > best[t][j] || bi[t][j] == 0xFFFFFFFFu) { best[t][j] = s; bi[t][j] = (uint32_t)r; }   // strict '>': first index
            }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int t = 0; t < CPL; ++t) a[i][t] = an[i][t];
    }
    // the four waves' results of a column: larger key (value, then lower row) and the sums in wave order (bit-reproducible)
#pragma unroll
    for (int t = 0; t < CPL; ++t)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int slot = (t * J + j) * 64 + lane;
            lkey[wave][slot] = bi[t][j] == 0xFFFFFFFFu ? 0ull : vt_pack_key(best[t][j], bi[t][j]);
            lsum[wave][slot] = sum[t][j];
        }
    __syncthreads();
    for (int e = threadIdx.x; e < NO * 64; e += 256) {
        const int ln = e & 63, tj = e >> 6;
        const int t = tj / J, j = tj - t * J;
        const int c = ln + 64 * t;
        if (c < C) {
            unsigned long long kk = lkey[0][e];
            float s = lsum[0][e];
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) {
                kk = max(kk, lkey[ww][e]);
                s += lsum[ww][e];
            }
            const int64_t o = cloud * (int64_t)(J * C) + c * J + j;
            atomicMax(&keys[o], kk);
            part[(int64_t)blockIdx.x * total + o] = s;
        }
    }
}

// unpack of the keys + ordered finish of the mean (pool.hip pool_maxmean_finish_kernel)
__global__ __launch_bounds__(256) void vtail_finish_kernel(const unsigned long long* __restrict__ keys, const float* __restrict__ part,
                                                           int64_t chunks, int64_t total, float invR, float* __restrict__ out_max,
                                                           float* __restrict__ out_mean, int32_t* __restrict__ argmax, int64_t inner,
                                                           int64_t out_ld) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long kk = keys[e];
        uint32_t u = (uint32_t)(kk >> 32);
        u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
        float s = 0.f;
        for (int64_t c = 0; c < chunks; ++c) s += part[c * total + e];
        const int64_t o = e / inner, i = e - o * inner;
        out_max[o * out_ld + i] = __uint_as_float(u);
        out_mean[o * out_ld + i] = s * invR;
        argmax[e] = (int32_t)(0xFFFFFFFFu - (uint32_t)(kk & 0xFFFFFFFFull));
    }
}

// ---- backward, first pass: everything up to VectorBN's batch sums; writes g5 = dL/dv5 for the apply pass (vbn_bwd_apply_kernel)
template <int CPL>
__global__ __launch_bounds__(256) void vtail_bwd_kernel(const float* __restrict__ v, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ gate, const float* __restrict__ w,
                                                        const float* __restrict__ gmax, const float* __restrict__ gmean, int64_t g_ld,
                                                        const int32_t* __restrict__ argmax, int N, int C, int rows_per_chunk,
                                                        float* __restrict__ red, float* __restrict__ dgate, float* __restrict__ GX,
                                                        float* __restrict__ g5) {
    constexpr int NO = J * CPL;
    __shared__ double lacc[4][2 * CPL * 64];
    __shared__ float lgw[4][(NO + CPL) * 64];            // [wave][gxw (J*CPL) | gsum (CPL)][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t cloud = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_chunk, r1 = min(N, r0 + rows_per_chunk);
    VtStats st{};
    st.mean_in = mean; st.invstd_in = invstd;
    VtChan<CPL> ch;
    vt_load_chan<CPL>(ch, st, gamma, beta, gate, w, cloud, 0, C, lane, false);
    const float invR = 1.f / (float)N;
    float gx[CPL][J], gm[CPL][J];
    int am[CPL][J];
#pragma unroll
    for (int t = 0; t < CPL; ++t)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int c = lane + 64 * t;
            const int o = (ch.ok[t] ? c : 0) * J + j;
            gx[t][j] = ch.ok[t] ? gmax[cloud * g_ld + o] : 0.f;
            gm[t][j] = ch.ok[t] ? gmean[cloud * g_ld + o] * invR : 0.f;
            am[t][j] = ch.ok[t] ? argmax[cloud * (int64_t)(J * C) + o] : -1;
        }
    float gxw[J][CPL], gsum[CPL];
    double acc0[CPL], acc1[CPL];
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        gsum[t] = 0.f; acc0[t] = 0.0; acc1[t] = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j) gxw[j][t] = 0.f;
    }
    float a[3][CPL], an[3][CPL];
    int r = r0 + wave;
    if (r < r1) vt_load_row<CPL>(a, v, cloud * N + r, C, lane);
    for (; r < r1; r += 4) {
        const int rn = min(r + 4, N - 1);
        vt_load_row<CPL>(an, v, cloud * N + rn, C, lane);
        __builtin_amdgcn_sched_barrier(0);
        float x[3][CPL], n[CPL], rr[CPL], z[3][J];
        vt_point<CPL>(a, ch, x, n, rr, z);
        // pooled gradient of this point's s_v, Vector2Scalar's backward (v2s_bwd_kernel's expressions)
        float d[CPL][J], dz[3][J];
#pragma unroll
        for (int t = 0; t < CPL; ++t)
#pragma unroll
            for (int j = 0; j < J; ++j) d[t][j] = gm[t][j] + (am[t][j] == r ? gx[t][j] : 0.f);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                float pd = 0.f;
#pragma unroll
                for (int t = 0; t < CPL; ++t) pd = fmaf(d[t][j], x[i][t], pd);
                dz[i][j] = group_sum_dpp<64>(pd);
            }
        const int64_t m = cloud * N + r;
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            float g[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < J; ++j) s += d[t][j] * z[i][j] + dz[i][j] * ch.w[j][t];
                g[i] = s;
            }
#pragma unroll
            for (int j = 0; j < J; ++j) gxw[j][t] += dz[0][j] * x[0][t] + dz[1][j] * x[1][t] + dz[2][j] * x[2][t];
            if (ch.ok[t]) {
                const int c = lane + 64 * t;
#pragma unroll
                for (int i = 0; i < 3; ++i) g5[(m * 3 + i) * C + c] = g[i];
                // VectorBN's reduce pass on g = dL/dv5 (vbn_bwd_reduce_kernel's expressions)
                const float nh = (n[t] - ch.mu[t]) * ch.is[t];
                const float gv = g[0] * a[0][t] + g[1] * a[1][t] + g[2] * a[2][t];
                gsum[t] += gv * (rr[t] / n[t]);
                const float dr = gv * ch.gt[t] / n[t];
                acc0[t] += (double)dr;
                acc1[t] += (double)dr * (double)nh;
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int t = 0; t < CPL; ++t) a[i][t] = an[i][t];
    }
    // workgroup sums in wave order, then ONE add per output and workgroup
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        lacc[wave][(2 * t) * 64 + lane] = acc0[t];
        lacc[wave][(2 * t + 1) * 64 + lane] = acc1[t];
        lgw[wave][(NO + t) * 64 + lane] = gsum[t];
#pragma unroll
        for (int j = 0; j < J; ++j) lgw[wave][(j * CPL + t) * 64 + lane] = gxw[j][t];
    }
    __syncthreads();
    float* rsl = svnet_slice_ptr(red, 2 * C);
    float* gsl = svnet_slice_ptr(GX, J * C);
    for (int e = threadIdx.x; e < (NO + CPL) * 64; e += 256) {
        const int ln = e & 63, q = e >> 6;
        float s = lgw[0][e] + lgw[1][e] + lgw[2][e] + lgw[3][e];
        if (q < NO) {
            const int j = q / CPL, t = q - j * CPL, c = ln + 64 * t;
            if (c < C) svnet_slice_add(&gsl[j * C + c], s);
        } else {
            const int c = ln + 64 * (q - NO);
            if (c < C && dgate) atomicAdd(&dgate[cloud * C + c], s);
        }
    }
    for (int e = threadIdx.x; e < 2 * CPL * 64; e += 256) {
        const int ln = e & 63, q = e >> 6;
        const int t = q >> 1, c = ln + 64 * t;
        if (c < C) {
            const double s = lacc[0][e] + lacc[1][e] + lacc[2][e] + lacc[3][e];
            svnet_slice_add(&rsl[(q & 1) * C + c], (float)s);
        }
    }
}

inline void vtail_chunks(int64_t B, int64_t N, int64_t& chunks, int64_t& rpc) {
    chunks = svnet_cdiv(256 * 8, B);                      // ~8 workgroups per CU
    if (chunks > svnet_cdiv(N, 16)) chunks = svnet_cdiv(N, 16);
    if (chunks < 1) chunks = 1;
    rpc = svnet_cdiv(svnet_cdiv(N, chunks), 4) * 4;
    chunks = svnet_cdiv(N, rpc);
}

}  // namespace

extern "C" size_t svnet_vtail_workspace_bytes(int64_t B, int64_t N, int64_t C) {
    if (B <= 0 || N <= 0 || C <= 0) return 0;
    int64_t chunks, rpc;
    vtail_chunks(B, N, chunks, rpc);
    return (size_t)(B * 3 * C) * 8 + (size_t)(chunks * B * 3 * C) * sizeof(float);
}

extern "C" int svnet_vtail_fwd_f32(const float* v, const double* sums, float eps, float momentum, float* mean, float* invstd,
                                   float* running_mean, float* running_var, long long* nbt, const float* gamma, const float* beta,
                                   const float* gate, const float* w_eff, int64_t B, int64_t N, int64_t C, float* out_max,
                                   float* out_mean, int64_t out_ld, int32_t* argmax, void* workspace, size_t workspace_bytes, void* stream) {
    SVNET_REQUIRE(v && mean && invstd && gamma && beta && w_eff && out_max && out_mean && argmax && workspace, SVNET_E_ARG,
                  "svnet_vtail_fwd_f32: null pointer");
    SVNET_REQUIRE(B > 0 && B <= 65535 && N > 0 && N < (1 << 30) && C > 0 && out_ld >= 3 * C, SVNET_E_ARG, "svnet_vtail_fwd_f32: bad sizes");
    SVNET_REQUIRE(C <= 192, SVNET_E_UNSUPPORTED, "svnet_vtail_fwd_f32: C=%lld > 192 vector channels", (long long)C);
    SVNET_REQUIRE(workspace_bytes >= svnet_vtail_workspace_bytes(B, N, C), SVNET_E_ARG, "svnet_vtail_fwd_f32: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int64_t chunks, rpc;
    vtail_chunks(B, N, chunks, rpc);
    const int64_t total = B * 3 * C;
    unsigned long long* keys = (unsigned long long*)workspace;
    float* part = (float*)((char*)workspace + (size_t)total * 8);
    hipError_t e = hipMemsetAsync(keys, 0, (size_t)total * 8, st);
    SVNET_REQUIRE(e == hipSuccess, SVNET_E_LAUNCH, "svnet_vtail_fwd_f32: memset failed");
    VtStats s{};
    s.sums = sums; s.mean_out = mean; s.invstd_out = invstd; s.rmean = running_mean; s.rvar = running_var; s.nbt = nbt;
    s.mean_in = mean; s.invstd_in = invstd; s.eps = eps; s.momentum = momentum;
    const dim3 grid((unsigned)chunks, (unsigned)B);
    if (C <= 64) hipLaunchKernelGGL((vtail_fwd_kernel<1>), grid, dim3(256), 0, st, v, s, gamma, beta, gate, w_eff, (int)N, (int)C, (int)rpc, keys, part, total);
    else if (C <= 128) hipLaunchKernelGGL((vtail_fwd_kernel<2>), grid, dim3(256), 0, st, v, s, gamma, beta, gate, w_eff, (int)N, (int)C, (int)rpc, keys, part, total);
    else hipLaunchKernelGGL((vtail_fwd_kernel<3>), grid, dim3(256), 0, st, v, s, gamma, beta, gate, w_eff, (int)N, (int)C, (int)rpc, keys, part, total);
    SVNET_CHECK_LAUNCH("vtail_fwd_kernel");
    hipLaunchKernelGGL(vtail_finish_kernel, dim3(svnet_grid(total, 256)), dim3(256), 0, st, keys, part, chunks, total, 1.f / (float)N, out_max,
                       out_mean, argmax, 3 * C, out_ld);
    SVNET_CHECK_LAUNCH("vtail_finish_kernel");
    return SVNET_OK;
}

extern "C" int svnet_vtail_bwd_f32(const float* v, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                   const float* gate, const float* w_eff, const float* gmax, const float* gmean, int64_t g_ld,
                                   const int32_t* argmax, int64_t B, int64_t N, int64_t C, float* red, float* dgate, float* GX, float* g5,
                                   void* stream) {
    SVNET_REQUIRE(v && mean && invstd && gamma && beta && w_eff && gmax && gmean && argmax && red && GX && g5, SVNET_E_ARG,
                  "svnet_vtail_bwd_f32: null pointer");
    SVNET_REQUIRE(B > 0 && B <= 65535 && N > 0 && N < (1 << 30) && C > 0 && g_ld >= 3 * C && (!gate || dgate), SVNET_E_ARG,
                  "svnet_vtail_bwd_f32: bad sizes");
    SVNET_REQUIRE(C <= 192, SVNET_E_UNSUPPORTED, "svnet_vtail_bwd_f32: C=%lld > 192 vector channels", (long long)C);
    hipStream_t st = (hipStream_t)stream;
    int64_t chunks, rpc;
    vtail_chunks(B, N, chunks, rpc);
    const dim3 grid((unsigned)chunks, (unsigned)B);
#define SVNET_VT_BWD(CPL_) hipLaunchKernelGGL((vtail_bwd_kernel<CPL_>), grid, dim3(256), 0, st, v, mean, invstd, gamma, beta, gate, w_eff, gmax, gmean, g_ld, \
                                              argmax, (int)N, (int)C, (int)rpc, red, dgate, GX, g5)
    if (C <= 64) SVNET_VT_BWD(1);
    else if (C <= 128) SVNET_VT_BWD(2);
    else SVNET_VT_BWD(3);
#undef SVNET_VT_BWD
    SVNET_CHECK_LAUNCH("vtail_bwd_kernel");
    return SVNET_OK;
}
