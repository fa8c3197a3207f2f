// The tail of the classifier's vector path as ONE pass over linear2's product (conv5 of sv_dgcnn_cls, 32 768 points x 3 x 170):
//   VectorBN (+ gate)        models/sv_layers.py:86-102,193-194      v5 = v / (|v| + EPS) * BN(|v| + EPS) * gate
//   Vector2Scalar (svfuse)   models/sv_layers.py:111-129,206-220     s_v[c*3+j] = sum_i v5[i,c] * (v5 . w_eff^T)[i,j]
//   [max | mean] over points models/sv_dgcnn_cls.py:70-74            adaptive max / avg pool of the s_v half of the [B,N,1022] feature
// Layer by layer this was: VectorBN read + write (134 MB), Vector2Scalar read + write (134 MB), pooling read (67 MB); its backward a
// broadcast of the pooled gradient (67 MB written), Vector2Scalar's backward (200 MB), VectorBN's reduce pass (134 MB).  Neither v5 nor
// s_v [P,510] nor the gradient of s_v is needed by anything else (sv_dgcnn_cls.py:69-74 pools the feature at once), so here
//   forward : ONE read of the product; v5 and s_v live in registers; packed arg-max keys + ordered partial sums per (cloud, row chunk);
//   backward: ONE read of the product, v5 / z recomputed, the pooled gradient (point == arg-max ? g_max : 0) + g_mean / N formed on the
//             fly, Vector2Scalar's backward in registers, VectorBN's batch sums and the gate's gradient accumulated; a second pass
//             (APPLY) makes the same recomputation and applies VectorBN's backward with the totals of those sums - dL/dv written once,
//             dL/dv5 never (svnet_vtail_bwd_apply_f32; with g5 given, the first pass stores it for svnet_vbn_bwd_apply_f32 instead).
// A wave owns a point: lane g holds channels g, g + 64, g + 128 (C <= 192) of the three axes; the 3 x 3 frame z is nine wave sums.
// The formulas are those of vbn_fwd_kernel (norm.hip) and v2s_fwd_kernel (v2s.hip) with ONE division per channel (rr / n * gate, applied
// to the three axes) and other contractions: values agree with the layer-wise chain to rounding (2e-5 of a tensor's largest element,
// arg-max equal but for near-ties: tests/test_hip_fused.py), not bit for bit.
#include "common.h"

namespace {

constexpr int J = 3;
constexpr float VT_VEPS = 1e-6f;      // EPS of sv_layers.py:18, added to the norm before BatchNorm (:94)

__device__ __forceinline__ unsigned long long vt_pack_key(float v, uint32_t r) {     // pool.hip pack_key: larger value, then lower row
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - r);
}

// ---- nine wave-wide sums per point (the 3 x 3 frame): eight of them packed (18 VALU: permlane swaps fold two quantities per step, as
// edgeblock_bwd.hip wave_sum8_packed) + one plain, results as WAVE-UNIFORM scalars (v_readlane) - nine separate 6-step DPP chains were
// most of the kernel's dependent latency
template <int CTRL>
__device__ __forceinline__ float vt_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float vt_fold32(float a, float b) {      // lanes 0-31: sum of a's halves, lanes 32-63: of b's
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float vt_fold16(float a, float b) {      // rows: a.r0+a.r1 | b.r0+b.r1 | a.r2+a.r3 | b.r2+b.r3
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float vt_lane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ void vt_sum9(const float (&p)[3][J], int lane, float (&z)[3][J]) {
    const float x0 = vt_fold16(vt_fold32(p[0][0], p[0][1]), vt_fold32(p[0][2], p[1][0]));   // rows: s0 | s2 | s1 | s3
    const float x1 = vt_fold16(vt_fold32(p[1][1], p[1][2]), vt_fold32(p[2][0], p[2][1]));   // rows: s4 | s6 | s5 | s7
    const bool hi = (lane & 8) != 0;
    float y = (hi ? x1 : x0) + vt_dpp<0x128>(hi ? x0 : x1);                                  // row_ror:8 - 8-lane groups: s0 s4 s2 s6 s1 s5 s3 s7
    y += vt_dpp<0xB1>(y);
    y += vt_dpp<0x4E>(y);
    y += vt_dpp<0x141>(y);
    z[0][0] = vt_lane(y, 0);  z[1][1] = vt_lane(y, 8);  z[0][2] = vt_lane(y, 16); z[2][0] = vt_lane(y, 24);
    z[0][1] = vt_lane(y, 32); z[1][2] = vt_lane(y, 40); z[1][0] = vt_lane(y, 48); z[2][1] = vt_lane(y, 56);
    z[2][2] = vt_lane(group_sum_dpp<64>(p[2][2]), 0);
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

// (training: wave 0 of the workgroup derives the statistics from the 2 x 16 fp64 slices of its channels - 96 loads, an fp64 division and
//  square root per lane - and hands them to the other waves through `lstat` [2][64 * CPL] floats of LDS; ends on a barrier)
template <int CPL>
__device__ __forceinline__ void vt_load_chan(VtChan<CPL>& ch, const VtStats& st, const float* gamma, const float* beta, const float* gate,
                                             const float* w, int64_t cloud, int64_t M, int C, int lane, int wave, bool keeper, float* lstat) {
    if (st.sums) {
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < CPL; ++t) {
                const int c = lane + 64 * t;
                const int cc = c < C ? c : 0;
                // bn_finalize_kernel's arithmetic (norm.hip)
                const double m = svnet_slices_total(st.sums, 2 * C, cc) / (double)M;
                double var = svnet_slices_total(st.sums, 2 * C, C + cc) / (double)M - m * m;
                if (var < 0.0) var = 0.0;
                const float mu = (float)m, is = (float)(1.0 / sqrt(var + (double)st.eps));
                lstat[c] = mu;
                lstat[64 * CPL + c] = is;
                if (keeper && c < C) {
                    st.mean_out[c] = mu;
                    st.invstd_out[c] = is;
                    if (st.rmean) st.rmean[c] = (1.f - st.momentum) * st.rmean[c] + st.momentum * (float)m;
                    if (st.rvar) {
                        const double unb = (M > 1) ? var * ((double)M / (double)(M - 1)) : var;
                        st.rvar[c] = (1.f - st.momentum) * st.rvar[c] + st.momentum * (float)unb;
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        const int c = lane + 64 * t;
        ch.ok[t] = c < C;
        const int cc = ch.ok[t] ? c : 0;
        if (st.sums) {
            ch.mu[t] = lstat[c];
            ch.is[t] = lstat[64 * CPL + c];
        } else {
            ch.mu[t] = st.mean_in[cc];
            ch.is[t] = st.invstd_in[cc];
        }
        ch.ga[t] = gamma[cc];
        ch.be[t] = beta[cc];
        ch.gt[t] = ch.ok[t] ? (gate ? gate[cloud * C + cc] : 1.f) : 0.f;     // (a dead channel's gate and weights are 0: all it adds is 0)
#pragma unroll
        for (int j = 0; j < J; ++j) ch.w[j][t] = ch.ok[t] ? w[j * C + c] : 0.f;
        // (pinned in registers: left alone, the compiler re-loads these per-channel constants from memory inside the point loop - 24
        //  loads per two points whose waits sit in the middle of the dependent chain - to save the registers)
        asm volatile("" : "+v"(ch.mu[t]), "+v"(ch.is[t]), "+v"(ch.ga[t]), "+v"(ch.be[t]), "+v"(ch.gt[t]));
        asm volatile("" : "+v"(ch.w[0][t]), "+v"(ch.w[1][t]), "+v"(ch.w[2][t]));
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
// (one division per channel: v * (BN(n) / n * gate) - vbn_fwd_kernel divides every component, v / n * BN(n) * gate: the same value to an ulp)
template <int CPL>
__device__ __forceinline__ void vt_point(const float (&a)[3][CPL], const VtChan<CPL>& ch, float (&x)[3][CPL], float (&n)[CPL], float (&rr)[CPL],
                                         float (&z)[3][J], int lane, float* nv_out = nullptr) {
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        const float nv = sqrtf(a[0][t] * a[0][t] + a[1][t] * a[1][t] + a[2][t] * a[2][t]);
        if (nv_out) nv_out[t] = nv;
        n[t] = nv + VT_VEPS;
        rr[t] = (n[t] - ch.mu[t]) * ch.is[t] * ch.ga[t] + ch.be[t];
        const float q = rr[t] / n[t] * ch.gt[t];        // (no branch around the division: a dead channel reads channel 0, n > 0, gate 0)
#pragma unroll
        for (int i = 0; i < 3; ++i) x[i][t] = a[i][t] * q;
    }
    float p[3][J];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            p[i][j] = 0.f;
#pragma unroll
            for (int t = 0; t < CPL; ++t) p[i][j] = fmaf(x[i][t], ch.w[j][t], p[i][j]);
        }
    vt_sum9(p, lane, z);
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
    __shared__ float lstat[2 * 64 * CPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t cloud = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_chunk, r1 = min(N, r0 + rows_per_chunk);
    const int64_t M = (int64_t)gridDim.y * N;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && st.sums && st.nbt) *st.nbt += 1;
    VtChan<CPL> ch;
    vt_load_chan<CPL>(ch, st, gamma, beta, gate, w, cloud, M, C, lane, wave, blockIdx.x == 0 && blockIdx.y == 0, lstat);

    float best[CPL][J], sum[CPL][J];
    uint32_t bi[CPL][J];
    const bool any = r0 + wave < r1;
#pragma unroll
    for (int t = 0; t < CPL; ++t)
#pragma unroll
        for (int j = 0; j < J; ++j) { best[t][j] = -INFINITY; sum[t][j] = 0.f; bi[t][j] = (uint32_t)(r0 + wave); }

    // one point: VectorBN + gate, frame, the 3 CPL scalars of this lane, running [max | sum].  `live` false (a clamped request past the
    // chunk's end) leaves the accumulators alone.  The body has no branch: basic-block boundaries made the waitcnt pass drain the requests.
    auto point = [&](const float (&a)[3][CPL], int r, bool live) {
        float x[3][CPL], n[CPL], rr[CPL], z[3][J];
        vt_point<CPL>(a, ch, x, n, rr, z, lane);
#pragma unroll
        for (int t = 0; t < CPL; ++t)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float sv = x[0][t] * z[0][j] + x[1][t] * z[1][j] + x[2][t] * z[2][j];
                sum[t][j] += live ? sv : 0.f;
                const bool up = live && sv > best[t][j];                  // strict '>': the first index keeps a tie
                best[t][j] = up ? sv : best[t][j];
                bi[t][j] = up ? (uint32_t)r : bi[t][j];
            }
    };
    // two register sets, requested alternately one point ahead (a rotated set - cur = next - made the copy wait for the request)
    float a[3][CPL], b[3][CPL];
    const int64_t base = cloud * N;
    vt_load_row<CPL>(a, v, base + min(r0 + wave, N - 1), C, lane);
    for (int r = r0 + wave; r < r1; r += 8) {
        vt_load_row<CPL>(b, v, base + min(r + 4, N - 1), C, lane);
        __builtin_amdgcn_sched_barrier(0);
        point(a, r, true);
        vt_load_row<CPL>(a, v, base + min(r + 8, N - 1), C, lane);
        __builtin_amdgcn_sched_barrier(0);
        point(b, r + 4, r + 4 < r1);
    }
    // the four waves' results of a column: larger key (value, then lower row) and the sums in wave order (bit-reproducible)
#pragma unroll
    for (int t = 0; t < CPL; ++t)
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int slot = (t * J + j) * 64 + lane;
            lkey[wave][slot] = any ? vt_pack_key(best[t][j], bi[t][j]) : 0ull;
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
        // (the chunks' partial sums in order - bit-reproducible - with eight loads in flight: one dependent L2 round trip per chunk made
        //  this 16 K-element kernel 8 - 10 us long; 32-bit division: total < 2^20)
        float s = 0.f;
        int64_t c = 0;
        for (; c + 7 < chunks; c += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = part[(c + u) * total + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; c < chunks; ++c) s += part[c * total + e];
        const uint32_t o32 = (uint32_t)e / (uint32_t)inner;
        const int64_t o = o32, i = e - o * inner;
        out_max[o * out_ld + i] = __uint_as_float(u);
        out_mean[o * out_ld + i] = s * invR;
        argmax[e] = (int32_t)(0xFFFFFFFFu - (uint32_t)(kk & 0xFFFFFFFFull));
    }
}

// ---- backward.  APPLY = false, first pass: everything up to VectorBN's batch sums (+ optionally g5 = dL/dv5 for vbn_bwd_apply_kernel).
// APPLY = true, second pass: the SAME per-point recomputation of dL/dv5 (one read of the product instead of reading it and a stored g5),
// then VectorBN's apply pass on it (vbn_bwd_apply_kernel's expressions) with the totals of the first pass's sums -> dv.
template <int CPL, bool APPLY>
__global__ __launch_bounds__(256) void vtail_bwd_kernel(const float* __restrict__ v, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ gate, const float* __restrict__ w,
                                                        const float* __restrict__ gmax, const float* __restrict__ gmean, int64_t g_ld,
                                                        const int32_t* __restrict__ argmax, int N, int C, int rows_per_chunk,
                                                        float* __restrict__ red, float* __restrict__ dgate, float* __restrict__ GX,
                                                        float* __restrict__ g5, int train_stats, float* __restrict__ dv) {
    constexpr int NO = J * CPL;
    __shared__ double lacc[4][2 * CPL * 64];
    __shared__ float lgw[4][(NO + CPL) * 64];            // [wave][gxw (J*CPL) | gsum (CPL)][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t cloud = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_chunk, r1 = min(N, r0 + rows_per_chunk);
    VtStats st{};
    st.mean_in = mean; st.invstd_in = invstd;
    VtChan<CPL> ch;
    vt_load_chan<CPL>(ch, st, gamma, beta, gate, w, cloud, 0, C, lane, wave, false, nullptr);
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
    float r0c[CPL], r1c[CPL];                       // (APPLY) the totals of the first pass's sliced sums: sum dr, sum dr * nhat per channel
    const float invM = 1.f / ((float)gridDim.y * (float)N);
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        gsum[t] = 0.f; acc0[t] = 0.0; acc1[t] = 0.0; r0c[t] = 0.f; r1c[t] = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) gxw[j][t] = 0.f;
        if (APPLY) {
            const int c = lane + 64 * t, cc = c < C ? c : 0;
            r0c[t] = svnet_slices_total(red, 2 * C, cc);
            r1c[t] = svnet_slices_total(red, 2 * C, C + cc);
        }
    }
    if (APPLY) {
        // (every workgroup has read the slices before anyone overwrites the totals' slots [0, 2C): they are separate addresses)
        if (blockIdx.x == 0 && blockIdx.y == 0 && wave == 0) {
#pragma unroll
            for (int t = 0; t < CPL; ++t) {
                const int c = lane + 64 * t;
                if (c < C) { red[c] = r0c[t]; red[C + c] = r1c[t]; }
            }
        }
    }
    // one point (see the forward kernel: no branch in the body; `live` false = a clamped request past the chunk's end, whose pooled
    // gradient is taken as 0 - every sum it feeds then receives 0 - and whose row is not stored)
    auto point = [&](const float (&a)[3][CPL], int r, bool live) {
        float x[3][CPL], n[CPL], rr[CPL], z[3][J], nvv[CPL];
        vt_point<CPL>(a, ch, x, n, rr, z, lane, nvv);
        // pooled gradient of this point's s_v, Vector2Scalar's backward (v2s_bwd_kernel's expressions)
        float d[CPL][J], dz[3][J];
#pragma unroll
        for (int t = 0; t < CPL; ++t)
#pragma unroll
            for (int j = 0; j < J; ++j) d[t][j] = live ? gm[t][j] + (am[t][j] == r ? gx[t][j] : 0.f) : 0.f;
        {
            float pd[3][J];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    pd[i][j] = 0.f;
#pragma unroll
                    for (int t = 0; t < CPL; ++t) pd[i][j] = fmaf(d[t][j], x[i][t], pd[i][j]);
                }
            vt_sum9(pd, lane, dz);
        }
        const int64_t m = cloud * N + r;
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            float g[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float sg = 0.f;
#pragma unroll
                for (int j = 0; j < J; ++j) sg += d[t][j] * z[i][j] + dz[i][j] * ch.w[j][t];
                g[i] = sg;
            }
            const float nh = (n[t] - ch.mu[t]) * ch.is[t];
            const float gv = g[0] * a[0][t] + g[1] * a[1][t] + g[2] * a[2][t];
            const float rn = 1.f / n[t];
            if (APPLY) {
                // VectorBN's apply pass on g = dL/dv5 (vbn_bwd_apply_kernel's expressions: the gate multiplies g, dq = sum_i g_i v_i)
                const float dq = gv * ch.gt[t];
                float dr = dq * rn;
                float dn = -dq * rr[t] * rn * rn;
                if (train_stats) dr -= (r0c[t] + nh * r1c[t]) * invM;
                dn += dr * ch.ga[t] * ch.is[t];
                const float kk = nvv[t] > 0.f ? dn / nvv[t] : 0.f;
                const float q = rr[t] * rn * ch.gt[t];
                if (ch.ok[t] && live) {
                    const int c = lane + 64 * t;
#pragma unroll
                    for (int i = 0; i < 3; ++i) dv[(m * 3 + i) * C + c] = g[i] * q + kk * a[i][t];
                }
            } else {
#pragma unroll
                for (int j = 0; j < J; ++j) gxw[j][t] += dz[0][j] * x[0][t] + dz[1][j] * x[1][t] + dz[2][j] * x[2][t];
                if (g5 && ch.ok[t] && live) {            // (exec-masked stores, no branch; g5 NULL: the apply pass recomputes it)
                    const int c = lane + 64 * t;
#pragma unroll
                    for (int i = 0; i < 3; ++i) g5[(m * 3 + i) * C + c] = g[i];
                }
                // VectorBN's reduce pass on g = dL/dv5 (vbn_bwd_reduce_kernel's expressions); a dead channel's g is 0
                gsum[t] += gv * (rr[t] * rn);
                const float dr = gv * ch.gt[t] * rn;
                acc0[t] += (double)dr;
                acc1[t] += (double)dr * (double)nh;
            }
        }
    };
    float a[3][CPL], b[3][CPL];
    const int64_t base = cloud * N;
    vt_load_row<CPL>(a, v, base + min(r0 + wave, N - 1), C, lane);
    for (int r = r0 + wave; r < r1; r += 8) {
        vt_load_row<CPL>(b, v, base + min(r + 4, N - 1), C, lane);
        __builtin_amdgcn_sched_barrier(0);
        point(a, r, true);
        vt_load_row<CPL>(a, v, base + min(r + 8, N - 1), C, lane);
        __builtin_amdgcn_sched_barrier(0);
        point(b, r + 4, r + 4 < r1);
    }
    if (APPLY) return;
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
    chunks = svnet_cdiv(256 * 3, B);                      // ~3 workgroups per CU: a workgroup's set-up (the channels' statistics and
    if (chunks > svnet_cdiv(N, 32)) chunks = svnet_cdiv(N, 32);   // constants, its final LDS combine + atomics) is paid per workgroup
    if (chunks < 1) chunks = 1;
    rpc = svnet_cdiv(svnet_cdiv(N, chunks), 8) * 8;       // (two points per wave and loop trip)
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
                                   float* out_mean, int64_t out_ld, int32_t* argmax, void* workspace, size_t workspace_bytes,
                                   int workspace_zeroed, void* stream) {
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
    if (!workspace_zeroed) {
        hipError_t e = hipMemsetAsync(keys, 0, (size_t)total * 8, st);
        SVNET_REQUIRE(e == hipSuccess, SVNET_E_LAUNCH, "svnet_vtail_fwd_f32: memset failed");
    }
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
    SVNET_REQUIRE(v && mean && invstd && gamma && beta && w_eff && gmax && gmean && argmax && red && GX, SVNET_E_ARG,
                  "svnet_vtail_bwd_f32: null pointer");
    SVNET_REQUIRE(B > 0 && B <= 65535 && N > 0 && N < (1 << 30) && C > 0 && g_ld >= 3 * C && (!gate || dgate), SVNET_E_ARG,
                  "svnet_vtail_bwd_f32: bad sizes");
    SVNET_REQUIRE(C <= 192, SVNET_E_UNSUPPORTED, "svnet_vtail_bwd_f32: C=%lld > 192 vector channels", (long long)C);
    hipStream_t st = (hipStream_t)stream;
    int64_t chunks, rpc;
    vtail_chunks(B, N, chunks, rpc);
    const dim3 grid((unsigned)chunks, (unsigned)B);
#define SVNET_VT_BWD(CPL_) hipLaunchKernelGGL((vtail_bwd_kernel<CPL_, false>), grid, dim3(256), 0, st, v, mean, invstd, gamma, beta, gate, w_eff, gmax, gmean, g_ld, \
                                              argmax, (int)N, (int)C, (int)rpc, red, dgate, GX, g5, 0, (float*)nullptr)
    if (C <= 64) SVNET_VT_BWD(1);
    else if (C <= 128) SVNET_VT_BWD(2);
    else SVNET_VT_BWD(3);
#undef SVNET_VT_BWD
    SVNET_CHECK_LAUNCH("vtail_bwd_kernel");
    return SVNET_OK;
}

extern "C" int svnet_vtail_bwd_apply_f32(const float* v, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                         const float* gate, const float* w_eff, const float* gmax, const float* gmean, int64_t g_ld,
                                         const int32_t* argmax, int64_t B, int64_t N, int64_t C, float* red, int train_stats, float* dv,
                                         void* stream) {
    SVNET_REQUIRE(v && mean && invstd && gamma && beta && w_eff && gmax && gmean && argmax && red && dv, SVNET_E_ARG,
                  "svnet_vtail_bwd_apply_f32: null pointer");
    SVNET_REQUIRE(B > 0 && B <= 65535 && N > 0 && N < (1 << 30) && C > 0 && g_ld >= 3 * C, SVNET_E_ARG, "svnet_vtail_bwd_apply_f32: bad sizes");
    SVNET_REQUIRE(C <= 192, SVNET_E_UNSUPPORTED, "svnet_vtail_bwd_apply_f32: C=%lld > 192 vector channels", (long long)C);
    hipStream_t st = (hipStream_t)stream;
    int64_t chunks, rpc;
    vtail_chunks(B, N, chunks, rpc);
    const dim3 grid((unsigned)chunks, (unsigned)B);
#define SVNET_VT_APPLY(CPL_) hipLaunchKernelGGL((vtail_bwd_kernel<CPL_, true>), grid, dim3(256), 0, st, v, mean, invstd, gamma, beta, gate, w_eff, gmax, gmean, \
                                                g_ld, argmax, (int)N, (int)C, (int)rpc, red, (float*)nullptr, (float*)nullptr, (float*)nullptr, train_stats, dv)
    if (C <= 64) SVNET_VT_APPLY(1);
    else if (C <= 128) SVNET_VT_APPLY(2);
    else SVNET_VT_APPLY(3);
#undef SVNET_VT_APPLY
    SVNET_CHECK_LAUNCH("vtail_bwd_kernel<apply>");
    return SVNET_OK;
}
