// Neighbour / point pooling, gate activations and the loss.
//
// Replaces  models/utils/sv_util.py:118-132 (svpool: max over scalars, mean over vectors),
//           models/sv_dgcnn_cls.py:72-73 (adaptive max / avg pool over points),
//           the ReLU / Sigmoid of the gate (sv_layers.py:156-161) and utils.py:33-50 (cal_loss).
// x is viewed as [outer, R, inner]; lanes run over `inner` (coalesced), the reduced axis R is walked
// sequentially (R = k = 20..40 for neighbour pooling).  Long reductions with few outputs (gate mean over
// 20 480 edge rows) are split over workgroups and combined with float atomics.
#include <float.h>

#include "common.h"
#include "gate_mlp.h"

namespace {

__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ x, int64_t outer, int64_t R, int64_t inner,
                                                       int mode, float* __restrict__ out, int64_t out_ld, int32_t* __restrict__ argmax) {
    const int64_t total = outer * inner;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = e / inner, i = e - o * inner;
        const float* p = x + o * R * inner + i;
        if (mode == 0) {
            float best = p[0];
            int32_t bi = 0;
            for (int64_t r = 1; r < R; ++r) {
                const float v = p[r * inner];
                if (v > best || (v != v && best == best)) {  // strict: first index wins ties; NaN propagates like torch
                    best = v;
                    bi = (int32_t)r;
                }
            }
            out[o * out_ld + i] = best;
            if (argmax) argmax[e] = bi;
        } else {
            float s = 0.f;
            for (int64_t r = 0; r < R; ++r) s += p[r * inner];
            out[o * out_ld + i] = s / (float)R;
        }
    }
}

// mean with a long reduced axis: grid (chunks, outer); every chunk writes its partial sums to part[chunk][outer*inner] and
// pool_mean_finish_kernel adds the chunks in a fixed order (no float atomics: the result is reproducible bit for bit, which
// matters because these means feed sign() one layer later)
__global__ __launch_bounds__(1024) void pool_mean_split_kernel(const float* __restrict__ x, int64_t R, int64_t inner,
                                                              int64_t rows_per_chunk, float* __restrict__ part, int64_t total) {
    const int64_t o = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    for (int64_t i = threadIdx.x; i < inner; i += blockDim.x) {
        const float* p = x + o * R * inner + i;
        float s = 0.f;
        int64_t r = r0;
        for (; r + 7 < r1; r += 8) {          // eight rows' loads in flight per thread
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = p[(r + u) * inner];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; r < r1; ++r) s += p[r * inner];
        part[(int64_t)blockIdx.x * total + o * inner + i] = s;
    }
}
__global__ __launch_bounds__(256) void pool_mean_finish_kernel(const float* __restrict__ part, int64_t chunks, int64_t total, float invR,
                                                               float* __restrict__ out, int64_t inner, int64_t out_ld) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int64_t c = 0; c < chunks; ++c) s += part[c * total + e];
        out[(e / inner) * out_ld + e % inner] = s * invR;
    }
}

// max with a long reduced axis: grid (chunks, outer); each (o,i) keeps one packed 64-bit key
//   key = (order-preserving bits of the value) << 32 | (0xFFFFFFFF - r)      -> atomicMax = largest value, first index
// in `keys` (pre-filled with 0 by the caller's memset); a second tiny kernel unpacks value and arg-max.
__device__ __forceinline__ unsigned long long pack_key(float v, int64_t r) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)r);
}
__global__ __launch_bounds__(1024) void pool_max_split_kernel(const float* __restrict__ x, int64_t R, int64_t inner,
                                                             int64_t rows_per_chunk, unsigned long long* __restrict__ keys) {
    const int64_t o = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    for (int64_t i = threadIdx.x; i < inner; i += blockDim.x) {
        const float* p = x + o * R * inner + i;
        float best = p[r0 * inner];
        int64_t bi = r0;
        int64_t r = r0 + 1;
        for (; r + 7 < r1; r += 8) {          // eight rows' loads in flight per thread; strict '>' keeps the first index
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = p[(r + u) * inner];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (t[u] > best) { best = t[u]; bi = r + u; }
        }
        for (; r < r1; ++r) {
            const float v = p[r * inner];
            if (v > best) { best = v; bi = r; }
        }
        atomicMax(&keys[o * inner + i], pack_key(best, bi));
    }
}
// max AND mean of a long reduced axis in ONE pass over x (the classifiers' global pooling reads its [B,N,C] features once instead
// of twice): the max part as pool_max_split_kernel (packed keys, atomicMax), the mean part as pool_mean_split_kernel (ordered
// partial sums).
__global__ __launch_bounds__(1024) void pool_maxmean_split_kernel(const float* __restrict__ x, int64_t R, int64_t inner,
                                                                 int64_t rows_per_chunk, unsigned long long* __restrict__ keys,
                                                                 float* __restrict__ part, int64_t total) {
    const int64_t o = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    for (int64_t i = threadIdx.x; i < inner; i += blockDim.x) {
        const float* p = x + o * R * inner + i;
        float best = p[r0 * inner], s = best;
        int64_t bi = r0;
        int64_t r = r0 + 1;
        for (; r + 7 < r1; r += 8) {          // eight rows' loads in flight per thread; strict '>' keeps the first index
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = p[(r + u) * inner];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s += t[u];
                if (t[u] > best) { best = t[u]; bi = r + u; }
            }
        }
        for (; r < r1; ++r) {
            const float v = p[r * inner];
            s += v;
            if (v > best) { best = v; bi = r; }
        }
        atomicMax(&keys[o * inner + i], pack_key(best, bi));
        part[(int64_t)blockIdx.x * total + o * inner + i] = s;
    }
}
// ---- BatchNorm (+ activation) and [max | mean] over the points in ONE pass over the pre-BN tensor y [outer, R, inner] (conv5 of the
// classifier, whose activated output is only ever pooled, sv_dgcnn_cls.py:69-74): the activated tensor is never written, and
// its gradient - (point == arg-max ? g_max : 0) + g_mean / R, the edge block's trick - is never written either: the two
// BatchNorm backward passes below build it from the pooled gradients.
__device__ __forceinline__ float bnp_act(float z, int act, float slope) {
    if (act == 1) return z > 0.f ? z : z * slope;
    if (act == 2) return z > 0.f ? z : 0.f;
    return z;
}
__device__ __forceinline__ float bnp_act_grad(float z, int act, float slope) {
    if (act == 1) return z > 0.f ? 1.f : slope;
    if (act == 2) return z > 0.f ? 1.f : 0.f;
    return 1.f;
}
__global__ __launch_bounds__(1024) void bn_pool_split_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int act, float slope, int64_t R,
                                                            int64_t inner, int64_t rows_per_chunk, unsigned long long* __restrict__ keys,
                                                            float* __restrict__ part, int64_t total) {
    const int64_t o = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    for (int64_t i = threadIdx.x; i < inner; i += blockDim.x) {
        const float mu = mean[i], is = invstd[i], ga = gamma[i], be = beta[i];
        const float* p = x + o * R * inner + i;
        // (the same arithmetic, in the same order, as bn_act_fwd_kernel: the pooled values equal pooling its output)
        float best = bnp_act((p[r0 * inner] - mu) * is * ga + be, act, slope), s = best;
        int64_t bi = r0;
        int64_t r = r0 + 1;
        for (; r + 15 < r1; r += 16) {
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = p[(r + u) * inner];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float z = bnp_act((t[u] - mu) * is * ga + be, act, slope);
                s += z;
                if (z > best) { best = z; bi = r + u; }
            }
        }
        for (; r + 7 < r1; r += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = p[(r + u) * inner];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float z = bnp_act((t[u] - mu) * is * ga + be, act, slope);
                s += z;
                if (z > best) { best = z; bi = r + u; }
            }
        }
        for (; r < r1; ++r) {
            const float z = bnp_act((p[r * inner] - mu) * is * ga + be, act, slope);
            s += z;
            if (z > best) { best = z; bi = r; }
        }
        atomicMax(&keys[o * inner + i], pack_key(best, bi));
        part[(int64_t)blockIdx.x * total + o * inner + i] = s;
    }
}
// red[0:C] += sum g', red[C:2C] += sum g' xhat with g' = g * act'(z), g = (r == argmax ? gmax : 0) + gmean / R
__global__ __launch_bounds__(1024) void bn_pool_bwd_reduce_kernel(const float* __restrict__ gmax, const float* __restrict__ gmean, int64_t g_ld,
                                                                 const int32_t* __restrict__ argmax, const float* __restrict__ x,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                                 float slope, int64_t R, int64_t inner, int64_t rows_per_chunk,
                                                                 float* __restrict__ red) {
    const int64_t o = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    const float invR = 1.f / (float)R;
    for (int64_t i = threadIdx.x; i < inner; i += blockDim.x) {
        const float mu = mean[i], is = invstd[i], ga = gamma[i], be = beta[i];
        const float gx = gmax[o * g_ld + i], gm = gmean[o * g_ld + i] * invR;
        const int am = argmax[o * inner + i];
        const float* p = x + o * R * inner + i;
        double a0 = 0.0, a1 = 0.0;
        int64_t r = r0;
        for (; r + 15 < r1; r += 16) {
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = p[(r + u) * inner];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float xh = (t[u] - mu) * is;
                const float gp = (gm + (am == (int32_t)(r + u) ? gx : 0.f)) * bnp_act_grad(xh * ga + be, act, slope);
                a0 += (double)gp;
                a1 += (double)gp * (double)xh;
            }
        }
        for (; r + 7 < r1; r += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = p[(r + u) * inner];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float xh = (t[u] - mu) * is;
                const float gp = (gm + (am == (int32_t)(r + u) ? gx : 0.f)) * bnp_act_grad(xh * ga + be, act, slope);
                a0 += (double)gp;
                a1 += (double)gp * (double)xh;
            }
        }
        for (; r < r1; ++r) {
            const float xh = (p[r * inner] - mu) * is;
            const float gp = (gm + (am == (int32_t)r ? gx : 0.f)) * bnp_act_grad(xh * ga + be, act, slope);
            a0 += (double)gp;
            a1 += (double)gp * (double)xh;
        }
        float* sl = svnet_slice_ptr(red, 2 * (int)inner);   // (chunks x outer adders per column: a few hundred - spread over slices)
        svnet_slice_add(&sl[i], (float)a0);
        svnet_slice_add(&sl[inner + i], (float)a1);
    }
}
__global__ __launch_bounds__(1024) void bn_pool_bwd_apply_kernel(const float* __restrict__ gmax, const float* __restrict__ gmean, int64_t g_ld,
                                                                const int32_t* __restrict__ argmax, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ red, int act, float slope, int train_stats,
                                                                int64_t outer, int64_t R, int64_t inner, int64_t rows_per_chunk,
                                                                float* __restrict__ dx) {
    const int64_t o = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    const float invR = 1.f / (float)R, invM = 1.f / (float)(outer * R);
    for (int64_t i = threadIdx.x; i < inner; i += blockDim.x) {
        // (`red`: the slices the reduce kernel filled; workgroup (0, 0) leaves the totals - dL/dbeta, dL/dgamma - in its first 2 * inner elements)
        const float q0 = svnet_slices_total(red, 2 * (int)inner, (int)i), q1 = svnet_slices_total(red, 2 * (int)inner, (int)(inner + i));
        if (blockIdx.x == 0 && blockIdx.y == 0) { red[i] = q0; red[inner + i] = q1; }
        const float mu = mean[i], is = invstd[i], ga = gamma[i], be = beta[i];
        const float gx = gmax[o * g_ld + i], gm = gmean[o * g_ld + i] * invR;
        const int am = argmax[o * inner + i];
        const float* p = x + o * R * inner + i;
        float* d = dx + o * R * inner + i;
        // (sixteen rows requested before the first is used: with four, a CU had 32 KB in flight and the pass ran at 2.3 TB/s)
        int64_t r = r0;
        for (; r + 15 < r1; r += 16) {
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = p[(r + u) * inner];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float xh = (t[u] - mu) * is;
                float gp = (gm + (am == (int32_t)(r + u) ? gx : 0.f)) * bnp_act_grad(xh * ga + be, act, slope);
                if (train_stats) gp -= (q0 + xh * q1) * invM;
                d[(r + u) * inner] = gp * ga * is;
            }
        }
        for (; r < r1; ++r) {
            const float xh = (p[r * inner] - mu) * is;
            float gp = (gm + (am == (int32_t)r ? gx : 0.f)) * bnp_act_grad(xh * ga + be, act, slope);
            if (train_stats) gp -= (q0 + xh * q1) * invM;
            d[r * inner] = gp * ga * is;
        }
    }
}
__global__ __launch_bounds__(256) void pool_max_unpack_kernel(const unsigned long long* __restrict__ keys, int64_t total,
                                                              float* __restrict__ out, int32_t* __restrict__ argmax, int64_t inner, int64_t out_ld) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long kk = keys[e];
        uint32_t u = (uint32_t)(kk >> 32);
        u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
        out[(e / inner) * out_ld + e % inner] = __uint_as_float(u);
        if (argmax) argmax[e] = (int32_t)(0xFFFFFFFFu - (uint32_t)(kk & 0xFFFFFFFFull));
    }
}

// unpack of the max keys and ordered finish of the mean in one launch (the [max | mean] pair)
__global__ __launch_bounds__(256) void pool_maxmean_finish_kernel(const unsigned long long* __restrict__ keys, const float* __restrict__ part,
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

__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ g, const int32_t* __restrict__ argmax,
                                                       int64_t outer, int64_t R, int64_t inner, int mode,
                                                       float* __restrict__ dx) {
    const int64_t total = outer * R * inner;
    const float invR = 1.f / (float)R;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e % inner;
        const int64_t orr = e / inner;
        const int64_t r = orr % R, o = orr / R;
        const float gv = g[o * inner + i];
        dx[e] = (mode == 0) ? ((argmax[o * inner + i] == (int32_t)r) ? gv : 0.f) : gv * invR;
    }
}

// dx[o,r,i] = add[(o*R + r)*add_ld + i] + gmean[o,i] / R: the backward of a mean over r ADDED to another gradient of the same tensor that
// arrives as rows of stride add_ld (a column slice of a wider gradient), written once - instead of a broadcast pass (pool_bwd_kernel),
// and a strided elementwise add of the two.  grid (row chunks, outer).
__global__ __launch_bounds__(1024) void pool_mean_bwd_add_kernel(const float* __restrict__ gmean, const float* __restrict__ add, int64_t add_ld,
                                                                int64_t R, int64_t inner, int64_t rows_per_chunk, float* __restrict__ dx) {
    const int64_t o = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    const float invR = 1.f / (float)R;
    for (int64_t i = threadIdx.x; i < inner; i += blockDim.x) {
        const float gm = gmean[o * inner + i] * invR;
        const float* a = add + (o * R) * add_ld + i;
        float* d = dx + (o * R) * inner + i;
        int64_t r = r0;
        for (; r + 7 < r1; r += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = a[(r + u) * add_ld];
#pragma unroll
            for (int u = 0; u < 8; ++u) d[(r + u) * inner] = t[u] + gm;
        }
        for (; r < r1; ++r) d[r * inner] = a[r * add_ld] + gm;
    }
}

// Backward of [max | mean] over the same reduced axis in one pass: dx[o,r,i] = (argmax[o,i] == r ? gmax[o,i] : 0) + gmean[o,i] / R
// (gmax / gmean: rows of stride g_ld).  grid (row chunks, outer); a thread keeps its columns' three operands in registers and
// streams the rows (no divisions).
__global__ __launch_bounds__(256) void pool_maxmean_bwd_kernel(const float* __restrict__ gmax, const float* __restrict__ gmean, int64_t g_ld,
                                                               const int32_t* __restrict__ argmax, int64_t R,
                                                               int64_t inner, int64_t rows_per_chunk, float* __restrict__ dx) {
    const int64_t o = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    const float invR = 1.f / (float)R;
    for (int64_t i0 = 0; i0 < inner; i0 += 4 * 256) {
        float gx[4], gm[4];
        int am[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * 256 + threadIdx.x;
            const bool ok = i < inner;
            gx[u] = (ok && gmax) ? gmax[o * g_ld + i] : 0.f;            // (either part may be absent: svnet_pool_bwd_f32)
            gm[u] = (ok && gmean) ? gmean[o * g_ld + i] * invR : 0.f;
            am[u] = (ok && gmax) ? argmax[o * inner + i] : -1;
        }
        for (int64_t r = r0; r < r1; ++r) {
            float* row = dx + (o * R + r) * inner;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + u * 256 + threadIdx.x;
                if (i < inner) row[i] = gm[u] + (am[u] == (int32_t)r ? gx[u] : 0.f);
            }
        }
    }
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, int64_t n, int kind, float* __restrict__ y) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[e];
        float r;
        if (kind == 1) r = v > 0.f ? v : 0.f;
        else if (kind == 2) r = 1.f / (1.f + expf(-v));
        else r = v > 0.f ? v : 0.2f * v;
        y[e] = r;
    }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, int64_t n, int kind,
                                                      float* __restrict__ dx) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float v = y[e];
        float d;
        if (kind == 1) d = v > 0.f ? 1.f : 0.f;
        else if (kind == 2) d = v * (1.f - v);
        else d = v > 0.f ? 1.f : 0.2f;
        dx[e] = g[e] * d;
    }
}

// one wave per row: log-softmax, smoothed target, loss and gradient.  Every workgroup writes ONE partial loss (its four
// waves added in a fixed order); smooth_ce_finish_kernel adds the partials in a fixed order: no float atomics, so the loss
// is reproducible bit for bit.
__global__ __launch_bounds__(256) void smooth_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                        int64_t R, int64_t C, float eps, float* __restrict__ partial,
                                                        float* __restrict__ dlogits) {
    __shared__ float wsum[4];
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float off = eps / (float)(C - 1), on = 1.f - eps;
    const float invR = 1.f / (float)R;
    float local = 0.f;
    for (int64_t r = wave; r < R; r += nwaves) {
        const float* row = logits + r * C;
        float mx = -FLT_MAX;
        for (int64_t c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float se = 0.f;
        for (int64_t c = lane; c < C; c += 64) se += expf(row[c] - mx);
        se = wave_sum(se);
        const float lse = logf(se) + mx;
        const int64_t t = target[r];
        float part = 0.f;
        for (int64_t c = lane; c < C; c += 64) {
            const float logp = row[c] - lse;
            const float soft = (c == t) ? on : off;
            part -= soft * logp;
            if (dlogits) dlogits[r * C + c] = (expf(logp) - soft) * invR;
        }
        local += wave_sum(part);
    }
    if (lane == 0) wsum[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) * invR;
}
__global__ void smooth_ce_finish_kernel(const float* __restrict__ partial, int n, float* __restrict__ loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) *loss = s;
}

}  // namespace

// Threads per workgroup of the (row chunk, outer) kernels whose threads own COLUMNS (`for i = threadIdx.x; i < inner; i += blockDim.x`): one
// column per thread up to 1024 - with 256 threads a 512-column tensor (conv5's [B, N, 512]) was walked as two column passes one after
// the other, i.e. with half the loads in flight (bn_pool_fwd 42 us for 67 MB = 1.6 TB/s)
static unsigned pool_col_block(int64_t inner) {
    const int64_t b = (inner + 63) / 64 * 64;
    return (unsigned)(b > 1024 ? 1024 : (b < 64 ? 64 : b));
}
static int64_t pool_split_chunks(int64_t outer, int64_t R) {
    int64_t chunks = svnet_cdiv(256 * 8, outer);
    if (chunks > svnet_cdiv(R, 32)) chunks = svnet_cdiv(R, 32);
    return chunks < 1 ? 1 : chunks;
}

/* bytes of workspace that let a long reduction (R >= 256, few outputs) be split over workgroups; 0 = no split path */
extern "C" size_t svnet_pool_workspace_bytes(int64_t outer, int64_t R, int64_t inner, int mode) {
    const int64_t total = outer * inner;
    if (R < 256 || total >= (1 << 20) || outer > 65535 || outer <= 0) return 0;
    if (mode == 0) return (size_t)total * 8;
    int64_t chunks = pool_split_chunks(outer, R);
    chunks = svnet_cdiv(R, svnet_cdiv(R, chunks));
    return (size_t)(chunks * total) * sizeof(float);
}

extern "C" int svnet_pool_fwd_f32(const float* x, int64_t outer, int64_t R, int64_t inner, int mode, float* out, int64_t out_ld,
                                  int32_t* argmax, void* workspace, size_t workspace_bytes, void* stream) {
    SVNET_REQUIRE(x && out && outer >= 0 && R > 0 && inner > 0 && (mode == 0 || mode == 1) && out_ld >= inner, SVNET_E_ARG,
                  "svnet_pool_fwd_f32: bad arguments");
    if (outer == 0) return SVNET_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = outer * inner;
    if (mode == 1 && R >= 256 && total < (1 << 20) && outer <= 65535) {
        int64_t chunks = pool_split_chunks(outer, R);
        const int64_t rpc = svnet_cdiv(R, chunks);
        chunks = svnet_cdiv(R, rpc);
        if (workspace && workspace_bytes >= (size_t)(chunks * total) * sizeof(float)) {
            float* part = (float*)workspace;
            hipLaunchKernelGGL(pool_mean_split_kernel, dim3((unsigned)chunks, (unsigned)outer), dim3(pool_col_block(inner)), 0, st, x, R, inner, rpc, part,
                               total);
            SVNET_CHECK_LAUNCH("pool_mean_split_kernel");
            hipLaunchKernelGGL(pool_mean_finish_kernel, dim3(svnet_grid(total, 256)), dim3(256), 0, st, part, chunks, total,
                               1.f / (float)R, out, inner, out_ld);
            SVNET_CHECK_LAUNCH("pool_mean_finish_kernel");
            return SVNET_OK;
        }
    }
    if (mode == 0 && R >= 256 && total < (1 << 20) && workspace && workspace_bytes >= (size_t)total * 8 && outer <= 65535) {
        // long max-reduction with few outputs (point pooling over N): split the rows over workgroups
        unsigned long long* keys = (unsigned long long*)workspace;
        hipError_t e = hipMemsetAsync(keys, 0, sizeof(unsigned long long) * total, st);
        SVNET_REQUIRE(e == hipSuccess, SVNET_E_LAUNCH, "svnet_pool_fwd_f32: memset failed");
        int64_t chunks = pool_split_chunks(outer, R);
        const int64_t rpc = svnet_cdiv(R, chunks);
        chunks = svnet_cdiv(R, rpc);
        hipLaunchKernelGGL(pool_max_split_kernel, dim3((unsigned)chunks, (unsigned)outer), dim3(pool_col_block(inner)), 0, st, x, R, inner, rpc, keys);
        SVNET_CHECK_LAUNCH("pool_max_split_kernel");
        hipLaunchKernelGGL(pool_max_unpack_kernel, dim3(svnet_grid(total, 256)), dim3(256), 0, st, keys, total, out, argmax, inner, out_ld);
        SVNET_CHECK_LAUNCH("pool_max_unpack_kernel");
        return SVNET_OK;
    }
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(svnet_grid(total, 256, 256 * 32)), dim3(256), 0, st, x, outer, R, inner, mode, out, out_ld, argmax);
    SVNET_CHECK_LAUNCH("pool_fwd_kernel");
    return SVNET_OK;
}

/* [max | mean] over R in one pass (R >= 256, split over workgroups; the partial sums of the mean are added in a fixed order).
 * workspace: svnet_pool_workspace_bytes(.., 0) + svnet_pool_workspace_bytes(.., 1) bytes, the key part first.               */
extern "C" int svnet_pool_maxmean_fwd_f32(const float* x, int64_t outer, int64_t R, int64_t inner, float* out_max, float* out_mean,
                                          int64_t out_ld, int32_t* argmax, void* workspace, size_t workspace_bytes, void* stream) {
    SVNET_REQUIRE(x && out_max && out_mean && argmax && outer > 0 && R >= 256 && inner > 0 && out_ld >= inner, SVNET_E_ARG,
                  "svnet_pool_maxmean_fwd_f32: bad arguments (R >= 256)");
    const int64_t total = outer * inner;
    SVNET_REQUIRE(total < (1 << 20) && outer <= 65535, SVNET_E_UNSUPPORTED, "svnet_pool_maxmean_fwd_f32: too many outputs");
    int64_t chunks = pool_split_chunks(outer, R);
    const int64_t rpc = svnet_cdiv(R, chunks);
    chunks = svnet_cdiv(R, rpc);
    const size_t key_bytes = (size_t)total * 8;
    SVNET_REQUIRE(workspace && workspace_bytes >= key_bytes + (size_t)(chunks * total) * sizeof(float), SVNET_E_ARG,
                  "svnet_pool_maxmean_fwd_f32: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* keys = (unsigned long long*)workspace;
    float* part = (float*)((char*)workspace + key_bytes);
    hipError_t e = hipMemsetAsync(keys, 0, key_bytes, st);
    SVNET_REQUIRE(e == hipSuccess, SVNET_E_LAUNCH, "svnet_pool_maxmean_fwd_f32: memset failed");
    hipLaunchKernelGGL(pool_maxmean_split_kernel, dim3((unsigned)chunks, (unsigned)outer), dim3(pool_col_block(inner)), 0, st, x, R, inner, rpc, keys, part, total);
    SVNET_CHECK_LAUNCH("pool_maxmean_split_kernel");
    hipLaunchKernelGGL(pool_maxmean_finish_kernel, dim3(svnet_grid(total, 256)), dim3(256), 0, st, keys, part, chunks, total, 1.f / (float)R,
                       out_max, out_mean, argmax, inner, out_ld);
    SVNET_CHECK_LAUNCH("pool_maxmean_finish_kernel");
    return SVNET_OK;
}

/* BatchNorm (+ activation) of y [outer*R, inner] with the given statistics, pooled [max | mean] over R in the same pass (the activated
 * tensor is not written).  Outputs and workspace as svnet_pool_maxmean_fwd_f32.                                                   */
extern "C" int svnet_bn_pool_fwd_f32(const float* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                     int64_t outer, int64_t R, int64_t inner, int act, float slope, float* out_max, float* out_mean,
                                     int64_t out_ld, int32_t* argmax, void* workspace, size_t workspace_bytes, int workspace_zeroed,
                                     void* stream) {
    SVNET_REQUIRE(y && mean && invstd && gamma && beta && out_max && out_mean && argmax && outer > 0 && R >= 256 && inner > 0 &&
                      out_ld >= inner, SVNET_E_ARG, "svnet_bn_pool_fwd_f32: bad arguments (R >= 256)");
    const int64_t total = outer * inner;
    SVNET_REQUIRE(total < (1 << 20) && outer <= 65535, SVNET_E_UNSUPPORTED, "svnet_bn_pool_fwd_f32: too many outputs");
    int64_t chunks = pool_split_chunks(outer, R);
    const int64_t rpc = svnet_cdiv(R, chunks);
    chunks = svnet_cdiv(R, rpc);
    const size_t key_bytes = (size_t)total * 8;
    SVNET_REQUIRE(workspace && workspace_bytes >= key_bytes + (size_t)(chunks * total) * sizeof(float), SVNET_E_ARG,
                  "svnet_bn_pool_fwd_f32: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* keys = (unsigned long long*)workspace;
    float* part = (float*)((char*)workspace + key_bytes);
    if (!workspace_zeroed) {      // (a caller that hands over zero-filled memory - one fill per step for everything - saves this launch)
        hipError_t e = hipMemsetAsync(keys, 0, key_bytes, st);
        SVNET_REQUIRE(e == hipSuccess, SVNET_E_LAUNCH, "svnet_bn_pool_fwd_f32: memset failed");
    }
    hipLaunchKernelGGL(bn_pool_split_kernel, dim3((unsigned)chunks, (unsigned)outer), dim3(pool_col_block(inner)), 0, st, y, mean, invstd, gamma, beta, act,
                       slope, R, inner, rpc, keys, part, total);
    SVNET_CHECK_LAUNCH("bn_pool_split_kernel");
    hipLaunchKernelGGL(pool_maxmean_finish_kernel, dim3(svnet_grid(total, 256)), dim3(256), 0, st, keys, part, chunks, total, 1.f / (float)R,
                       out_max, out_mean, argmax, inner, out_ld);
    SVNET_CHECK_LAUNCH("pool_maxmean_finish_kernel");
    return SVNET_OK;
}

/* Backward of svnet_bn_pool_fwd_f32 from the POOLED gradients (gmax / gmean rows of stride g_ld): red [2*inner] (caller zero-fills)
 * receives sum g' and sum g'*xhat (= dbeta, dgamma), then dy = gamma*invstd*(g' - (red0 + xhat*red1)/M) when train_stats.          */
extern "C" int svnet_bn_pool_bwd_f32(const float* gmax, const float* gmean, int64_t g_ld, const int32_t* argmax, const float* y,
                                     const float* mean, const float* invstd, const float* gamma, const float* beta, int64_t outer,
                                     int64_t R, int64_t inner, int act, float slope, int train_stats, float* red, float* dy, void* stream) {
    SVNET_REQUIRE(gmax && gmean && argmax && y && mean && invstd && gamma && beta && red && outer > 0 && R > 0 && inner > 0 &&
                      g_ld >= inner && outer <= 65535, SVNET_E_ARG, "svnet_bn_pool_bwd_f32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    // reduce: ~512 workgroups in all (every one ends in an atomic per column onto the same 2*inner addresses: thousands of adders
    // per cache line serialise at the memory side); apply: ~2048 (it streams two tensors and has nothing to combine)
    auto split = [&](int64_t target, int64_t& chunks, int64_t& rpc) {
        chunks = svnet_cdiv(target, outer);
        if (chunks > svnet_cdiv(R, 8)) chunks = svnet_cdiv(R, 8);
        if (chunks < 1) chunks = 1;
        rpc = svnet_cdiv(R, chunks);
        chunks = svnet_cdiv(R, rpc);
    };
    int64_t chunks, rpc;
    split(512, chunks, rpc);
    hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3((unsigned)chunks, (unsigned)outer), dim3(pool_col_block(inner)), 0, st, gmax, gmean, g_ld, argmax, y, mean,
                       invstd, gamma, beta, act, slope, R, inner, rpc, red);
    SVNET_CHECK_LAUNCH("bn_pool_bwd_reduce_kernel");
    if (dy) {
        split(2048, chunks, rpc);
        hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3((unsigned)chunks, (unsigned)outer), dim3(pool_col_block(inner)), 0, st, gmax, gmean, g_ld, argmax, y, mean,
                           invstd, gamma, beta, red, act, slope, train_stats, outer, R, inner, rpc, dy);
        SVNET_CHECK_LAUNCH("bn_pool_bwd_apply_kernel");
    } else {
        return svnet_slices_sum_f32(red, 2 * inner, stream);          // (no apply pass: the totals by a kernel of their own)
    }
    return SVNET_OK;
}

extern "C" int svnet_pool_bwd_f32(const float* g, const int32_t* argmax, int64_t outer, int64_t R, int64_t inner, int mode,
                                  float* dx, void* stream) {
    SVNET_REQUIRE(g && dx && outer >= 0 && R > 0 && inner > 0 && (mode == 1 || (mode == 0 && argmax)), SVNET_E_ARG, "svnet_pool_bwd_f32: bad arguments");
    if (outer == 0) return SVNET_OK;
    if (outer <= 65535 && inner >= 128) {
        // rows streamed by a thread per column (pool_maxmean_bwd_kernel with one of its two parts): the flat kernel below spends three
        // 64-bit divisions on every element
        int64_t chunks = svnet_cdiv(256 * 16, outer);
        if (chunks > R) chunks = R;
        const int64_t rpc = svnet_cdiv(R, chunks);
        chunks = svnet_cdiv(R, rpc);
        hipLaunchKernelGGL(pool_maxmean_bwd_kernel, dim3((unsigned)chunks, (unsigned)outer), dim3(256), 0, (hipStream_t)stream,
                           mode == 0 ? g : nullptr, mode == 1 ? g : nullptr, inner, argmax, R, inner, rpc, dx);
        SVNET_CHECK_LAUNCH("pool_maxmean_bwd_kernel");
        return SVNET_OK;
    }
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(svnet_grid(outer * R * inner, 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream, g, argmax,
                       outer, R, inner, mode, dx);
    SVNET_CHECK_LAUNCH("pool_bwd_kernel");
    return SVNET_OK;
}

extern "C" int svnet_pool_mean_bwd_add_f32(const float* gmean, const float* add, int64_t add_ld, int64_t outer, int64_t R, int64_t inner,
                                           float* dx, void* stream) {
    SVNET_REQUIRE(gmean && add && dx && outer >= 0 && R > 0 && inner > 0 && add_ld >= inner && outer <= 65535, SVNET_E_ARG,
                  "svnet_pool_mean_bwd_add_f32: bad arguments");
    if (outer == 0) return SVNET_OK;
    int64_t chunks = svnet_cdiv(256 * 8, outer);
    if (chunks > svnet_cdiv(R, 8)) chunks = svnet_cdiv(R, 8);
    if (chunks < 1) chunks = 1;
    const int64_t rpc = svnet_cdiv(R, chunks);
    chunks = svnet_cdiv(R, rpc);
    hipLaunchKernelGGL(pool_mean_bwd_add_kernel, dim3((unsigned)chunks, (unsigned)outer), dim3(pool_col_block(inner)), 0, (hipStream_t)stream, gmean, add, add_ld,
                       R, inner, rpc, dx);
    SVNET_CHECK_LAUNCH("pool_mean_bwd_add_kernel");
    return SVNET_OK;
}

extern "C" int svnet_pool_maxmean_bwd_f32(const float* gmax, const float* gmean, int64_t g_ld, const int32_t* argmax, int64_t outer,
                                          int64_t R, int64_t inner, float* dx, void* stream) {
    SVNET_REQUIRE(gmax && gmean && argmax && dx && outer >= 0 && R > 0 && inner > 0 && g_ld >= inner, SVNET_E_ARG,
                  "svnet_pool_maxmean_bwd_f32: bad arguments");
    SVNET_REQUIRE(outer <= 65535, SVNET_E_UNSUPPORTED, "svnet_pool_maxmean_bwd_f32: outer > 65535");
    if (outer == 0) return SVNET_OK;
    int64_t chunks = svnet_cdiv(256 * 16, outer);
    if (chunks > R) chunks = R;
    const int64_t rpc = svnet_cdiv(R, chunks);
    chunks = svnet_cdiv(R, rpc);
    hipLaunchKernelGGL(pool_maxmean_bwd_kernel, dim3((unsigned)chunks, (unsigned)outer), dim3(256), 0, (hipStream_t)stream, gmax, gmean,
                       g_ld, argmax, R, inner, rpc, dx);
    SVNET_CHECK_LAUNCH("pool_maxmean_bwd_kernel");
    return SVNET_OK;
}

extern "C" int svnet_act_fwd_f32(const float* x, int64_t n, int kind, float* y, void* stream) {
    SVNET_REQUIRE(x && y && n >= 0 && kind >= 1 && kind <= 3, SVNET_E_ARG, "svnet_act_fwd_f32: bad arguments");
    if (n == 0) return SVNET_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(svnet_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, kind, y);
    SVNET_CHECK_LAUNCH("act_fwd_kernel");
    return SVNET_OK;
}

extern "C" int svnet_act_bwd_f32(const float* g, const float* y, int64_t n, int kind, float* dx, void* stream) {
    SVNET_REQUIRE(g && y && dx && n >= 0 && kind >= 1 && kind <= 3, SVNET_E_ARG, "svnet_act_bwd_f32: bad arguments");
    if (n == 0) return SVNET_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(svnet_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, g, y, n, kind, dx);
    SVNET_CHECK_LAUNCH("act_bwd_kernel");
    return SVNET_OK;
}

extern "C" int svnet_smooth_ce_f32(const float* logits, const int64_t* target, int64_t R, int64_t C, float eps, float* loss,
                                   float* dlogits, float* workspace, int64_t workspace_floats, void* stream) {
    SVNET_REQUIRE(logits && target && loss && R > 0 && C > 1, SVNET_E_ARG, "svnet_smooth_ce_f32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = svnet_grid(R * 64, 256, 1024);
    SVNET_REQUIRE(workspace && workspace_floats >= blocks, SVNET_E_ARG, "svnet_smooth_ce_f32: workspace of 1024 floats required");
    hipLaunchKernelGGL(smooth_ce_kernel, dim3(blocks), dim3(256), 0, st, logits, target, R, C, eps, workspace, dlogits);
    SVNET_CHECK_LAUNCH("smooth_ce_kernel");
    hipLaunchKernelGGL(smooth_ce_finish_kernel, dim3(1), dim3(64), 0, st, workspace, blocks, loss);
    SVNET_CHECK_LAUNCH("smooth_ce_finish_kernel");
    return SVNET_OK;
}

// ------------------------------------------------------------------------------------------------ gate MLP
// The per-cloud gate of an SVBlock (sv_layers.py:156-161,179-183): gate = sigmoid(W2 . relu(W0 . (in_scale * gin))).
// [B, Cin] -> [B, H] -> [B, Ov] with B = 32, Cin <= 2048: one workgroup per cloud does the whole thing (forward) and the
// whole chain rule (backward) instead of 4 + 8 launch-bound micro-kernels.
namespace {

__global__ __launch_bounds__(256) void gate_mlp_fwd_kernel(svnet_gate_fwd_job j) { svnet_gate_fwd_block(j, blockIdx.x); }
__global__ __launch_bounds__(256) void gate_mlp_bwd_kernel(svnet_gate_bwd_job j) { svnet_gate_bwd_block(j, blockIdx.x, blockIdx.y, gridDim.y); }

}  // namespace

extern "C" int svnet_gate_mlp_fwd_f32(const float* gin, const double* gin_f64, float* gin_out, float in_scale, const float* W0,
                                      const float* W2, int64_t B, int64_t Cin, int64_t H, int64_t Ov, float* h, float* gate,
                                      const float* rows, int64_t R, void* stream) {
    SVNET_REQUIRE((gin || (gin_f64 && gin_out) || (rows && gin_out && R > 0 && Cin <= 256)) && W0 && W2 && h && gate && B >= 0 && Cin > 0 && H > 0 && Ov > 0, SVNET_E_ARG,
                  "svnet_gate_mlp_fwd_f32: bad arguments");
    SVNET_REQUIRE(H <= 256 && Ov <= 256, SVNET_E_UNSUPPORTED, "svnet_gate_mlp_fwd_f32: H, Ov must be <= 256");
    if (B == 0) return SVNET_OK;
    const svnet_gate_fwd_job j = {gin, gin_f64, gin_out, in_scale, W0, W2, B, Cin, H, Ov, h, gate, rows, R};
    hipLaunchKernelGGL(gate_mlp_fwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, j);
    SVNET_CHECK_LAUNCH("gate_mlp_fwd_kernel");
    return SVNET_OK;
}

extern "C" int svnet_gate_mlp_bwd_f32(const float* dgate, const float* gate, const float* h, const float* gin, float in_scale,
                                      const float* W0, const float* W2, int64_t B, int64_t Cin, int64_t H, int64_t Ov, float out_scale,
                                      float* dgin, float* dW0, float* dW2, void* stream) {
    SVNET_REQUIRE(dgate && gate && h && gin && W0 && W2 && dW0 && dW2 && B >= 0 && Cin > 0 && H > 0 && Ov > 0, SVNET_E_ARG,
                  "svnet_gate_mlp_bwd_f32: bad arguments");
    SVNET_REQUIRE(H <= 256 && Ov <= 256, SVNET_E_UNSUPPORTED, "svnet_gate_mlp_bwd_f32: H, Ov must be <= 256");
    if (B == 0) return SVNET_OK;
    const int chunks = svnet_gate_bwd_chunks(Cin, H, Ov);
    const svnet_gate_bwd_job j = {dgate, gate, h, gin, in_scale, W0, W2, B, Cin, H, Ov, out_scale, dgin, dW0, dW2};
    hipLaunchKernelGGL(gate_mlp_bwd_kernel, dim3((unsigned)B, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, j);
    SVNET_CHECK_LAUNCH("gate_mlp_bwd_kernel");
    return SVNET_OK;
}
