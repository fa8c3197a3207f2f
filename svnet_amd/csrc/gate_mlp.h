// The per-cloud gate MLP of an SVBlock as device functions: one workgroup per cloud (forward) / per (cloud, chunk) (backward).  Used by
// the stand-alone kernels (pool.hip) and by the coefficient launches of the fused layers (edgeblock.hip, xyzblock.hip, edgeblock_bwd.hip),
// whose extra workgroups run a gate job BESIDE the coefficients: the two only share their inputs, and as two dependent launches of
// latency-bound single-wave work they cost 8 + 12 us (forward) and 30 + 7 us (backward) of every fused layer's critical path.
#pragma once
#include "common.h"

// gate = sigmoid(W2 . relu(W0 . (in_scale * gin)))   (sv_layers.py:156-161,179-183)
__device__ __forceinline__ void svnet_gate_fwd_block(const svnet_gate_fwd_job& j, int b) {
    __shared__ float hs[256];
    const int tid = threadIdx.x, Cin = (int)j.Cin, H = (int)j.H, Ov = (int)j.Ov;
    const float* gin = j.gin;
    if (j.gin_f64) {   // fp64 sums of a fused edge layer: rounded to fp32 once, kept (gin_out) for the backward
        for (int c = tid; c < Cin; c += blockDim.x) j.gin_out[(size_t)b * Cin + c] = (float)j.gin_f64[(size_t)b * Cin + c];
        __syncthreads();
        gin = j.gin_out;
    } else if (j.rows) {
        // the mean over the cloud's R rows, formed here (a pooling pass of its own was two launches - split and finish - in front of
        // this one): CW = the power of two >= Cin lanes walk a row, 256 / CW rows at a time, eight loads in flight; fixed order
        __shared__ float part[256];
        int cw = 1;
        while (cw < Cin) cw <<= 1;                                      // (host: Cin <= 256)
        const int RL = 256 / cw, c = tid & (cw - 1), rl = tid / cw;
        const float* base = j.rows + (size_t)b * j.R * Cin;
        float acc = 0.f;
        if (c < Cin) {
            int64_t r = rl;
            for (; r + 7 * RL < j.R; r += 8 * RL) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = base[(r + u * RL) * Cin + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; r < j.R; r += RL) acc += base[r * Cin + c];
        }
        part[tid] = acc;
        __syncthreads();
        if (rl == 0 && c < Cin) {
            float s = 0.f;
            for (int q = 0; q < RL; ++q) s += part[q * cw + c];
            j.gin_out[(size_t)b * Cin + c] = s / (float)j.R;
        }
        __syncthreads();
        gin = j.gin_out;
    }
    const float* g = gin + (size_t)b * Cin;
    for (int r = tid; r < H; r += blockDim.x) {
        float a = 0.f;
#pragma unroll 8
        for (int c = 0; c < Cin; ++c) a = fmaf(g[c] * j.in_scale, j.W0[r * Cin + c], a);
        a = a > 0.f ? a : 0.f;
        hs[r] = a;
        j.h[(size_t)b * H + r] = a;
    }
    __syncthreads();
    for (int o = tid; o < Ov; o += blockDim.x) {
        float a = 0.f;
        for (int r = 0; r < H; ++r) a = fmaf(hs[r], j.W2[o * H + r], a);
        j.gate[(size_t)b * Ov + o] = 1.f / (1.f + expf(-a));
    }
}

// dgin[b,c] = out_scale * sum_j dhpre[b,j] W0[j,c];  dW0 += dhpre^T (in_scale*gin);  dW2 += dgpre^T h   (atomics, zero-filled).
// (cloud b, chunk of chunks): every workgroup recomputes the two short per-cloud vectors (cheap) and takes every chunks-th slice of
// the three output loops, so that a wide layer (conv5: 85 x 256 + 170 x 85 outputs) is not 32 long serial loops
__device__ __forceinline__ void svnet_gate_bwd_block(const svnet_gate_bwd_job& j, int b, int chunk, int chunks) {
    __shared__ float dgp[256], dhp[256];
    const int tid = threadIdx.x, Cin = (int)j.Cin, H = (int)j.H, Ov = (int)j.Ov;
    const int t0 = chunk * blockDim.x + tid, ts = chunks * blockDim.x;
    for (int o = tid; o < Ov; o += blockDim.x) {
        const float gt = j.gate[(size_t)b * Ov + o];
        dgp[o] = j.dgate[(size_t)b * Ov + o] * gt * (1.f - gt);
    }
    __syncthreads();
    for (int r = tid; r < H; r += blockDim.x) {
        float a = 0.f;
#pragma unroll 8
        for (int o = 0; o < Ov; ++o) a = fmaf(dgp[o], j.W2[o * H + r], a);
        dhp[r] = j.h[(size_t)b * H + r] > 0.f ? a : 0.f;
    }
    __syncthreads();
    for (int e = t0; e < Ov * H; e += ts) {
        const int o = e / H, r = e - o * H;
        atomicAdd(&j.dW2[e], dgp[o] * j.h[(size_t)b * H + r]);
    }
    for (int e = t0; e < H * Cin; e += ts) {
        const int r = e / Cin, c = e - r * Cin;
        atomicAdd(&j.dW0[e], dhp[r] * j.gin[(size_t)b * Cin + c] * j.in_scale);
    }
    if (j.dgin) {
        for (int c = t0; c < Cin; c += ts) {
            float a = 0.f;
#pragma unroll 8
            for (int r = 0; r < H; ++r) a = fmaf(dhp[r], j.W0[r * Cin + c], a);
            j.dgin[(size_t)b * Cin + c] = a * j.out_scale;
        }
    }
}
static inline int svnet_gate_bwd_chunks(int64_t Cin, int64_t H, int64_t Ov) {
    int64_t chunks = svnet_cdiv(H * Cin + Ov * H, 256 * 8);          // ~8 outputs per thread
    if (chunks > 16) chunks = 16;
    if (chunks < 1) chunks = 1;
    return (int)chunks;
}
static inline bool svnet_gate_fwd_job_ok(const svnet_gate_fwd_job* j) {
    return j && (j->gin || (j->gin_f64 && j->gin_out) || (j->rows && j->gin_out && j->R > 0 && j->Cin <= 256)) && j->W0 && j->W2 && j->h && j->gate && j->B >= 0 && j->Cin > 0 && j->H > 0 && j->Ov > 0 && j->H <= 256 && j->Ov <= 256;
}
static inline bool svnet_gate_bwd_job_ok(const svnet_gate_bwd_job* j) {
    return j && j->dgate && j->gate && j->h && j->gin && j->W0 && j->W2 && j->dW0 && j->dW2 && j->B >= 0 && j->Cin > 0 && j->H > 0 && j->Ov > 0 && j->H <= 256 && j->Ov <= 256;
}
