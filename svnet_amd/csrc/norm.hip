// Batch-statistic normalisation over edge / point rows: BatchNorm1d(+LeakyReLU/ReLU) and VectorBN(+gate).
//
// Replaces  models/sv_layers.py:189-190 (bn1 + LeakyReLU), :81-102 (VectorBN), :194 (gate scaling) and
// their autograd.  All kernels are HBM-bound streaming passes over [M,C] / [M,3,C] rows; statistics are
// accumulated in fp64 (per-thread registers -> LDS -> one atomic per column per workgroup) so that
// 655 360-row batch statistics do not lose digits to fp32 summation order.
//
// Thread mapping for the column-reduction kernels: CW = next pow2 >= min(C,256) consecutive threads
// cover consecutive channels of one row (coalesced), 256/CW "row lanes" walk the rows.
#include "common.h"

namespace {

constexpr float VEPS = 1e-6f;  // sv_layers.py:18

struct ColMap {
    int cw_shift;  // log2(CW)
    __host__ static ColMap make(int64_t C) {
        ColMap m;
        m.cw_shift = 0;
        while ((1 << m.cw_shift) < C && m.cw_shift < 8) ++m.cw_shift;
        return m;
    }
};

__device__ __forceinline__ float act_grad(float z, int act, float slope) {
    if (act == 1) return z > 0.f ? 1.f : slope;
    if (act == 2) return z > 0.f ? 1.f : 0.f;
    return 1.f;
}

// Block-level reduction of per-thread doubles over the row lanes, then one atomic per column.
template <int NQ>
__device__ __forceinline__ void block_col_reduce(double (&val)[NQ], int col_in, int rl, int RL, int CW, bool col_ok,
                                                 double* lds /*[NQ][256]*/, double* out0, double* out1, int64_t col,
                                                 float* fout0, float* fout1) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) lds[q * 256 + rl * CW + col_in] = val[q];
    __syncthreads();
    if (rl == 0 && col_ok) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            double s = 0.0;
            for (int r = 0; r < RL; ++r) s += lds[q * 256 + r * CW + col_in];
            if (q == 0) {
                if (out0) svnet_slice_add(&out0[col], s);
                if (fout0) svnet_slice_add(&fout0[col], (float)s);
            } else {
                if (out1) svnet_slice_add(&out1[col], s);
                if (fout1) svnet_slice_add(&fout1[col], (float)s);
            }
        }
    }
}

// sums[0:C] += sum x (or n), sums[C:2C] += sum x^2 (or n^2)
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ x, int64_t M, int64_t C, int kind, int cw_shift,
                                                       int64_t rows_per_block, double* __restrict__ sums) {
    __shared__ double lds[2 * 256];
    const int CW = 1 << cw_shift, RL = 256 >> cw_shift;
    const int col_in = threadIdx.x & (CW - 1), rl = threadIdx.x >> cw_shift;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    for (int64_t cbase = 0; cbase < C; cbase += CW) {
        const int64_t c = cbase + col_in;
        const bool ok = c < C;
        double acc[2] = {0.0, 0.0};
        if (ok) {
            int64_t r = r0 + rl;
            // 16 (12) loads in flight per thread: with the 512-workgroup cap that is 8 MB on the wire - a few per thread left
            // the memory pipe mostly empty (1.7-2.4 TB/s)
            if (kind == 0) {
                for (; r + 15 * RL < r1; r += 16 * RL) {
                    float vv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) vv[u] = x[(r + u * RL) * C + c];
#pragma unroll
                    for (int u = 0; u < 16; ++u) { acc[0] += (double)vv[u]; acc[1] += (double)vv[u] * (double)vv[u]; }
                }
            } else {
                for (; r + 3 * RL < r1; r += 4 * RL) {
                    float va[4], vb[4], vd[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float* p3 = x + ((r + u * RL) * 3) * C + c;
                        va[u] = p3[0]; vb[u] = p3[C]; vd[u] = p3[2 * C];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float v = sqrtf(va[u] * va[u] + vb[u] * vb[u] + vd[u] * vd[u]) + VEPS;
                        acc[0] += (double)v; acc[1] += (double)v * (double)v;
                    }
                }
            }
#pragma unroll 2
            for (; r < r1; r += RL) {
                float v;
                if (kind == 0) {
                    v = x[r * C + c];
                } else {
                    const float a = x[(r * 3 + 0) * C + c], b = x[(r * 3 + 1) * C + c], d = x[(r * 3 + 2) * C + c];
                    v = sqrtf(a * a + b * b + d * d) + VEPS;
                }
                acc[0] += (double)v;
                acc[1] += (double)v * (double)v;
            }
        }
        double* sl = svnet_slice_ptr(sums, 2 * (int)C);
        block_col_reduce<2>(acc, col_in, rl, RL, CW, ok, lds, sl, sl + C, c, nullptr, nullptr);
    }
}

__global__ void bn_finalize_kernel(double* __restrict__ sums, int64_t M, int64_t C, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ rmean,
                                   float* __restrict__ rvar, long long* __restrict__ nbt) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    // `sums`: a sliced accumulator a previous kernel filled (colstats_kernel, binlinear_i8_fwd_kernel); the totals are left in its first 2C
    const double t0 = svnet_slices_total(sums, 2 * (int)C, (int)c), t1 = svnet_slices_total(sums, 2 * (int)C, (int)(C + c));
    sums[c] = t0;
    sums[C + c] = t1;
    const double m = t0 / (double)M;
    double var = t1 / (double)M - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
    if (rvar) {
        const double unb = (M > 1) ? var * ((double)M / (double)(M - 1)) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, int64_t C, float eps,
                                     float* __restrict__ mean, float* __restrict__ invstd) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = rmean[c];
    invstd[c] = 1.f / sqrtf(rvar[c] + eps);
}

// Thread mapping of the ELEMENTWISE kernels below (same as the reductions'): CW consecutive threads cover consecutive channels of a
// row, the 256/CW row lanes and the workgroups stride over the rows; the channel's constants sit in registers.  (The flat
// e -> (e / C, e % C) mapping these kernels had costs two 64-bit divisions - ~300 instructions - per element: bn_act_fwd ran at
// 2.2 TB/s on the [32768, 512] tensor of conv5.)
#define SVNET_ELEM_PROLOGUE()                                                                  \
    const int CW = 1 << cw_shift, RL = 256 >> cw_shift;                                        \
    const int col_in = threadIdx.x & (CW - 1), rl = threadIdx.x >> cw_shift;                   \
    const int64_t rstart = (int64_t)blockIdx.x * RL + rl, rstride = (int64_t)gridDim.x * RL

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int64_t M, int64_t C, int act,
                                                         float slope, int cw_shift, float* __restrict__ y) {
    SVNET_ELEM_PROLOGUE();
    for (int64_t c = col_in; c < C; c += CW) {
        const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
#pragma unroll 4
        for (int64_t r = rstart; r < M; r += rstride) {
            float z = (x[r * C + c] - mu) * is * ga + be;
            if (act == 1) z = z > 0.f ? z : z * slope;
            else if (act == 2) z = z > 0.f ? z : 0.f;
            y[r * C + c] = z;
        }
    }
}

__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                int64_t M, int64_t C, int act, float slope, int cw_shift,
                                                                int64_t rows_per_block, float* __restrict__ red) {
    __shared__ double lds[2 * 256];
    const int CW = 1 << cw_shift, RL = 256 >> cw_shift;
    const int col_in = threadIdx.x & (CW - 1), rl = threadIdx.x >> cw_shift;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    for (int64_t cbase = 0; cbase < C; cbase += CW) {
        const int64_t c = cbase + col_in;
        const bool ok = c < C;
        double acc[2] = {0.0, 0.0};
        if (ok) {
            const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
            int64_t r = r0 + rl;
            for (; r + 7 * RL < r1; r += 8 * RL) {   // eight rows' loads (16 values) in flight per thread
                float xv[8], gv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { xv[u] = x[(r + u * RL) * C + c]; gv[u] = g[(r + u * RL) * C + c]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float xh = (xv[u] - mu) * is;
                    const float gp = gv[u] * act_grad(xh * ga + be, act, slope);
                    acc[0] += (double)gp;
                    acc[1] += (double)gp * (double)xh;
                }
            }
            for (; r < r1; r += RL) {
                const float xh = (x[r * C + c] - mu) * is;
                const float gp = g[r * C + c] * act_grad(xh * ga + be, act, slope);
                acc[0] += (double)gp;
                acc[1] += (double)gp * (double)xh;
            }
        }
        float* sl = svnet_slice_ptr(red, 2 * (int)C);
        block_col_reduce<2>(acc, col_in, rl, RL, CW, ok, lds, nullptr, nullptr, c, sl, sl + C);
    }
}

__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ red, int64_t M, int64_t C, int act,
                                                               float slope, int train_stats, int cw_shift, float* __restrict__ dx) {
    SVNET_ELEM_PROLOGUE();
    const float invM = 1.f / (float)M;
    for (int64_t c = col_in; c < C; c += CW) {
        // (`red`: the slices the reduce kernel filled; workgroup 0 leaves the totals - dL/dbeta, dL/dgamma - in its first 2C elements)
        const float r0c = svnet_slices_total(red, 2 * (int)C, (int)c), r1c = svnet_slices_total(red, 2 * (int)C, (int)(C + c));
        if (blockIdx.x == 0 && rl == 0) { red[c] = r0c; red[C + c] = r1c; }
        const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
#pragma unroll 4
        for (int64_t r = rstart; r < M; r += rstride) {
            const float xh = (x[r * C + c] - mu) * is;
            float gp = g[r * C + c] * act_grad(xh * ga + be, act, slope);
            if (train_stats) gp -= (r0c + xh * r1c) * invM;
            dx[r * C + c] = gp * ga * is;
        }
    }
}

// ------------------------------------------------------------------------------------------------ VectorBN

// (cloud of a row without a division: the thread's rows advance by a fixed stride, so (cloud, row inside it) advance with it)
#define SVNET_CLOUD_CURSOR()                                                  \
    int64_t cb = rstart / rpb, crem = rstart - cb * rpb;                      \
    const int64_t cstep_b = rstride / rpb, cstep_r = rstride - cstep_b * rpb
#define SVNET_CLOUD_ADVANCE()                                                 \
    do { cb += cstep_b; crem += cstep_r; if (crem >= rpb) { crem -= rpb; ++cb; } } while (0)

// Statistics of the fused form (svnet_vbn_fwd_stats_f32): every thread derives mean / invstd of its channel from the fp64 sums itself
// (bn_finalize_kernel's arithmetic) and workgroup 0 keeps them + updates the running statistics - the one-workgroup finalize launch
// between the statistics pass and this kernel is gone (beside a kernel that fills the chip it waited up to 60 us for a free CU).
struct VbnStats {      // sums: the sliced accumulator colstats_kernel filled - every thread adds the slices of its channel up itself
    const double* sums; float* mean_out; float* invstd_out; float* rmean; float* rvar; long long* nbt;
    float eps, momentum;
};

template <bool FUSED>
__global__ __launch_bounds__(256) void vbn_fwd_kernel(const float* __restrict__ v, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ gate,
                                                      int64_t rpb, int64_t M, int64_t C, int cw_shift, float* __restrict__ out, VbnStats st) {
    SVNET_ELEM_PROLOGUE();
    if (FUSED && blockIdx.x == 0 && threadIdx.x == 0 && st.nbt) *st.nbt += 1;
    for (int64_t c = col_in; c < C; c += CW) {
        float mu, is;
        if (FUSED) {
            const double m = svnet_slices_total(st.sums, 2 * (int)C, (int)c) / (double)M;
            double var = svnet_slices_total(st.sums, 2 * (int)C, (int)(C + c)) / (double)M - m * m;
            if (var < 0.0) var = 0.0;
            mu = (float)m;
            is = (float)(1.0 / sqrt(var + (double)st.eps));
            if (blockIdx.x == 0 && rl == 0) {
                st.mean_out[c] = mu;
                st.invstd_out[c] = is;
                if (st.rmean) st.rmean[c] = (1.f - st.momentum) * st.rmean[c] + st.momentum * (float)m;
                if (st.rvar) {
                    const double unb = (M > 1) ? var * ((double)M / (double)(M - 1)) : var;
                    st.rvar[c] = (1.f - st.momentum) * st.rvar[c] + st.momentum * (float)unb;
                }
            }
        } else {
            mu = mean[c]; is = invstd[c];
        }
        const float ga = gamma[c], be = beta[c];
        SVNET_CLOUD_CURSOR();
#pragma unroll 2
        for (int64_t m = rstart; m < M; m += rstride) {
            const float a = v[(m * 3 + 0) * C + c], b = v[(m * 3 + 1) * C + c], d = v[(m * 3 + 2) * C + c];
            const float n = sqrtf(a * a + b * b + d * d) + VEPS;
            const float r = (n - mu) * is * ga + be;
            // reference order: v / n * n_bn (* gate)
            const float gt = gate ? gate[cb * C + c] : 1.f;
            out[(m * 3 + 0) * C + c] = a / n * r * gt;
            out[(m * 3 + 1) * C + c] = b / n * r * gt;
            out[(m * 3 + 2) * C + c] = d / n * r * gt;
            SVNET_CLOUD_ADVANCE();
        }
    }
}

__global__ __launch_bounds__(256) void vbn_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ gate, int64_t rpb, int64_t M, int64_t C,
                                                             int cw_shift, int64_t rows_per_block, float* __restrict__ red,
                                                             float* __restrict__ dgate) {
    __shared__ double lds[2 * 256];
    const int CW = 1 << cw_shift, RL = 256 >> cw_shift;
    const int col_in = threadIdx.x & (CW - 1), rl = threadIdx.x >> cw_shift;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    for (int64_t cbase = 0; cbase < C; cbase += CW) {
        const int64_t c = cbase + col_in;
        const bool ok = c < C;
        double acc[2] = {0.0, 0.0};
        if (ok) {
            const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
            int64_t cur_b = -1, b_end = 0;               // rows [.., b_end) belong to cloud cur_b
            float gsum = 0.f, gt = 1.f;
            int64_t r = r0 + rl;
            while (r < r1) {
                if (r >= b_end) {                         // next cloud: one division per cloud, not per row
                    if (cur_b >= 0 && dgate) atomicAdd(&dgate[cur_b * C + c], gsum);
                    cur_b = r / rpb;
                    b_end = (cur_b + 1) * rpb;
                    gsum = 0.f;
                    gt = gate ? gate[cur_b * C + c] : 1.f;
                }
                // up to four rows of this cloud at a time: their 24 loads are issued before the first use
                const int64_t lim = min(r1, b_end);
                float a0[4], a1[4], a2[4], g0[4], g1[4], g2[4];
                int nb = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t ru = r + (int64_t)u * RL;
                    const bool in = ru < lim;
                    const int64_t rc = in ? ru : r;
                    a0[u] = v[(rc * 3 + 0) * C + c]; a1[u] = v[(rc * 3 + 1) * C + c]; a2[u] = v[(rc * 3 + 2) * C + c];
                    g0[u] = g[(rc * 3 + 0) * C + c]; g1[u] = g[(rc * 3 + 1) * C + c]; g2[u] = g[(rc * 3 + 2) * C + c];
                    nb += in ? 1 : 0;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u < nb) {
                        const float n = sqrtf(a0[u] * a0[u] + a1[u] * a1[u] + a2[u] * a2[u]) + VEPS;
                        const float nh = (n - mu) * is;
                        const float rr = nh * ga + be;
                        const float gv = g0[u] * a0[u] + g1[u] * a1[u] + g2[u] * a2[u];  // sum_i g_i v_i
                        gsum += gv * (rr / n);
                        const float dr = gv * gt / n;
                        acc[0] += (double)dr;
                        acc[1] += (double)dr * (double)nh;
                    }
                }
                r += (int64_t)nb * RL;
            }
            if (cur_b >= 0 && dgate) atomicAdd(&dgate[cur_b * C + c], gsum);
        }
        float* sl = svnet_slice_ptr(red, 2 * (int)C);
        block_col_reduce<2>(acc, col_in, rl, RL, CW, ok, lds, nullptr, nullptr, c, sl, sl + C);
    }
}

__global__ __launch_bounds__(256) void vbn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ gate, float* __restrict__ red,
                                                            int64_t rpb, int64_t M, int64_t C, int train_stats, int cw_shift,
                                                            float* __restrict__ dv) {
    SVNET_ELEM_PROLOGUE();
    const float invM = 1.f / (float)M;
    for (int64_t c = col_in; c < C; c += CW) {
        const float r0c = svnet_slices_total(red, 2 * (int)C, (int)c), r1c = svnet_slices_total(red, 2 * (int)C, (int)(C + c));
        if (blockIdx.x == 0 && rl == 0) { red[c] = r0c; red[C + c] = r1c; }
        const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
        SVNET_CLOUD_CURSOR();
#pragma unroll 2
        for (int64_t m = rstart; m < M; m += rstride) {
            const float a0 = v[(m * 3 + 0) * C + c], a1 = v[(m * 3 + 1) * C + c], a2 = v[(m * 3 + 2) * C + c];
            const float gt = gate ? gate[cb * C + c] : 1.f;
            const float g0 = g[(m * 3 + 0) * C + c] * gt, g1 = g[(m * 3 + 1) * C + c] * gt, g2 = g[(m * 3 + 2) * C + c] * gt;
            const float nv = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
            const float n = nv + VEPS;
            const float nh = (n - mu) * is;
            const float rr = nh * ga + be;
            const float q = rr / n;
            const float dq = g0 * a0 + g1 * a1 + g2 * a2;
            float dr = dq / n;
            float dn = -dq * rr / (n * n);
            if (train_stats) dr -= (r0c + nh * r1c) * invM;
            dn += dr * ga * is;
            const float k = nv > 0.f ? dn / nv : 0.f;
            dv[(m * 3 + 0) * C + c] = g0 * q + k * a0;
            dv[(m * 3 + 1) * C + c] = g1 * q + k * a1;
            dv[(m * 3 + 2) * C + c] = g2 * q + k * a2;
            SVNET_CLOUD_ADVANCE();
        }
    }
}
#undef SVNET_CLOUD_CURSOR
#undef SVNET_CLOUD_ADVANCE
#undef SVNET_ELEM_PROLOGUE

// grid of an elementwise kernel in the (channel, row lane) mapping: >= 4 rows per thread, capped
inline void elem_geometry(int64_t M, int64_t C, int& cw_shift, unsigned& grid) {
    cw_shift = ColMap::make(C).cw_shift;
    const int RL = 256 >> cw_shift;
    grid = svnet_grid(svnet_cdiv(M, (int64_t)RL * 4) * 256, 256, 256 * 16);
}

inline void reduce_geometry(int64_t M, int64_t C, int& cw_shift, int64_t& rpb, unsigned& grid) {
    cw_shift = ColMap::make(C).cw_shift;
    const int RL = 256 >> cw_shift;
    int64_t blocks = svnet_cdiv(M, (int64_t)RL * 8);  // >= 8 rows per thread
    // every workgroup ends with one atomic per output column onto the SAME 2C addresses, and same-address atomics
    // serialise at the memory side: 512 workgroups (2 per CU) keep the loads flowing with 4x fewer of them than 2048
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    rpb = svnet_cdiv(svnet_cdiv(M, blocks), RL) * RL;
    grid = (unsigned)svnet_cdiv(M, rpb);
}

}  // namespace

extern "C" int svnet_colstats_f64(const float* x, int64_t M, int64_t C, int kind, double* sums, void* stream) {
    SVNET_REQUIRE(x && sums && M >= 0 && C > 0 && (kind == 0 || kind == 1), SVNET_E_ARG, "svnet_colstats_f64: bad arguments");
    if (M == 0) return SVNET_OK;
    int cw; int64_t rpb; unsigned grid;
    reduce_geometry(M, C, cw, rpb, grid);
    hipLaunchKernelGGL(colstats_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, M, C, kind, cw, rpb, sums);
    SVNET_CHECK_LAUNCH("colstats_kernel");
    return SVNET_OK;
}

extern "C" int svnet_bn_finalize_f32(double* sums, int64_t M, int64_t C, float eps, float momentum, float* mean,
                                     float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                     void* stream) {
    SVNET_REQUIRE(sums && mean && invstd && M > 0 && C > 0, SVNET_E_ARG, "svnet_bn_finalize_f32: bad arguments");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)svnet_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, M, C, eps,
                       momentum, mean, invstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked));
    SVNET_CHECK_LAUNCH("bn_finalize_kernel");
    return SVNET_OK;
}

extern "C" int svnet_bn_eval_stats_f32(const float* running_mean, const float* running_var, int64_t C, float eps, float* mean,
                                       float* invstd, void* stream) {
    SVNET_REQUIRE(running_mean && running_var && mean && invstd && C > 0, SVNET_E_ARG, "svnet_bn_eval_stats_f32: bad arguments");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((unsigned)svnet_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, running_mean,
                       running_var, C, eps, mean, invstd);
    SVNET_CHECK_LAUNCH("bn_eval_stats_kernel");
    return SVNET_OK;
}

extern "C" int svnet_bn_act_fwd_f32(const float* x, const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, int64_t M, int64_t C, int act, float slope, float* y, void* stream) {
    SVNET_REQUIRE(x && mean && invstd && gamma && beta && y && M >= 0 && C > 0, SVNET_E_ARG, "svnet_bn_act_fwd_f32: bad arguments");
    if (M == 0) return SVNET_OK;
    int cw; unsigned grid;
    elem_geometry(M, C, cw, grid);
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma, beta, M, C, act, slope, cw, y);
    SVNET_CHECK_LAUNCH("bn_act_fwd_kernel");
    return SVNET_OK;
}

extern "C" int svnet_bn_act_bwd_reduce_f32(const float* g, const float* x, const float* mean, const float* invstd,
                                           const float* gamma, const float* beta, int64_t M, int64_t C, int act, float slope,
                                           float* red, void* stream) {
    SVNET_REQUIRE(g && x && mean && invstd && gamma && beta && red && M >= 0 && C > 0, SVNET_E_ARG, "svnet_bn_act_bwd_reduce_f32: bad arguments");
    if (M == 0) return SVNET_OK;
    int cw; int64_t rpb; unsigned grid;
    reduce_geometry(M, C, cw, rpb, grid);
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, x, mean, invstd, gamma, beta, M, C,
                       act, slope, cw, rpb, red);
    SVNET_CHECK_LAUNCH("bn_act_bwd_reduce_kernel");
    return SVNET_OK;
}

extern "C" int svnet_bn_act_bwd_apply_f32(const float* g, const float* x, const float* mean, const float* invstd,
                                          const float* gamma, const float* beta, float* red, int64_t M, int64_t C, int act,
                                          float slope, int train_stats, float* dx, void* stream) {
    SVNET_REQUIRE(g && x && mean && invstd && gamma && beta && red && dx && M >= 0 && C > 0, SVNET_E_ARG, "svnet_bn_act_bwd_apply_f32: bad arguments");
    if (M == 0) return SVNET_OK;
    int cw; unsigned grid;
    elem_geometry(M, C, cw, grid);
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, x, mean, invstd, gamma, beta, red, M, C,
                       act, slope, train_stats, cw, dx);
    SVNET_CHECK_LAUNCH("bn_act_bwd_apply_kernel");
    return SVNET_OK;
}

extern "C" int svnet_vbn_fwd_f32(const float* v, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                 const float* gate, int64_t rows_per_batch, int64_t M, int64_t C, float* out, void* stream) {
    SVNET_REQUIRE(v && mean && invstd && gamma && beta && out && M >= 0 && C > 0 && rows_per_batch > 0, SVNET_E_ARG, "svnet_vbn_fwd_f32: bad arguments");
    if (M == 0) return SVNET_OK;
    int cw; unsigned grid;
    elem_geometry(M, C, cw, grid);
    hipLaunchKernelGGL(vbn_fwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, v, mean, invstd, gamma, beta, gate, rows_per_batch,
                       M, C, cw, out, VbnStats{});
    SVNET_CHECK_LAUNCH("vbn_fwd_kernel");
    return SVNET_OK;
}

extern "C" int svnet_vbn_fwd_stats_f32(const float* v, const double* sums, float eps, float momentum, float* mean, float* invstd,
                                       float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* gamma,
                                       const float* beta, const float* gate, int64_t rows_per_batch, int64_t M, int64_t C, float* out,
                                       void* stream) {
    SVNET_REQUIRE(v && sums && mean && invstd && gamma && beta && out && M > 0 && C > 0 && rows_per_batch > 0, SVNET_E_ARG,
                  "svnet_vbn_fwd_stats_f32: bad arguments");
    int cw; unsigned grid;
    elem_geometry(M, C, cw, grid);
    VbnStats st{sums, mean, invstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), eps, momentum};
    hipLaunchKernelGGL(vbn_fwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, v, nullptr, nullptr, gamma, beta, gate,
                       rows_per_batch, M, C, cw, out, st);
    SVNET_CHECK_LAUNCH("vbn_fwd_kernel");
    return SVNET_OK;
}

extern "C" int svnet_vbn_bwd_reduce_f32(const float* g, const float* v, const float* mean, const float* invstd,
                                        const float* gamma, const float* beta, const float* gate, int64_t rows_per_batch,
                                        int64_t M, int64_t C, float* red, float* dgate, void* stream) {
    SVNET_REQUIRE(g && v && mean && invstd && gamma && beta && red && M >= 0 && C > 0 && rows_per_batch > 0, SVNET_E_ARG, "svnet_vbn_bwd_reduce_f32: bad arguments");
    if (M == 0) return SVNET_OK;
    int cw; int64_t rpb; unsigned grid;
    reduce_geometry(M, C, cw, rpb, grid);
    hipLaunchKernelGGL(vbn_bwd_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, v, mean, invstd, gamma, beta, gate,
                       rows_per_batch, M, C, cw, rpb, red, dgate);
    SVNET_CHECK_LAUNCH("vbn_bwd_reduce_kernel");
    return SVNET_OK;
}

extern "C" int svnet_vbn_bwd_apply_f32(const float* g, const float* v, const float* mean, const float* invstd,
                                       const float* gamma, const float* beta, const float* gate, float* red,
                                       int64_t rows_per_batch, int64_t M, int64_t C, int train_stats, float* dv, void* stream) {
    SVNET_REQUIRE(g && v && mean && invstd && gamma && beta && red && dv && M >= 0 && C > 0 && rows_per_batch > 0, SVNET_E_ARG, "svnet_vbn_bwd_apply_f32: bad arguments");
    if (M == 0) return SVNET_OK;
    int cw; unsigned grid;
    elem_geometry(M, C, cw, grid);
    hipLaunchKernelGGL(vbn_bwd_apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, v, mean, invstd, gamma, beta, gate, red,
                       rows_per_batch, M, C, train_stats, cw, dv);
    SVNET_CHECK_LAUNCH("vbn_bwd_apply_kernel");
    return SVNET_OK;
}


// ---- the totals of a sliced accumulator where no consuming kernel of ours follows (svnet_hip.h SVNET_SLICED_LEN): buf[0:L] = sum of the slices
namespace {
template <typename T>
__global__ void slices_sum_kernel(T* __restrict__ buf, int L) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L) buf[i] = svnet_slices_total(buf, L, i);
}
}  // namespace

extern "C" int svnet_slices_sum_f32(float* buf, int64_t L, void* stream) {
    SVNET_REQUIRE(buf && L > 0 && L < ((int64_t)1 << 30), SVNET_E_ARG, "svnet_slices_sum_f32: bad arguments");
    hipLaunchKernelGGL(slices_sum_kernel<float>, dim3((unsigned)svnet_cdiv(L, 256)), dim3(256), 0, (hipStream_t)stream, buf, (int)L);
    SVNET_CHECK_LAUNCH("slices_sum_kernel");
    return SVNET_OK;
}

extern "C" int svnet_slices_sum_f64(double* buf, int64_t L, void* stream) {
    SVNET_REQUIRE(buf && L > 0 && L < ((int64_t)1 << 30), SVNET_E_ARG, "svnet_slices_sum_f64: bad arguments");
    hipLaunchKernelGGL(slices_sum_kernel<double>, dim3((unsigned)svnet_cdiv(L, 256)), dim3(256), 0, (hipStream_t)stream, buf, (int)L);
    SVNET_CHECK_LAUNCH("slices_sum_kernel");
    return SVNET_OK;
}
