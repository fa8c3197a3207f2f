// Backward of the fused edge block (csrc/edgeblock.hip): one pass over the edges that RECOMPUTES the forward
// quantities from the point tables (nothing edge-sized was saved) and produces every gradient.
//
// Autograd of  get_graph_feature_sv -> SVBlock(binary) -> svpool  (sv_util.py:90-132, sv_layers.py:172-196), with the
// batch-statistic terms of both BatchNorms reduced at POINT level first (edgeblock_bwd_prelude / _coeffs):
//   scalar path: dL/dy_pre[e,o] = cs[o] * (g[e,o] - m1[o] - xhat[e,o]*m2[o]),  g = gy[p,o] on the arg-max edge, else 0
//   vector path: dL/dn'[e,c]    = direct + c0[c] + c1[c]*n'[e,c]
// Workgroup = 4 waves = one tile of 32 consecutive edge rows:
//   phase A (lanes = channels, 8 edges per wave): gather, re-binarize (ballots), popcount n, dL/dy_pre -> LDS (+ HBM for
//            the weight-gradient GEMM), ternary planes -> LDS -> row-sliced 32-bit halves in HBM; whole vector path
//            (dv' scattered to dU with float atomics, one contiguous row segment per wave-instruction);
//   phase B (MFMA): dx_b[32 x 320] = dy[32 x Os] . sign(W1)[Os x 320] on v_mfma_f32_32x32x16_bf16, exact 3-way split
//            of dy, STE mask applied from the LDS planes, result to LDS;
//   phase C (lanes = channels): beta / s / v2s backward, scatter-add to the point tables.
// The weight gradient GX = dy^T . x_b is left to mfma_tn_kernel (gemm_mfma.hip) on the dn_out + planes this kernel
// writes (4 B + 0.25 B per edge-channel instead of the 4 B fp32 input the layer-wise path keeps).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr float VEPS = 1e-6f;
constexpr int NW = 5;
constexpr int NCOL = NW * 64;      // 320 feature columns in fused bit order
constexpr int TE = 32;             // edges per tile
constexpr int DXS = NCOL + 4;      // LDS row stride of the dx tile (floats)

__device__ __forceinline__ int tdot(uint64_t xs, uint64_t xz, uint64_t ws, uint64_t wz) {
    const uint64_t m = xz & wz;
    return __popcll(m) - 2 * __popcll(m & (xs ^ ws));
}
__device__ __forceinline__ __bf16 bf16_from_bits(uint32_t b) { return __builtin_bit_cast(__bf16, (unsigned short)b); }

__device__ __forceinline__ void split3_frag(const float (&x)[8], bf16x8& fh, bf16x8& fm, bf16x8& fl) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t hu = __float_as_uint(x[j]) & 0xFFFF0000u;
        const float r1 = x[j] - __uint_as_float(hu);
        const uint32_t mu = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(mu);
        fh[j] = bf16_from_bits(hu >> 16);
        fm[j] = bf16_from_bits(mu >> 16);
        fl[j] = bf16_from_bits(__float_as_uint(r2) >> 16);
    }
}

// sign(W1) in fused column order as bf16 [NCOL][Os] (k = o contiguous: the MFMA B fragment is one 16-byte load)
__global__ void edgeblock_wbt_kernel(const uint64_t* __restrict__ w_sign, const uint64_t* __restrict__ w_nz, int Os,
                                     uint16_t* __restrict__ wbt) {
    const int total = NCOL * Os;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int col = e / Os, o = e - col * Os;
        const int w = col >> 6, b = col & 63;
        const uint64_t nz = w_nz[o * NW + w], sg = w_sign[o * NW + w];
        wbt[e] = ((nz >> b) & 1ull) ? (((sg >> b) & 1ull) ? 0x3F80 : 0xBF80) : 0;
    }
}

// ---------------------------------------------------------------------------------------------- point-level prelude
// gy[p,o] = Gs * lrelu'(y*) ; red[0:Os] += gy ; red[Os:2Os] += gy * xhat*      (y*, xhat* at the pooled edge)
// dgate[b,c] += sum_d Gv*(Av*mv + Bv*mvn) ; redv[0:Ov] += Gv*gate*mv ; redv[Ov:2Ov] += Gv*gate*mvn
__global__ __launch_bounds__(256) void edgeblock_bwd_prelude_kernel(
    const float* __restrict__ gs, const float* __restrict__ gv, const int32_t* __restrict__ n_max, const int32_t* __restrict__ n_min,
    const float* __restrict__ mv, const float* __restrict__ mvn, const float* __restrict__ coef, const float* __restrict__ scale1,
    const float* __restrict__ gate, int64_t P, int64_t N, int Os, int Ov, float slope, int64_t rows_per_block,
    float* __restrict__ gy, float* __restrict__ red, float* __restrict__ redv, float* __restrict__ dgate) {
    const float* A1 = coef; const float* B1 = coef + Os; const float* MY = coef + 2 * Os; const float* IY = coef + 3 * Os;
    const float* Av = coef + 4 * Os; const float* Bv = Av + Ov;
    const int64_t p0 = (int64_t)blockIdx.x * rows_per_block, p1 = min(P, p0 + rows_per_block);
    for (int o = threadIdx.x; o < Os; o += blockDim.x) {
        const float a = A1[o], bb = B1[o], sc = scale1[o], my = MY[o], iy = IY[o];
        float r1 = 0.f, r2 = 0.f;
        for (int64_t p = p0; p < p1; ++p) {
            const float sel = (float)(a >= 0.f ? n_max[p * Os + o] : n_min[p * Os + o]);
            const float y = a * sel + bb;
            const float g = gs[p * Os + o] * (y > 0.f ? 1.f : slope);
            gy[p * Os + o] = g;
            r1 += g;
            r2 += g * (sc * sel - my) * iy;
        }
        atomicAdd(&red[o], r1);
        atomicAdd(&red[Os + o], r2);
    }
    for (int c = threadIdx.x; c < Ov; c += blockDim.x) {
        const float av = Av[c], bv = Bv[c];
        float ra = 0.f, rb = 0.f, gsum = 0.f;
        int64_t cur_b = -1;
        for (int64_t p = p0; p < p1; ++p) {
            const int64_t b = p / N;
            if (b != cur_b) {
                if (cur_b >= 0) atomicAdd(&dgate[cur_b * Ov + c], gsum);
                cur_b = b;
                gsum = 0.f;
            }
            const float gt = gate[b * Ov + c];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int64_t q = (p * 3 + d) * Ov + c;
                const float g = gv[q], a = mv[q], n = mvn[q];
                gsum += g * (av * a + bv * n);
                ra += g * gt * a;
                rb += g * gt * n;
            }
        }
        if (cur_b >= 0) atomicAdd(&dgate[cur_b * Ov + c], gsum);
        atomicAdd(&redv[c], ra);
        atomicAdd(&redv[Ov + c], rb);
    }
}

// bcoef = [m1 | m2 | cs (Os each) | c0 | c1 (Ov each)];  BN parameter gradients written (not accumulated).
__global__ void edgeblock_bwd_coeffs_kernel(const float* __restrict__ red, const float* __restrict__ redv, const float* __restrict__ coef,
                                            const float* __restrict__ g1, const float* __restrict__ g2, int64_t E, int Os, int Ov,
                                            int training, float* __restrict__ bcoef, float* __restrict__ dg1, float* __restrict__ db1,
                                            float* __restrict__ dg2, float* __restrict__ db2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const float invE = 1.f / (float)E;
    if (c < Os) {
        const float iy = coef[3 * Os + c];
        bcoef[c] = training ? red[c] * invE : 0.f;
        bcoef[Os + c] = training ? red[Os + c] * invE : 0.f;
        bcoef[2 * Os + c] = g1[c] * iy;
        dg1[c] = red[Os + c];
        db1[c] = red[c];
    }
    if (c < Ov) {
        const float* Avp = coef + 4 * Os;
        const float mean = Avp[2 * Ov + c], is = Avp[3 * Ov + c];
        const float dAv = redv[c], dBv = redv[Ov + c];
        dg2[c] = dAv * is - dBv * mean * is;
        db2[c] = dBv;
        float c0 = 0.f, c1 = 0.f;
        if (training) {
            const float dmean = -g2[c] * is * dBv;
            const float dinv = g2[c] * (dAv - mean * dBv);
            const float dvar = -0.5f * is * is * is * dinv;
            c1 = 2.f * dvar * invE;
            c0 = dmean * invE - c1 * mean;
        }
        bcoef[3 * Os + c] = c0;
        bcoef[3 * Os + Ov + c] = c1;
    }
}

// ---------------------------------------------------------------------------------------------- edge pass
template <int OP>
__global__ __launch_bounds__(256) void edgeblock_bwd_kernel(svnet_edgeblock_bwd_desc d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Cs = d.Cs, Cv = d.Cv, Os = d.Os, Ov = d.Ov, k = (int)d.k;
    const int DNS = Os + 4;
    float* dnl = reinterpret_cast<float*>(smem);                    // [TE][DNS]   dL/dn = dy_pre*scale
    float* dxl = dnl + TE * DNS;                                     // [TE][DXS]   masked dx_b
    uint64_t* pl = reinterpret_cast<uint64_t*>(dxl + TE * DXS);      // [3][TE][NW] sign | nz | ste (row-major words)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t E = d.B * d.N * d.k;
    const int64_t e0 = (int64_t)blockIdx.x * TE;

    const float* A1 = d.coef; const float* MY = d.coef + 2 * Os; const float* IY = d.coef + 3 * Os;
    const float* Av = d.coef + 4 * Os; const float* Bv = Av + Ov;
    const float* M1 = d.bcoef; const float* M2 = d.bcoef + Os; const float* CS = d.bcoef + 2 * Os;
    const float* C0 = d.bcoef + 3 * Os; const float* C1 = C0 + Ov;

    const bool s_lane = lane < Cs, v2_lane = lane < 2 * Cv, diff_lane = lane < Cv, o_lane = lane < Ov;
    const int cm = diff_lane ? lane : lane - Cv;
    const float bd = d.beta_perm[lane], bc = d.beta_perm[64 + lane];
    float bv[3];
#pragma unroll
    for (int jz = 0; jz < 3; ++jz) bv[jz] = d.beta_perm[128 + 64 * jz + lane];

    // ================= phase A =================
    {
        uint64_t wsg[OP][NW], wnz[OP][NW];
        float a1[OP], my[OP], iy[OP], m1[OP], m2[OP], cs[OP], sc1[OP];
#pragma unroll
        for (int op = 0; op < OP; ++op) {
            const int o = lane + 64 * op;
            const bool ok = o < Os;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                wsg[op][w] = ok ? d.w_sign[o * NW + w] : 0ull;
                wnz[op][w] = ok ? d.w_nz[o * NW + w] : 0ull;
            }
            a1[op] = ok ? A1[o] : 0.f; my[op] = ok ? MY[o] : 0.f; iy[op] = ok ? IY[o] : 0.f;
            m1[op] = ok ? M1[o] : 0.f; m2[op] = ok ? M2[o] : 0.f; cs[op] = ok ? CS[o] : 0.f; sc1[op] = ok ? d.scale1[o] : 0.f;
        }
        const float avc = o_lane ? Av[lane] : 0.f, bvc = o_lane ? Bv[lane] : 0.f;
        const float c0 = o_lane ? C0[lane] : 0.f, c1 = o_lane ? C1[lane] : 0.f;
        const float invk = 1.f / (float)k;

        int64_t cur_p = -1;
        float cvsum[3] = {0.f, 0.f, 0.f};   // centre sum of dv' for the current point
        for (int rr = 0; rr < TE / 4; ++rr) {
            const int r = wave * (TE / 4) + rr;
            const int64_t e = e0 + r;
            if (e >= E) {  // wave-uniform: rows past the end contribute zeros
                for (int o = lane; o < Os; o += 64) dnl[r * DNS + o] = 0.f;
                if (lane < NW) { pl[(0 * TE + r) * NW + lane] = 0ull; pl[(1 * TE + r) * NW + lane] = 0ull; pl[(2 * TE + r) * NW + lane] = 0ull; }
                continue;
            }
            const int64_t gp = e / k;
            const int t = (int)(e - gp * k);
            const int64_t b = gp / d.N;
            const int64_t jloc = d.idx[e];
            if ((uint64_t)jloc >= (uint64_t)d.N) {  // corrupted neighbour id: never dereference it
                if (d.debug && lane == 0) {
                    if (atomicAdd(reinterpret_cast<unsigned long long*>(d.debug), 1ull) == 0ull) { d.debug[1] = e; d.debug[2] = jloc; d.debug[3] = d.N; }
                }
                for (int o = lane; o < Os; o += 64) dnl[r * DNS + o] = 0.f;
                if (lane < NW) { pl[(0 * TE + r) * NW + lane] = 0ull; pl[(1 * TE + r) * NW + lane] = 0ull; pl[(2 * TE + r) * NW + lane] = 0ull; }
                continue;
            }
            const int64_t gj = b * d.N + jloc;
            if (gp != cur_p) {
                if (cur_p >= 0 && o_lane) {
#pragma unroll
                    for (int dd = 0; dd < 3; ++dd) atomicAdd(&d.dvc[(cur_p * 3 + dd) * Ov + lane], cvsum[dd]);
                }
                cur_p = gp;
                cvsum[0] = cvsum[1] = cvsum[2] = 0.f;
            }
            // ---- recompute the edge row (same arithmetic as edgeblock_fwd_kernel)
            const float s_i = s_lane ? d.s[gp * Cs + lane] : 0.f;
            const float tc = s_i + bc;
            const float sd = (s_lane ? d.s[gj * Cs + lane] : 0.f) - s_i;
            const float td = sd + bd;
            uint64_t xs[NW], xz[NW], xt[NW];
            xs[1] = __ballot(s_lane && tc > 0.f); xz[1] = __ballot(s_lane && tc != 0.f); xt[1] = __ballot(s_lane && fabsf(tc) <= 1.2f);
            xs[0] = __ballot(s_lane && td > 0.f); xz[0] = __ballot(s_lane && td != 0.f); xt[0] = __ballot(s_lane && fabsf(td) <= 1.2f);
            float ve[3], z[3][3];
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
                const float vi = v2_lane ? d.v[(gp * 3 + dd) * Cv + cm] : 0.f;
                const float vj = diff_lane ? d.v[(gj * 3 + dd) * Cv + lane] : 0.f;
                ve[dd] = diff_lane ? (vj - vi) : vi;
                const float* zi = d.zz + (gp * 3 + dd) * 6;
                const float* zj = d.zz + (gj * 3 + dd) * 6;
#pragma unroll
                for (int jz = 0; jz < 3; ++jz) z[dd][jz] = zj[jz] + (zi[3 + jz] - zi[jz]);
            }
#pragma unroll
            for (int jz = 0; jz < 3; ++jz) {
                const float tv = ve[0] * z[0][jz] + ve[1] * z[1][jz] + ve[2] * z[2][jz] + bv[jz];
                xs[2 + jz] = __ballot(v2_lane && tv > 0.f);
                xz[2 + jz] = __ballot(v2_lane && tv != 0.f);
                xt[2 + jz] = __ballot(v2_lane && fabsf(tv) <= 1.2f);
            }
            if (lane < NW) {
                uint64_t a = xs[0], bq = xz[0], cq = xt[0];
#pragma unroll
                for (int w = 1; w < NW; ++w) {
                    if (lane == w) { a = xs[w]; bq = xz[w]; cq = xt[w]; }
                }
                pl[(0 * TE + r) * NW + lane] = a;
                pl[(1 * TE + r) * NW + lane] = bq;
                pl[(2 * TE + r) * NW + lane] = cq;
            }
            // ---- scalar path: n, dL/dy_pre
#pragma unroll
            for (int op = 0; op < OP; ++op) {
                const int o = lane + 64 * op;
                if (o < Os) {
                    int n = 0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) n += tdot(xs[w], xz[w], wsg[op][w], wnz[op][w]);
                    const int slot = (a1[op] >= 0.f) ? d.slot_max[gp * Os + o] : d.slot_min[gp * Os + o];
                    const float g = (slot == t) ? d.gy[gp * Os + o] : 0.f;
                    const float xh = (sc1[op] * (float)n - my[op]) * iy[op];
                    const float dyp = cs[op] * (g - m1[op] - xh * m2[op]);
                    d.dn_out[e * Os + o] = dyp;
                    dnl[r * DNS + o] = dyp * sc1[op];
                }
            }
            // ---- vector path: v' = U_j - U_i + T_i, out = gate * mean_k v'*(Av + Bv/n')
            if (o_lane) {
                float vp[3], ge[3];
                const float gt = d.gate[b * Ov + lane] * invk;
#pragma unroll
                for (int dd = 0; dd < 3; ++dd) {
                    const float* ui = d.ut + (gp * 3 + dd) * 2 * Ov;
                    vp[dd] = d.ut[(gj * 3 + dd) * 2 * Ov + lane] + (ui[Ov + lane] - ui[lane]);
                    ge[dd] = d.gv[(gp * 3 + dd) * Ov + lane] * gt;
                }
                const float nv = sqrtf(vp[0] * vp[0] + vp[1] * vp[1] + vp[2] * vp[2]);
                const float nn = nv + VEPS;
                const float q = avc + bvc / nn;
                const float gdot = ge[0] * vp[0] + ge[1] * vp[1] + ge[2] * vp[2];
                const float dnn = -gdot * bvc / (nn * nn) + c0 + c1 * nn;
                const float kk = nv > 0.f ? dnn / nv : 0.f;
#pragma unroll
                for (int dd = 0; dd < 3; ++dd) {
                    const float dvp = ge[dd] * q + kk * vp[dd];
                    atomicAdd(&d.du_acc[(gj * 3 + dd) * Ov + lane], dvp);
                    cvsum[dd] += dvp;
                }
            }
        }
        if (cur_p >= 0 && o_lane) {
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) atomicAdd(&d.dvc[(cur_p * 3 + dd) * Ov + lane], cvsum[dd]);
        }
    }
    __syncthreads();

    // ---- ternary planes of this tile -> row-sliced 32-bit halves (rows = the tile's 32 edges)
    {
        const int64_t word_row = e0 >> 6;
        const int half = (int)((e0 >> 5) & 1);
        for (int item = wave; item < 2 * NW; item += 4) {
            const int plane = item / NW, w = item - plane * NW;
            const uint64_t mine = (lane < TE) ? pl[(plane * TE + lane) * NW + w] : 0ull;
            uint32_t colword = 0u;
#pragma unroll 8
            for (int bb = 0; bb < 64; ++bb) {
                const uint64_t tb = __ballot((mine >> bb) & 1ull);
                if (lane == bb) colword = (uint32_t)tb;
            }
            uint32_t* dst = plane == 0 ? d.x_sign32 : d.x_nz32;
            dst[((word_row * NCOL) + w * 64 + lane) * 2 + half] = colword;
        }
    }

    // ================= phase B: dx_b = dnl . sign(W1), masked by the STE plane =================
    {
        const int r = lane & 31, h = lane >> 5;
        const int nks = (Os + 15) >> 4;
        const bf16x8* wbt = reinterpret_cast<const bf16x8*>(d.w1bt);   // [(col*Os + k) / 8]
        f32x16 acc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
        for (int ks = 0; ks < nks; ++ks) {
            float x[8];
            const int kk = ks * 16 + 8 * h;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (kk + j < Os) ? dnl[r * DNS + kk + j] : 0.f;
            bf16x8 fh, fm, fl;
            split3_frag(x, fh, fm, fl);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ct = wave + 4 * q;
                if (ct < NCOL / 32) {  // wave-uniform
                    bf16x8 bfr;
                    if (kk + 8 <= Os) {
                        bfr = wbt[((int64_t)(ct * 32 + r) * Os + kk) >> 3];
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) bfr[j] = bf16_from_bits(0);
                    }
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bfr, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fm, bfr, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, bfr, acc[q], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int ct = wave + 4 * q;
            if (ct < NCOL / 32) {
                const int col = ct * 32 + r;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                    const uint64_t st = pl[(2 * TE + row) * NW + (col >> 6)];
                    dxl[row * DXS + col] = ((st >> (col & 63)) & 1ull) ? acc[q][i] : 0.f;
                }
            }
        }
    }
    __syncthreads();

    // ================= phase C: scatter the input gradients =================
    {
        float dbd = 0.f, dbc = 0.f, dbv[3] = {0.f, 0.f, 0.f};
        int64_t cur_p = -1;
        float csum = 0.f;                       // centre part of ds for the current point (lane c < Cs)
        float cvd[3] = {0.f, 0.f, 0.f};         // centre part of dv  (lanes < 2Cv: diff lanes carry -sum, centre lanes +sum)
        float czq0 = 0.f, czq1 = 0.f, czq2 = 0.f, czq3 = 0.f, czq4 = 0.f, czq5 = 0.f, czq6 = 0.f, czq7 = 0.f, czq8 = 0.f;

// (kept as a macro: a by-reference lambda forces the per-point accumulators into scratch memory)
#define SVNET_FLUSH_POINT(p)                                                                                     \
    do {                                                                                                         \
        if (s_lane) atomicAdd(&d.ds_acc[(p) * Cs + lane], csum);                                                 \
        if (v2_lane) {                                                                                           \
            atomicAdd(&d.dv_acc[((p) * 3 + 0) * Cv + cm], cvd[0]);                                               \
            atomicAdd(&d.dv_acc[((p) * 3 + 1) * Cv + cm], cvd[1]);                                               \
            atomicAdd(&d.dv_acc[((p) * 3 + 2) * Cv + cm], cvd[2]);                                               \
        }                                                                                                        \
        if (lane < 9) {                                                                                          \
            float val_ = czq0;                                                                                   \
            val_ = (lane == 1) ? czq1 : val_; val_ = (lane == 2) ? czq2 : val_; val_ = (lane == 3) ? czq3 : val_; \
            val_ = (lane == 4) ? czq4 : val_; val_ = (lane == 5) ? czq5 : val_; val_ = (lane == 6) ? czq6 : val_; \
            val_ = (lane == 7) ? czq7 : val_; val_ = (lane == 8) ? czq8 : val_;                                   \
            atomicAdd(&d.dzc[(p) * 9 + lane], val_);                                                             \
        }                                                                                                        \
    } while (0)

        for (int rr = 0; rr < TE / 4; ++rr) {
            const int r = wave * (TE / 4) + rr;
            const int64_t e = e0 + r;
            if (e >= E) continue;
            const int64_t gp = e / k;
            const int64_t b = gp / d.N;
            const int64_t jloc = d.idx[e];
            if ((uint64_t)jloc >= (uint64_t)d.N) continue;
            const int64_t gj = b * d.N + jloc;
            if (gp != cur_p) {
                if (cur_p >= 0) SVNET_FLUSH_POINT(cur_p);
                cur_p = gp;
                csum = 0.f;
                cvd[0] = cvd[1] = cvd[2] = 0.f;
                czq0 = czq1 = czq2 = czq3 = czq4 = czq5 = czq6 = czq7 = czq8 = 0.f;
            }
            const float* row = dxl + r * DXS;
            const float gx0 = s_lane ? row[lane] : 0.f;            // d/d(s_j - s_i) through the binarization
            const float gx1 = s_lane ? row[64 + lane] : 0.f;       // d/d(s_i)
            dbd += gx0;
            dbc += gx1;
            const float gc0 = s_lane ? d.gconst[b * 2 * Cs + lane] : 0.f;        // gate path (not binarized)
            const float gc1 = s_lane ? d.gconst[b * 2 * Cs + Cs + lane] : 0.f;
            const float d0 = gx0 + gc0;
            if (s_lane) atomicAdd(&d.ds_acc[gj * Cs + lane], d0);
            csum += (gx1 + gc1) - d0;

            float gxv[3];
#pragma unroll
            for (int jz = 0; jz < 3; ++jz) {
                gxv[jz] = v2_lane ? row[128 + 64 * jz + lane] : 0.f;
                dbv[jz] += gxv[jz];
            }
            // v2s backward: s_v[c2][jz] = sum_d ve[d][c2] * z[d][jz]
            float ve[3], z[3][3];
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
                const float vi = v2_lane ? d.v[(gp * 3 + dd) * Cv + cm] : 0.f;
                const float vj = diff_lane ? d.v[(gj * 3 + dd) * Cv + lane] : 0.f;
                ve[dd] = diff_lane ? (vj - vi) : vi;
                const float* zi = d.zz + (gp * 3 + dd) * 6;
                const float* zj = d.zz + (gj * 3 + dd) * 6;
#pragma unroll
                for (int jz = 0; jz < 3; ++jz) z[dd][jz] = zj[jz] + (zi[3 + jz] - zi[jz]);
            }
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
                const float dve = gxv[0] * z[dd][0] + gxv[1] * z[dd][1] + gxv[2] * z[dd][2];
                if (diff_lane) {
                    atomicAdd(&d.dv_acc[(gj * 3 + dd) * Cv + lane], dve);
                    cvd[dd] -= dve;
                } else if (v2_lane) {
                    cvd[dd] += dve;
                }
            }
            const float dz0 = wave_sum(gxv[0] * ve[0]), dz1 = wave_sum(gxv[1] * ve[0]), dz2 = wave_sum(gxv[2] * ve[0]);
            const float dz3 = wave_sum(gxv[0] * ve[1]), dz4 = wave_sum(gxv[1] * ve[1]), dz5 = wave_sum(gxv[2] * ve[1]);
            const float dz6 = wave_sum(gxv[0] * ve[2]), dz7 = wave_sum(gxv[1] * ve[2]), dz8 = wave_sum(gxv[2] * ve[2]);
            czq0 += dz0; czq1 += dz1; czq2 += dz2; czq3 += dz3; czq4 += dz4; czq5 += dz5; czq6 += dz6; czq7 += dz7; czq8 += dz8;
            if (lane < 9) {
                float val = dz0;
                val = (lane == 1) ? dz1 : val; val = (lane == 2) ? dz2 : val; val = (lane == 3) ? dz3 : val;
                val = (lane == 4) ? dz4 : val; val = (lane == 5) ? dz5 : val; val = (lane == 6) ? dz6 : val;
                val = (lane == 7) ? dz7 : val; val = (lane == 8) ? dz8 : val;
                atomicAdd(&d.dzp_acc[gj * 9 + lane], val);
            }
        }
        if (cur_p >= 0) SVNET_FLUSH_POINT(cur_p);
#undef SVNET_FLUSH_POINT
        if (s_lane) {
            atomicAdd(&d.dbeta_perm[lane], dbd);
            atomicAdd(&d.dbeta_perm[64 + lane], dbc);
        }
        if (v2_lane) {
#pragma unroll
            for (int jz = 0; jz < 3; ++jz) atomicAdd(&d.dbeta_perm[128 + 64 * jz + lane], dbv[jz]);
        }
    }
}

}  // namespace

extern "C" int svnet_edgeblock_wbt_bf16(const uint64_t* w_sign, const uint64_t* w_nz, int64_t Os, uint16_t* wbt, void* stream) {
    SVNET_REQUIRE(w_sign && w_nz && wbt && Os > 0 && Os % 8 == 0, SVNET_E_ARG, "svnet_edgeblock_wbt_bf16: bad arguments (Os must be a multiple of 8)");
    hipLaunchKernelGGL(edgeblock_wbt_kernel, dim3((unsigned)svnet_cdiv(NCOL * Os, 256)), dim3(256), 0, (hipStream_t)stream, w_sign, w_nz,
                       (int)Os, wbt);
    SVNET_CHECK_LAUNCH("edgeblock_wbt_kernel");
    return SVNET_OK;
}

extern "C" int svnet_edgeblock_bwd_prelude_f32(const float* gs, const float* gv, const int32_t* n_max, const int32_t* n_min,
                                               const float* mv, const float* mvn, const float* coef, const float* scale1,
                                               const float* gate, int64_t P, int64_t N, int64_t Os, int64_t Ov, float slope,
                                               float* gy, float* red, float* redv, float* dgate, void* stream) {
    SVNET_REQUIRE(gs && gv && n_max && n_min && mv && mvn && coef && scale1 && gate && gy && red && redv && dgate, SVNET_E_ARG,
                  "svnet_edgeblock_bwd_prelude_f32: null pointer");
    SVNET_REQUIRE(P > 0 && N > 0 && Os > 0 && Ov > 0, SVNET_E_ARG, "svnet_edgeblock_bwd_prelude_f32: bad sizes");
    int64_t blocks = svnet_cdiv(P, 16);
    if (blocks > 2048) blocks = 2048;
    const int64_t rpb = svnet_cdiv(P, blocks);
    blocks = svnet_cdiv(P, rpb);
    hipLaunchKernelGGL(edgeblock_bwd_prelude_kernel, dim3((unsigned)blocks), dim3(128), 0, (hipStream_t)stream, gs, gv, n_max, n_min, mv,
                       mvn, coef, scale1, gate, P, N, (int)Os, (int)Ov, slope, rpb, gy, red, redv, dgate);
    SVNET_CHECK_LAUNCH("edgeblock_bwd_prelude_kernel");
    return SVNET_OK;
}

extern "C" int svnet_edgeblock_bwd_coeffs_f32(const float* red, const float* redv, const float* coef, const float* gamma1,
                                              const float* gamma2, int64_t E, int64_t Os, int64_t Ov, int training, float* bcoef,
                                              float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, void* stream) {
    SVNET_REQUIRE(red && redv && coef && gamma1 && gamma2 && bcoef && dgamma1 && dbeta1 && dgamma2 && dbeta2 && E > 0, SVNET_E_ARG,
                  "svnet_edgeblock_bwd_coeffs_f32: bad arguments");
    const int64_t n = Os > Ov ? Os : Ov;
    hipLaunchKernelGGL(edgeblock_bwd_coeffs_kernel, dim3((unsigned)svnet_cdiv(n, 128)), dim3(128), 0, (hipStream_t)stream, red, redv, coef,
                       gamma1, gamma2, E, (int)Os, (int)Ov, training, bcoef, dgamma1, dbeta1, dgamma2, dbeta2);
    SVNET_CHECK_LAUNCH("edgeblock_bwd_coeffs_kernel");
    return SVNET_OK;
}

extern "C" int svnet_edgeblock_bwd_f32(const svnet_edgeblock_bwd_desc* desc, void* stream) {
    SVNET_REQUIRE(desc, SVNET_E_ARG, "svnet_edgeblock_bwd_f32: null descriptor");
    const svnet_edgeblock_bwd_desc& d = *desc;
    SVNET_REQUIRE(d.s && d.v && d.idx && d.zz && d.ut && d.w_sign && d.w_nz && d.beta_perm && d.w1bt && d.scale1 && d.slot_max &&
                      d.slot_min && d.coef && d.gate && d.gy && d.bcoef && d.gv && d.gconst && d.dn_out && d.x_sign32 && d.x_nz32 &&
                      d.ds_acc && d.dv_acc && d.du_acc && d.dvc && d.dzp_acc && d.dzc && d.dbeta_perm,
                  SVNET_E_ARG, "svnet_edgeblock_bwd_f32: null pointer");
    SVNET_REQUIRE(d.B >= 0 && d.N > 0 && d.k > 0 && d.k <= 255, SVNET_E_ARG, "svnet_edgeblock_bwd_f32: bad sizes");
    SVNET_REQUIRE(d.Cs > 0 && d.Cs <= 64 && d.Cv > 0 && 2 * d.Cv <= 64 && d.Os > 0 && d.Os <= 128 && d.Os % 8 == 0 && d.Ov > 0 &&
                      d.Ov <= 64, SVNET_E_UNSUPPORTED, "svnet_edgeblock_bwd_f32: channel counts outside Cs<=64, 2Cv<=64, Os<=128 (mult of 8), Ov<=64");
    const int64_t E = d.B * d.N * d.k;
    if (E == 0) return SVNET_OK;
    const size_t lds = (size_t)TE * (d.Os + 4) * 4 + (size_t)TE * DXS * 4 + (size_t)3 * TE * NW * 8;
    const unsigned grid = (unsigned)svnet_cdiv(E, TE);
    if (d.Os <= 64) hipLaunchKernelGGL((edgeblock_bwd_kernel<1>), dim3(grid), dim3(256), lds, (hipStream_t)stream, d);
    else hipLaunchKernelGGL((edgeblock_bwd_kernel<2>), dim3(grid), dim3(256), lds, (hipStream_t)stream, d);
    SVNET_CHECK_LAUNCH("edgeblock_bwd_kernel");
    return SVNET_OK;
}
