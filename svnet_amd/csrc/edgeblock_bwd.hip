// Backward of the fused edge block (csrc/edgeblock.hip): every gradient of the layer from the point tables plus 2.4 B per
// edge-channel kept by the forward (no fp32 edge tensor exists in either direction).
//
// Autograd of  get_graph_feature_sv -> SVBlock(binary) -> svpool  (sv_util.py:90-132, sv_layers.py:172-196), with the
// batch-statistic terms of both BatchNorms reduced at POINT level first (edgeblock_bwd_prelude / _coeffs):
//   scalar path: dL/dy_pre[e,o] = cs[o] * (g[e,o] - m1[o] - xhat[e,o]*m2[o]),  g = gy[p,o] on the arg-max edge, else 0
//   vector path: dL/dn'[e,c]    = direct + c0[c] + c1[c]*n'[e,c]
// The forward kept n (int16) and the ternary / STE bit planes of every edge row, so nothing is re-binarized here.
// Two kernels:
//   edgeblock_bwd_vec_kernel (wave per point): the whole vector path; dv' scattered to dU with float atomics (contiguous
//            row segments), the centre sums written once per point;
//   edgeblock_bwd_kernel (workgroup = 4 waves = one tile of 32 consecutive edge rows):
//     phase A (elementwise, coalesced): dL/dy_pre from n -> HBM (operand of the weight-gradient GEMM) and LDS; planes -> LDS
//              -> row-sliced 32-bit halves in HBM (the other operand);
//     phase B (MFMA): dx_b[32 x 320] = dy[32 x Os] . sign(W1)[Os x 320] on v_mfma_f32_32x32x16_bf16, exact 3-way split
//              of dy, STE mask applied from the LDS planes, result to LDS;
//     phase C (lanes = channels): beta / s / v2s backward, scatter-add to the point tables.
// The weight gradient GX = dy^T . x_b is left to mfma_tn_kernel (gemm_mfma.hip) on the dn_out + planes this kernel
// writes (4 B + 0.25 B per edge-channel instead of the 4 B fp32 input the layer-wise path keeps).
#include <stdlib.h>

#include "common.h"
#include "gate_mlp.h"
#include "prelude.h"

namespace {

#define ATOMIC_ADD(p, v) do { if (MODE != 1) atomicAdd((p), (v)); else asm volatile("" :: "v"(v)); } while (0)

// Optional phase stopwatch (-DSVNET_PHASE_CLOCK, diagnostic builds only): thread 0 of every workgroup adds the cycles between
// phase boundaries to debug[8 + phase]; debug must then hold >= 8 + 256 * 16 entries (256 slices, summed by the reader: 20 480 workgroups adding to the SAME 16 addresses took 1.3 ms).  Marks of the tile kernel: 0 phase A (loads -> dy -> split ->
// LDS), 1 plane transposition, 5 phase B MFMAs, 6 the barrier behind them, 2 phase B epilogue (STE mask, dbeta, dx tile to LDS),
// 7 phase C pass 1 (scalar part), 9 passes 2 + 3 (neighbour rows, Vector2Scalar backward), 4 pass 4 (message rows, centre sums).
#ifdef SVNET_PHASE_CLOCK
// (the deltas are parked in LDS and added to debug[] in ONE go when the workgroup ends: an atomic per mark sat in the wave's vmcnt queue
//  in front of the next phase's loads - which retire in order - and 20 480 workgroups adding to the same eight addresses made each one
//  slow: the first table of this round charged that wait to whatever phase came next)
#define PHASE_MARK(i) do { if (threadIdx.x == 0) { const long long t_ = clock64(); ph_acc[i] += (unsigned long long)(t_ - ph_t); ph_t = t_; } } while (0)
#define PHASE_INIT() __shared__ unsigned long long ph_acc[16]; long long ph_t = 0, ph_t0 = 0, ph_w0 = 0; \
    if (threadIdx.x == 0) { for (int i_ = 0; i_ < 16; ++i_) ph_acc[i_] = 0ull; ph_t = ph_t0 = clock64(); ph_w0 = wall_clock64(); }
/* [14] = the workgroup's lifetime in clock64 ticks, [15] = in wall_clock64 ticks (constant 100 MHz): their ratio calibrates the tick */
#define PHASE_FLUSH() do { if (threadIdx.x == 0 && d.debug) { ph_acc[14] = (unsigned long long)(clock64() - ph_t0); \
        ph_acc[15] = (unsigned long long)(wall_clock64() - ph_w0); for (int i_ = 0; i_ < 16; ++i_) if (ph_acc[i_]) \
        atomicAdd(reinterpret_cast<unsigned long long*>(d.debug) + 8 + 16 * (blockIdx.x & 255) + i_, ph_acc[i_]); } } while (0)   /* 256 slices of 16: debug holds >= 8 + 4096 entries */
#else
#define PHASE_MARK(i) do { } while (0)
#define PHASE_INIT() do { } while (0)
#define PHASE_FLUSH() do { } while (0)
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// bf16 pieces of dL/dn that phase B of the tile kernel multiplies: 3 = the exact split (fp32-exact products); 2 = hi + mid only (a relative
// 2^-16 cut of every dL/dn entry: measured this round as an A/B build)
#ifndef SVNET_DN_PIECES
#define SVNET_DN_PIECES 3
#endif

constexpr float VEPS = 1e-6f;
constexpr int NW = 5;
constexpr int NCOL = NW * 64;      // 320 feature columns in fused bit order
constexpr int TE = 32;             // edges per tile
// LDS row stride (floats) of the dx tile.  Only the columns in use are kept, compacted as [d (Cs) | c (Cs) | v0 | v1 | v2 (2Cv each)];
// the stride is odd, so row-wise and column-wise walks are both free of bank conflicts.
__host__ __device__ __forceinline__ int dx_stride(int Cs, int Cv) { return (2 * Cs + 6 * Cv) | 1; }

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

// The same exact split for a pair of values, packed as two bf16 per word (low half = a): the phase-A producer of the tile kernel
// splits each dL/dn once and leaves the three bf16 planes in LDS (the four waves of phase B used to split the same 32 x Os tile
// four times over, 40 VALU instructions per k-step and wave).
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
    const float a1 = a - __uint_as_float(__float_as_uint(a) & 0xFFFF0000u), b1 = b - __uint_as_float(__float_as_uint(b) & 0xFFFF0000u);
    const float a2 = a1 - __uint_as_float(__float_as_uint(a1) & 0xFFFF0000u), b2 = b1 - __uint_as_float(__float_as_uint(b1) & 0xFFFF0000u);
    h = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);      // upper halves of (a, b)
    m = __builtin_amdgcn_perm(__float_as_uint(b1), __float_as_uint(a1), 0x07060302u);
    l = __builtin_amdgcn_perm(__float_as_uint(b2), __float_as_uint(a2), 0x07060302u);
}
// bf16 elements per row of the dL/dn planes in LDS: 16 bytes of padding put consecutive rows 4 banks apart (16-byte fragment reads)
__host__ __device__ __forceinline__ int dn_stride(int Os) { return Os + 8; }

// ---- cross-lane sums without the LDS crossbar (DPP modifiers + the gfx950 half / row swaps)
// (bound_ctrl for the controls that read a valid lane everywhere: the `old` operand is then dead and the DPP read folds into
//  the consuming add instead of costing a v_mov for `old` plus a v_mov_dpp)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF));
}
constexpr int DPP_QUAD_1032 = 0xB1, DPP_QUAD_2301 = 0x4E, DPP_ROW_ROR8 = 0x128, DPP_ROW_MIRROR = 0x140, DPP_ROW_HALF_MIRROR = 0x141,
              DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
// sum over the wave of a (lanes 0..31 of the result) and of b (lanes 32..63), each still spread over its 32 lanes
__device__ __forceinline__ float fold32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// rows (16 lanes) 0..3 of the result: a.row0+a.row1 | b.row0+b.row1 | a.row2+a.row3 | b.row2+b.row3
__device__ __forceinline__ float fold16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// index of the n-th (0-based) set bit among the low 10 bits of mask, or -1
__device__ __forceinline__ int nth_set_bit10(uint32_t mask, int n) {
    int res = -1;
#pragma unroll
    for (int b = 9; b >= 0; --b) {
        // count of set bits below b
        const int below = __popc(mask & ((1u << b) - 1u));
        if (((mask >> b) & 1u) && below == n) res = b;
    }
    return res;
}
// value held by lane (l ^ 32)
__device__ __forceinline__ uint32_t lane_half_swap(uint32_t x) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (threadIdx.x & 32) ? r[0] : r[1];
}
// Eight wave-wide sums in 18 VALU instructions: every lane of the 8-lane group g = lane >> 3 ends up with the sum over the
// wave of s[bitreverse3(g)]  (groups 0..7 hold s0, s4, s2, s6, s1, s5, s3, s7).
__device__ __forceinline__ float wave_sum8_packed(const float (&s)[8], int lane) {
    const float x0 = fold16(fold32(s[0], s[1]), fold32(s[2], s[3]));   // rows: s0 | s2 | s1 | s3
    const float x1 = fold16(fold32(s[4], s[5]), fold32(s[6], s[7]));   // rows: s4 | s6 | s5 | s7
    const bool hi = (lane & 8) != 0;
    float y = (hi ? x1 : x0) + dpp_get<DPP_ROW_ROR8>(hi ? x0 : x1);
    y += dpp_get<DPP_QUAD_1032>(y);
    y += dpp_get<DPP_QUAD_2301>(y);
    y += dpp_get<DPP_ROW_HALF_MIRROR>(y);
    return y;
}
// wave-wide sum, valid in lane 63
__device__ __forceinline__ float wave_sum_last(float v) {
    v += dpp_get<DPP_QUAD_1032>(v);
    v += dpp_get<DPP_QUAD_2301>(v);
    v += dpp_get<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_get<DPP_ROW_MIRROR>(v);
    v += dpp_get<DPP_ROW_BCAST15, 0xA>(v);
    v += dpp_get<DPP_ROW_BCAST31, 0xC>(v);
    return v;
}

// sign(W1) in fused column order as bf16, in MFMA B-FRAGMENT order: [column tile ct (10)][k-step ks (ceil(Os/16))][lane (64)][8],
// lane = h*32 + r holding column ct*32 + r, outputs o = ks*16 + 8h .. +7 (zero past Os).  A wave's fragment load is then 1 KiB
// contiguous (8 cache lines); with the plain [column][o] table every lane pair sat on its own line - 32 lines per load instruction,
// 24 such loads per wave and tile at Os = 128 - and the L1 tag pipe, not the matrix pipe, set the pace of phase B.
__global__ void edgeblock_wbt_kernel(const uint64_t* __restrict__ w_sign, const uint64_t* __restrict__ w_nz, int Os, int Cs, int Cv,
                                     uint16_t* __restrict__ wbt, uint32_t* __restrict__ w_dense) {
    // (optional) is every weight of the layer non-zero?  sign(0) = 0 makes the weights ternary in principle (sv_layers.py:44-45), but a
    // trained or freshly initialised layer holds no exact zero: the forward kernels then skip the weights' non-zero plane.  The planes are
    // complete here (the packing kernel ran before this one on the stream): the first wave checks the Os x 5 words against the columns in use.
    if (w_dense && blockIdx.x == 0 && threadIdx.x < 64) {
        bool ok = true;
        for (int i = threadIdx.x; i < Os * NW; i += 64) {
            const int w = i % NW, nb = w < 2 ? Cs : 2 * Cv;
            const uint64_t used = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
            ok = ok && ((w_nz[i] & used) == used);
        }
        const uint64_t bad = __ballot(!ok);
        if (threadIdx.x == 0) *w_dense = bad == 0ull ? 1u : 0u;
    }
    const int nks = (Os + 15) >> 4;
    const int total = (NCOL / 32) * nks * 512;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int j = e & 7, ln = (e >> 3) & 63, rest = e >> 9;
        const int ct = rest / nks, ks = rest - ct * nks;
        const int col = ct * 32 + (ln & 31), o = ks * 16 + 8 * (ln >> 5) + j;
        const int w = col >> 6, b = col & 63;
        uint16_t val = 0;
        if (o < Os) {
            const uint64_t nz = w_nz[o * NW + w], sg = w_sign[o * NW + w];
            val = ((nz >> b) & 1ull) ? (((sg >> b) & 1ull) ? 0x3F80 : 0xBF80) : 0;
        }
        wbt[e] = val;
    }
}

// ---------------------------------------------------------------------------------------------- point-level prelude
__global__ __launch_bounds__(256) void edgeblock_bwd_prelude_kernel(
    const float* __restrict__ gs, const float* __restrict__ gv, const int32_t* __restrict__ n_max, const int32_t* __restrict__ n_min,
    const float* __restrict__ mv, const float* __restrict__ mvn, const float* __restrict__ coef, const float* __restrict__ scale1,
    const float* __restrict__ gate, int64_t P, int64_t N, int Os, int Ov, float slope, int64_t rows_per_block,
    float* __restrict__ gy, float* __restrict__ red, float* __restrict__ redv, float* __restrict__ dgate,
    const float* __restrict__ gs2, int64_t gs2_ld, const float* __restrict__ gv2, int64_t gv2_ld, float* __restrict__ gv_sum) {
    svnet_prelude_body<int32_t>(gs, gv, n_max, n_min, mv, mvn, coef, scale1, gate, P, N, Os, Ov, slope, rows_per_block, gy, red, redv, dgate, gs2,
                                gs2_ld, gv2, gv2_ld, gv_sum);
}

// bcoef = [m1 | m2 | cs (Os each) | c0 | c1 (Ov each)];  BN parameter gradients written (not accumulated).
__global__ void edgeblock_bwd_coeffs_kernel(const float* __restrict__ red, const float* __restrict__ redv, const float* __restrict__ coef,
                                            const float* __restrict__ g1, const float* __restrict__ g2, int64_t E, int Os, int Ov,
                                            int training, const float* __restrict__ scale1, float* __restrict__ bcoef,
                                            float* __restrict__ dg1, float* __restrict__ db1, float* __restrict__ dg2,
                                            float* __restrict__ db2, svnet_gate_bwd_job job, int coef_blocks, int gate_chunks) {
    if ((int)blockIdx.x >= coef_blocks) {                              // the gate MLP's backward beside the coefficients
        const int g = (int)blockIdx.x - coef_blocks;
        svnet_gate_bwd_block(job, g / gate_chunks, g % gate_chunks, gate_chunks);
        return;
    }
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const float invE = 1.f / (float)E;
    if (c < Os) {
        const float iy = coef[3 * Os + c];
        float r0 = 0.f, r1 = 0.f;                      // the prelude's slices, added in a fixed order
#pragma unroll
        for (int sl = 0; sl < SVNET_RED_SLICES; ++sl) { r0 += red[sl * 2 * Os + c]; r1 += red[sl * 2 * Os + Os + c]; }
        const float m1 = training ? r0 * invE : 0.f, m2 = training ? r1 * invE : 0.f, cs = g1[c] * iy;
        bcoef[c] = m1;
        bcoef[Os + c] = m2;
        bcoef[2 * Os + c] = cs;
        dg1[c] = r1;
        db1[c] = r0;
        if (scale1) {   // per-channel constants of the binarized tile kernel: dy_pre = cs*g - (alpha + beta*n); pooled edge = arg-max / arg-min
            float* chc = bcoef + ((3 * Os + 2 * Ov + 3) & ~3);     // 16-byte aligned for the float4 reads of the tile kernel
            const float sc = scale1[c], my = coef[2 * Os + c];
            chc[c] = cs;
            chc[Os + c] = cs * (m1 - my * iy * m2);
            chc[2 * Os + c] = cs * sc * iy * m2;
            chc[3 * Os + c] = sc;
            chc[4 * Os + c] = coef[c] >= 0.f ? 1.f : 0.f;
        }
    }
    if (c < Ov) {
        const float* Avp = coef + 4 * Os;
        const float mean = Avp[2 * Ov + c], is = Avp[3 * Ov + c];
        float dAv = 0.f, dBv = 0.f;
#pragma unroll
        for (int sl = 0; sl < SVNET_RED_SLICES; ++sl) { dAv += redv[sl * 2 * Ov + c]; dBv += redv[sl * 2 * Ov + Ov + c]; }
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

__device__ __forceinline__ int msg_stride(int Cs, int Cv, int Ov) { (void)Ov; return ((Cs + 3 * Cv + 9) + 3) / 4 * 4; }   // = svnet_edgeblock_msg_stride

// ---------------------------------------------------------------------------------------------- vector path
// dL/dv' of every edge (v' = U_j - U_i + T_i, out = gate * mean_k v'*(Av + Bv/n'), n' = |v'| + eps):
//   ub_tab / ge_tab [P,3,Ov] = T_i - U_i and gv_i*gate/k (the gather kernel recomputes the neighbour's share of dv' from them),
//   dvc[i] = sum_k dv' (plain store: one wave owns a point).
// Wave per point like the forward; lanes = (edge slot g, channel c) with G = 64/Ov edges per wave-instruction.
struct VecArgs {
    svnet_edgeblock_bwd_desc d;
    int waves_per_cloud, points_per_wave;
};

template <int MODE>
__global__ __launch_bounds__(256) void edgeblock_bwd_vec_kernel(VecArgs va) {
    const svnet_edgeblock_bwd_desc& d = va.d;
    const int lane = threadIdx.x & 63;
    int64_t blk = blockIdx.x;
    if ((d.B & 7) == 0 && (va.waves_per_cloud & 3) == 0) {   // XCD-aware order, as in edgeblock_fwd_kernel
        const int64_t bpc = va.waves_per_cloud >> 2, xcd = blk & 7, slot = blk >> 3;
        blk = ((slot / bpc) * 8 + xcd) * bpc + (slot % bpc);
    }
    const int64_t wave_g = blk * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t b = wave_g / va.waves_per_cloud;
    if (b >= d.B) return;
    const int wi = (int)(wave_g - b * va.waves_per_cloud);
    const int p_begin = wi * va.points_per_wave, p_end = min((int)d.N, p_begin + va.points_per_wave);
    const int Os = d.Os, Ov = d.Ov, k = (int)d.k;
    const int G = 64 / Ov, g = lane / Ov, c = lane - g * Ov;
    const bool act = g < G;
    const float* Av = d.coef + 4 * Os; const float* Bv = Av + Ov;
    const float* C0 = d.bcoef + 3 * Os; const float* C1 = C0 + Ov;
    const float avc = Av[c], bvc = Bv[c], c0 = C0[c], c1 = C1[c];
    const float gtk = d.gate[b * Ov + c] / (float)k;

    // neighbour ids of a point: one coalesced load (lane t holds idx[p][t], k <= 64), requested a point ahead; each lane picks
    // its edge slot's id with a cross-lane read, so the row gathers never wait on a load of their own index
    const int lk = min(lane, k - 1);
    int jv_next = (p_begin < p_end) ? (int)d.idx[(b * d.N + p_begin) * k + lk] : 0;
    for (int p = p_begin; p < p_end; ++p) {
        const int64_t gp = b * d.N + p;
        const int jv = jv_next;
        if (p + 1 < p_end) jv_next = (int)d.idx[(gp + 1) * k + lk];
        const float* ui = d.ut + gp * 6 * Ov;
        const float ub0 = ui[0 * 2 * Ov + Ov + c] - ui[0 * 2 * Ov + c];   // T_i - U_i
        const float ub1 = ui[1 * 2 * Ov + Ov + c] - ui[1 * 2 * Ov + c];
        const float ub2 = ui[2 * 2 * Ov + Ov + c] - ui[2 * 2 * Ov + c];
        const float ge0 = d.gv[(gp * 3 + 0) * Ov + c] * gtk, ge1 = d.gv[(gp * 3 + 1) * Ov + c] * gtk, ge2 = d.gv[(gp * 3 + 2) * Ov + c] * gtk;
        if (lane < Ov) {   // per-point operands of dv', kept for the gather kernel (which visits this point from its neighbours' side)
            d.ub_tab[(gp * 3 + 0) * Ov + lane] = ub0; d.ub_tab[(gp * 3 + 1) * Ov + lane] = ub1; d.ub_tab[(gp * 3 + 2) * Ov + lane] = ub2;
            d.ge_tab[(gp * 3 + 0) * Ov + lane] = ge0; d.ge_tab[(gp * 3 + 1) * Ov + lane] = ge1; d.ge_tab[(gp * 3 + 2) * Ov + lane] = ge2;
        }
        float cv0 = 0.f, cv1 = 0.f, cv2 = 0.f;
        // neighbour rows one iteration ahead
        int64_t n_gj; bool n_ok; float n_u0, n_u1, n_u2;
#define SVNET_LOAD_U(T0)                                                                            \
    do {                                                                                            \
        const int t_ = (T0) + g;                                                                    \
        n_ok = act && t_ < k;                                                                       \
        int64_t jl_ = __shfl(jv, min(t_, k - 1));                                                   \
        if (!n_ok) jl_ = 0;                                                                         \
        if ((uint64_t)jl_ >= (uint64_t)d.N) { /* corrupted neighbour id: never dereference it */   \
            if (d.debug && c == 0) {                                                                \
                if (atomicAdd(reinterpret_cast<unsigned long long*>(d.debug), 1ull) == 0ull) {      \
                    d.debug[1] = gp * k + t_; d.debug[2] = jl_; d.debug[3] = d.N;                   \
                }                                                                                   \
            }                                                                                       \
            n_ok = false; jl_ = 0;                                                                  \
        }                                                                                           \
        n_gj = b * d.N + jl_;                                                                       \
        const float* uj_ = d.ut + n_gj * 6 * Ov;                                                    \
        n_u0 = uj_[0 * 2 * Ov + c]; n_u1 = uj_[1 * 2 * Ov + c]; n_u2 = uj_[2 * 2 * Ov + c];         \
    } while (0)
        SVNET_LOAD_U(0);
        for (int t0 = 0; t0 < k; t0 += G) {
            const bool ok = n_ok;
            const float u0 = n_u0, u1 = n_u1, u2 = n_u2;
            if (t0 + G < k) SVNET_LOAD_U(t0 + G);
            const float vp0 = u0 + ub0, vp1 = u1 + ub1, vp2 = u2 + ub2;
            const float nv = fast_sqrt(vp0 * vp0 + vp1 * vp1 + vp2 * vp2);
            const float nn = nv + VEPS;
            const float rn = fast_rcp(nn);
            const float q = avc + bvc * rn;
            const float gdot = ge0 * vp0 + ge1 * vp1 + ge2 * vp2;
            const float dnn = -gdot * bvc * rn * rn + c0 + c1 * nn;
            const float kk = nv > 0.f ? dnn * fast_rcp(nv) : 0.f;
            const float d0 = ge0 * q + kk * vp0, d1 = ge1 * q + kk * vp1, d2 = ge2 * q + kk * vp2;
            if (ok) { cv0 += d0; cv1 += d1; cv2 += d2; }   // (the neighbour's share of dv' is recomputed by the gather kernel from the tables)
        }
#undef SVNET_LOAD_U
        for (int gg = 1; gg < G; ++gg) {   // fold the edge slots: lanes g*Ov + c -> lane c
            const int src = (lane + gg * Ov) & 63;
            const float t0 = __shfl(cv0, src), t1 = __shfl(cv1, src), t2 = __shfl(cv2, src);   // all lanes take part
            if (lane < Ov) { cv0 += t0; cv1 += t1; cv2 += t2; }
        }
        if (lane < Ov) {
            d.dvc[(gp * 3 + 0) * Ov + lane] = cv0;
            d.dvc[(gp * 3 + 1) * Ov + lane] = cv1;
            d.dvc[(gp * 3 + 2) * Ov + lane] = cv2;
        }
    }
}

// ---------------------------------------------------------------------------------------------- edge pass (tiles)
// What one lane needs for one edge row in phase C.  The row index is wave-uniform (readfirstlane'd wave id), so gp / gj /
// the zz rows go through scalar loads and SGPR-based addressing.
struct EdgeIn {
    uint32_t gp, gj;                              // global point ids (P * 384 < 2^32 is checked on the host: 32-bit row offsets)
    bool valid, in_range;                         // in_range: e < E; valid: also a usable neighbour id
    uint32_t b;                                   // cloud
    float vj0, vj1, vj2;                          // neighbour's v (diff lanes only)
    float zj[9];                                  // Zp_j, RAW: anything computed from a load at request time waits for it there
};
// What phase C2 needs from the edge's OWN point: it changes once per k edges, so it is requested when the ring first meets
// the point (two iterations before its first edge is consumed) and turned into (vi, zc = Zq_i - Zp_i) when that edge is.
// One pending set is enough for k >= 3: the next point's first edge is requested after this one's has been consumed.
struct PointIn {
    float vi0, vi1, vi2;                          // lane c2 < 2Cv
    float zi[18];                                 // [Zp | Zq] rows interleaved as in zz (raw)
};

// Row cursor of a wave: edge id, its point / cloud / slot, advanced one edge at a time (no divisions, 32-bit scalar math: the
// CU's single scalar unit is shared by its four SIMDs and was the busiest unit of this kernel).
struct EdgeCursor {
    int64_t e;
    uint32_t gp, b;
    int t, pin;
};
// Position of edge row e0 + off (off < 64) given the position (gp0, t0, b0, pin0) of row e0: no division (a 64-bit integer
// division is ~160 instructions on this machine; the kernel used to do four per thread and tile in phase A and four per wave on
// the scalar unit - more instructions than the rest of the tile put together)
struct TilePos { uint32_t gp0, b0; int t0, pin0; uint32_t kmagic; };
// n / k for n < 1024, k <= 64:  (n * ceil(65536 / k)) >> 16 is exact while n * (k - 1) < 65536
__device__ __forceinline__ uint32_t small_div(uint32_t n, uint32_t kmagic) { return (n * kmagic) >> 16; }
__device__ __forceinline__ void cursor_init(const svnet_edgeblock_bwd_desc& d, const TilePos& tp, int64_t e0, int off, EdgeCursor& c) {
    const int k = (int)d.k, N = (int)d.N;
    c.e = e0 + off;
    const uint32_t n = (uint32_t)(tp.t0 + off);
    const uint32_t dq = small_div(n, tp.kmagic);
    c.gp = tp.gp0 + dq;
    c.t = (int)(n - dq * (uint32_t)k);
    int pin = tp.pin0 + (int)dq;
    uint32_t b = tp.b0;
    while (pin >= N) { pin -= N; ++b; }                  // (a tile covers at most 16 points)
    c.b = b;
    c.pin = pin;
}
__device__ __forceinline__ void cursor_next(const svnet_edgeblock_bwd_desc& d, EdgeCursor& c) {
    ++c.e;
    if (++c.t == (int)d.k) {
        c.t = 0;
        ++c.gp;
        if (++c.pin == (int)d.N) { c.pin = 0; ++c.b; }
    }
}

__device__ __forceinline__ void load_point(const svnet_edgeblock_bwd_desc& d, uint32_t gp, bool v2_lane, int cm, PointIn& pt) {
    const uint32_t Cv = (uint32_t)d.Cv, lc = v2_lane ? (uint32_t)cm : 0u;
    const uint32_t oi = gp * 3u * Cv + lc;
    pt.vi0 = d.v[oi];
    pt.vi1 = d.v[oi + Cv];
    pt.vi2 = d.v[oi + 2u * Cv];
    const float* zi = d.zz + gp * 18u;
#pragma unroll
    for (int q = 0; q < 18; ++q) pt.zi[q] = zi[q];
}

// jloc: the edge's neighbour id (wave-uniform, from the wave's pre-loaded id vector); loaded_p: the point whose PointIn was
// requested last (wave-uniform)
__device__ __forceinline__ void load_edge(const svnet_edgeblock_bwd_desc& d, const EdgeCursor& c, int jloc, int64_t E, int lane,
                                          bool v2_lane, int cm, EdgeIn& in, PointIn& pend, uint32_t& loaded_p) {
    const uint32_t Cv = (uint32_t)d.Cv, N = (uint32_t)d.N;
    // EVERY edge row issues its loads, rows past E and rows with a corrupted neighbour id from clamped addresses (point 0 /
    // neighbour 0 of the cloud), and is masked where it is consumed: with a path on which a ring slot's loads are skipped or
    // never consumed, the compiler's waitcnt pass drains the whole queue before it refills the slot.
    const bool in_range = c.e < E;
    const bool j_ok = (uint32_t)jloc < N;
    if (in_range && !j_ok && d.debug && lane == 0) {
        if (atomicAdd(reinterpret_cast<unsigned long long*>(d.debug), 1ull) == 0ull) { d.debug[1] = c.e; d.debug[2] = jloc; d.debug[3] = d.N; }
    }
    in.valid = in_range && j_ok;
    in.in_range = in_range;
    in.b = in_range ? c.b : 0u;
    in.gp = in_range ? c.gp : 0u;
    if (in_range && c.gp != loaded_p) { load_point(d, c.gp, v2_lane, cm, pend); loaded_p = c.gp; }
    in.gj = in.valid ? c.b * N + (uint32_t)jloc : 0u;
    // clamped lane indices: every lane issues every load (no exec-masked branches, loads go out back to back);
    // lanes outside a channel range read a valid neighbour element that is masked where it is consumed
    const uint32_t ld = (uint32_t)min(lane, (int)Cv - 1);
    const float* vj = d.v + in.gj * 3u * Cv;        // wave-uniform row base + 32-bit lane byte offsets: SGPR-base loads
    in.vj0 = ld_f32_sbase(vj, 4u * ld);
    in.vj1 = ld_f32_sbase(vj, 4u * (ld + Cv));
    in.vj2 = ld_f32_sbase(vj, 4u * (ld + 2u * Cv));
    const float* zj = d.zz + in.gj * 18u;
    in.zj[0] = zj[0];  in.zj[1] = zj[1];  in.zj[2] = zj[2];
    in.zj[3] = zj[6];  in.zj[4] = zj[7];  in.zj[5] = zj[8];
    in.zj[6] = zj[12]; in.zj[7] = zj[13]; in.zj[8] = zj[14];
}

// Phase C's global operands (NC2 > 0 form): the neighbour rows v_j of the wave's eight edges (lane = element of the [3][Cv] row) and, per
// lane = (edge, channel half, axis), the own point's v, the neighbour's Zp row and the own point's [Zp | Zq] row.  The neighbour id of
// the lane's edge comes from the wave's pre-loaded id vector (lane rr < 8 of jv8 holds idx[ew + rr]) through one cross-lane read - it used
// to be a global load of idx[e] in front of the zz-row load it addresses: two latencies in a row.
#define SVNET_PHASEC_REQUESTS()                                                                             \
    do {                                                                                                    \
            const uint32_t Nu_ = (uint32_t)d.N, uCv_ = (uint32_t)Cv; \
            const int row0_ = 8 * wave, n3_ = 3 * Cv; \
            { \
                EdgeCursor c2; \
                cursor_init(d, tp, e0, row0_, c2); \
_Pragma("unroll") \
                for (int rr = 0; rr < 8; ++rr) { \
                    const int jl = __builtin_amdgcn_readlane(jv8, rr); \
                    const bool ok = c2.e < E && (uint32_t)jl < Nu_; \
                    const uint32_t gj = ok ? c2.b * Nu_ + (uint32_t)jl : 0u; \
                    const float* vr = d.v + gj * 3u * uCv_; \
                    pc_vst[rr][0] = ld_f32_sbase(vr, 4u * (uint32_t)min(lane, n3_ - 1)); \
                    if (NC2 > 44) pc_vst[rr][1] = ld_f32_sbase(vr, 4u * (uint32_t)min(lane + 64, n3_ - 1)); \
                    cursor_next(d, c2); \
                } \
            } \
            const int el_ = lane >> 3, ch_ = (lane >> 2) & 1, qq_ = min(lane & 3, 2); \
            const int rw_ = row0_ + el_; \
            const bool in_range_ = e0 + rw_ < E; \
            const uint32_t dq_ = small_div((uint32_t)(tp.t0 + rw_), tp.kmagic); \
            int pin_ = tp.pin0 + (int)dq_; \
            uint32_t b_ = tp.b0; \
            while (pin_ >= (int)Nu_) { pin_ -= (int)Nu_; ++b_; } \
            const int jloc_ = __builtin_amdgcn_ds_bpermute(4 * el_, jv8); \
            const bool valid_ = in_range_ && (uint32_t)jloc_ < Nu_; \
            const uint32_t gj_ = valid_ ? b_ * Nu_ + (uint32_t)jloc_ : 0u; \
            const uint32_t gpc_ = in_range_ ? tp.gp0 + dq_ : 0u; \
            const float* vip_ = d.v + (gpc_ * 3u + (uint32_t)qq_) * uCv_; \
_Pragma("unroll") \
            for (int i = 0; i < PC_NCH; ++i) pc_vi[i] = vip_[min(ch_ * PC_NCH + i, Cv - 1)]; \
            const float* zjp_ = d.zz + gj_ * 18u + (uint32_t)qq_ * 6u; \
            const float* zip_ = d.zz + gpc_ * 18u + (uint32_t)qq_ * 6u; \
            pc_zj[0] = zjp_[0]; pc_zj[1] = zjp_[1]; pc_zj[2] = zjp_[2]; \
_Pragma("unroll") \
            for (int i = 0; i < 6; ++i) pc_zi[i] = zip_[i]; \
    } while (0)

// MODE: 0 = product; 1..3 = timing-only ablations (SVNET_BWD_MODE), wrong results.  NKS = k-steps of phase B (Os <= 16*NKS)
// NC2: 0 = phase C as one edge per wave iteration (lanes = channels); > 0 = the lanes = (edge, axis) form of phase C for 2 Cv <= NC2
// (20: Cv <= 10 - five channel pairs per lane; 24: Cv <= 12; 44: Cv <= 21 - a neighbour's [3][Cv] row fits one 64-lane load and 11 channel pairs per lane; 48: Cv <= 24)
// (One tile per workgroup.  A loop over two tiles per workgroup - so that the drain of a tile's last atomics and the next workgroup's
//  prologue lie under the next tile's phase A - was built this round and not kept: whatever form the loop took (by-value descriptor,
//  descriptor re-read through an opaque kernarg offset) the allocator kept 56 .. 382 registers live across the back edge and spilled
//  them; as a non-inlined function the callee-saved registers went to scratch twice per tile.)
template <int MODE, int NKS, int NC2 = 0>
__global__ __launch_bounds__(256, 4) void edgeblock_bwd_kernel(svnet_edgeblock_bwd_desc d) {
    const uint32_t tile_lin = blockIdx.x;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Cs = d.Cs, Cv = d.Cv, Os = d.Os;
    const int DNB = dn_stride(Os);
    const int DXS = dx_stride(Cs, Cv);
    const int dx_floats = max(TE * DXS, (3 * TE * DNB + 1) / 2);    // dnb (phase A -> B) shares the bytes of dxl
    float* dxl = reinterpret_cast<float*>(smem);                    // [TE][DXS]   masked dx_b            (phases B -> C)
    uint16_t* dnb = reinterpret_cast<uint16_t*>(smem);              // [3][TE][DNB] dL/dn = dy_pre*scale as hi | mid | lo bf16 planes (phases
                                                                     //             A -> B), ALIASES dxl: phase B has consumed it before dxl is written
    uint64_t* pl = reinterpret_cast<uint64_t*>(dxl + ((dx_floats + 3) & ~3));   // [3][TE][NW] sign | nz | ste (row-major words)

    PHASE_INIT();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // uniform: scalar loads / SGPR addressing
    const int64_t E = d.B * d.N * d.k;
    // XCD-aware tile order: workgroups b and b+8 share an XCD (round-robin dispatch), so give XCD x the clouds
    // x, x+8, ... one after the other: a cloud's point tables (~1.6 MB) then live in that XCD's 4 MiB L2
    // (32-bit arithmetic throughout: E < 2^31 is checked on the host)
    uint32_t tile = tile_lin;
    const uint32_t nk = (uint32_t)d.N * (uint32_t)d.k;
    const uint32_t tpc = nk / TE;                                    // tiles per cloud
    if ((d.B & 7) == 0 && tpc * TE == nk) {
        const uint32_t xcd = tile & 7u, slot = tile >> 3;
        const uint32_t sq = slot / tpc;
        tile = (sq * 8u + xcd) * tpc + (slot - sq * tpc);
    }
    const int64_t e0 = (int64_t)tile * TE;
    const int64_t ew = e0 + wave * (TE / 4);                          // first edge row of this wave
    TilePos tp;
    {
        const uint32_t ku = (uint32_t)d.k, Nu = (uint32_t)d.N, e0u = tile * (uint32_t)TE;
        tp.gp0 = e0u / ku;
        tp.t0 = (int)(e0u - tp.gp0 * ku);
        tp.b0 = tp.gp0 / Nu;
        tp.pin0 = (int)(tp.gp0 - tp.b0 * Nu);
        tp.kmagic = (65536u + ku - 1u) / ku;
    }

    const bool v2_lane = lane < 2 * Cv, diff_lane = lane < Cv;
    const int cm = diff_lane ? lane : lane - Cv;
    // neighbour ids of this wave's 8 edge rows for phase C2 (lane rr holds idx[ew + rr]): requested now, consumed through
    // v_readlane, so the row gathers of phase C2 never wait on a load of their own index
    // (the LOW dword of the int64 id only: with a 64-bit load the allocator recycled the dead high register at once, and the write-after-
    //  write hazard cost a full `s_waitcnt vmcnt(0)` - an HBM round trip - before any other load of the tile had been issued)
    const int jv8 = reinterpret_cast<const int*>(d.idx)[2 * min(ew + (lane & 7), E - 1)];

    // The 320 fused columns are 5 words x 64, of which only Cs / Cs / 2Cv / 2Cv / 2Cv are in use: 32-column tiles that lie
    // entirely in the padding (5 of 10 for the Cs = 32, Cv = 10 layers) are skipped by phase B, and the used ones are dealt
    // round-robin to the four waves.
    uint32_t used_tiles = 0;
#pragma unroll
    for (int ct = 0; ct < NCOL / 32; ++ct) {
        const int width = (ct >> 1) < 2 ? Cs : 2 * Cv;
        if (width > 32 * (ct & 1)) used_tiles |= 1u << ct;
    }
    int cts[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) cts[q] = __builtin_amdgcn_readfirstlane(nth_set_bit10(used_tiles, wave + 4 * q));

    // sign(W1) fragments of this wave's column tiles for phase B (NKS k-steps x 3 tiles x 4 VGPRs).  Up to Os = 64 they are
    // requested before anything else, so that their L2 latency is hidden behind phase A (at Os = 128 the 96 registers would
    // spill across phase A: loaded at the start of phase B instead)
    const int nks = (Os + 15) >> 4;
    constexpr int NKB = NKS < 4 ? NKS : 4;          // k-steps whose fragments are in registers at a time
    bf16x8 bfr[NKB][3];
#define SVNET_LOAD_BFR(KS0)                                                                                    \
    do {                                                                                                       \
        const bf16x8* wbt_ = reinterpret_cast<const bf16x8*>(d.w1bt); /* [ct][ks][lane] fragments */           \
        _Pragma("unroll") for (int kb = 0; kb < NKB; ++kb)                                                     \
            _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                    \
                const int ct = cts[q], ks_ = (KS0) + kb;                                                       \
                if (ks_ < nks && ct >= 0) {                                                                    \
                    bfr[kb][q] = wbt_[(ct * nks + ks_) * 64 + lane];                                           \
                } else {                                                                                       \
                    _Pragma("unroll") for (int j = 0; j < 8; ++j) bfr[kb][q][j] = bf16_from_bits(0);           \
                }                                                                                              \
            }                                                                                                  \
    } while (0)
    constexpr bool HOIST_B = NKS <= 4;
    if (HOIST_B) SVNET_LOAD_BFR(0);
    // the gate path's per-cloud constants of phase C's scalar pass (lane = scalar channel): requested HERE.  Loaded where they are used
    // - first thing in pass 1, behind the ~20 gather requests of phase C that are still in flight (vmcnt retires in order) - they made
    // that pass wait out the whole gather: 19 % (conv4) to 42 % (conv2) of a workgroup's time (phase clocks, profiles/r04_tile_phases.txt)
    float gc_pre0 = 0.f, gc_pre1 = 0.f;
    if constexpr (NC2 > 0) {
        const int sl_ = min(lane, Cs - 1);
        gc_pre0 = d.gconst[tp.b0 * 2u * (uint32_t)Cs + sl_];
        gc_pre1 = d.gconst[tp.b0 * 2u * (uint32_t)Cs + Cs + sl_];
    }

    // ================= phase A: dL/dy_pre of the tile's 32 x Os edge-channels from the saved integer sums =========
    //   dy_pre = cs*(g - m1 - xhat*m2), xhat = (scale*n - mean)*invstd, g = gy[p,o] on the pooled edge, else 0
    //          = cs*g - (alpha + beta*n)
    {
        // ternary / STE planes of the tile: [e][plane][word] in HBM -> [plane][row][word] in LDS.  REQUESTED here (two words per thread),
        // stored behind the other requests of phase A: as a load -> store loop the second word was requested after the first had arrived
        static_assert(TE * 3 * NW <= 512, "two plane words per thread");
        uint64_t plw[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int item = min(tid + 256 * u, TE * 3 * NW - 1);
            const int r = item / (3 * NW);
            plw[u] = d.planes[min(e0 + r, E - 1) * (3 * NW) + (item - r * (3 * NW))];
        }
        // this thread's edge-channel quads (<= 4: Os <= 128): their loads go out before the barrier, with the constants'
        const int O4 = Os >> 2;
        const int o4_shift = __builtin_ctz((unsigned)O4);   // Os is a power of two (checked on the host): item / O4 is a shift
        const int k = (int)d.k;
        // per-channel constants [cs | alpha | beta | scale | pooled-is-max] from svnet_edgeblock_bwd_coeffs_f32.  256 is a multiple
        // of Os/4 (Os in {32, 64, 128}: asserted on the host), so a thread's quads all sit on the same four channels
        const float* chc = d.bcoef + ((3 * Os + 2 * d.Ov + 3) & ~3);
        const int o4c = (tid & (O4 - 1)) << 2;
        const float4 cs = *reinterpret_cast<const float4*>(chc + o4c);
        const float4 al = *reinterpret_cast<const float4*>(chc + Os + o4c);
        const float4 be = *reinterpret_cast<const float4*>(chc + 2 * Os + o4c);
        const float4 sc = *reinterpret_cast<const float4*>(chc + 3 * Os + o4c);
        const float4 ps = *reinterpret_cast<const float4*>(chc + 4 * Os + o4c);
        constexpr int NI = NKS / 2;       // TE * (Os / 4) / 256 quads per thread
        // (n and the pooled slots stay PACKED until they are used: as short4 / uchar4 they were unpacked right behind their loads - a
        //  wait per quad, four dependent round trips per tile - profiles/r04_tile_phases.txt)
        uint2 n4[NI]; float4 gy4[NI]; uint32_t smx[NI], smn[NI]; int tt[NI];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int item = it * 256 + tid;
            const int r = item >> o4_shift, o4 = (item & (O4 - 1)) << 2;
            const int64_t e = e0 + r;
            tt[it] = -1;
            if (item < TE * O4 && e < E) {
                const uint32_t n_ = (uint32_t)(tp.t0 + r), dq_ = small_div(n_, tp.kmagic);
                const int64_t gp = (int64_t)(tp.gp0 + dq_);
                tt[it] = (int)(n_ - dq_ * (uint32_t)k);
                n4[it] = *reinterpret_cast<const uint2*>(d.n16 + e * Os + o4);
                gy4[it] = *reinterpret_cast<const float4*>(d.gy + gp * Os + o4);
                smx[it] = *reinterpret_cast<const uint32_t*>(d.slot_max + gp * Os + o4);
                smn[it] = *reinterpret_cast<const uint32_t*>(d.slot_min + gp * Os + o4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // every request of phase A is out before the first wait
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int item = tid + 256 * u;
            if (item < TE * 3 * NW) {
                const int r = item / (3 * NW), q = item - r * (3 * NW);
                const int plane = q / NW, w = q - plane * NW;
                pl[(plane * TE + r) * NW + w] = (e0 + r < E) ? plw[u] : 0ull;
            }
        }
        PHASE_MARK(11);   // phase A: requests issued
        __syncthreads();
        PHASE_MARK(12);   // phase A: planes in LDS (first barrier)
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int item = it * 256 + tid;
            if (item >= TE * O4) break;
            const int r = item >> o4_shift, o4 = (item & (O4 - 1)) << 2;
            const int64_t e = e0 + r;
            const int t = tt[it];
            float4 dn = make_float4(0.f, 0.f, 0.f, 0.f);
            if (Os == 8 && o4 == 0) {   // the k-step reads 16 columns: zero the 8 padding columns of the row (anything times a zero weight must be 0)
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(dnb + (p * TE + r) * DNB + 8) = make_uint4(0u, 0u, 0u, 0u);
            }
            if (t >= 0) {
                const uint32_t sx = smx[it], sn = smn[it];
                const float g0 = ((int)((ps.x != 0.f ? sx : sn) & 0xFFu) == t) ? gy4[it].x : 0.f;
                const float g1 = ((int)(((ps.y != 0.f ? sx : sn) >> 8) & 0xFFu) == t) ? gy4[it].y : 0.f;
                const float g2 = ((int)(((ps.z != 0.f ? sx : sn) >> 16) & 0xFFu) == t) ? gy4[it].z : 0.f;
                const float g3 = ((int)((ps.w != 0.f ? sx : sn) >> 24) == t) ? gy4[it].w : 0.f;
                const float n0 = (float)(short)(n4[it].x & 0xFFFFu), n1 = (float)((int)n4[it].x >> 16);
                const float n2 = (float)(short)(n4[it].y & 0xFFFFu), n3 = (float)((int)n4[it].y >> 16);
                float4 dy;
                dy.x = cs.x * g0 - (al.x + be.x * n0);
                dy.y = cs.y * g1 - (al.y + be.y * n1);
                dy.z = cs.z * g2 - (al.z + be.z * n2);
                dy.w = cs.w * g3 - (al.w + be.w * n3);
                if (d.dn_out) *reinterpret_cast<float4*>(d.dn_out + e * Os + o4) = dy;   // (optional: svnet_edgeblock_wgrad_f32 recomputes it)
                dn = make_float4(dy.x * sc.x, dy.y * sc.y, dy.z * sc.z, dy.w * sc.w);
            }
            uint32_t h0, m0, l0, h1, m1, l1;
            split3_pair(dn.x, dn.y, h0, m0, l0);
            split3_pair(dn.z, dn.w, h1, m1, l1);
            uint16_t* o_ = dnb + r * DNB + o4;
            *reinterpret_cast<uint2*>(o_) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(o_ + TE * DNB) = make_uint2(m0, m1);
#if SVNET_DN_PIECES >= 3
            *reinterpret_cast<uint2*>(o_ + 2 * TE * DNB) = make_uint2(l0, l1);
#endif
        }
    }
    __syncthreads();
    PHASE_MARK(0);   // phase A
    if (MODE == 2) { PHASE_FLUSH(); return; }

    // ---- ternary planes of this tile -> row-sliced 32-bit halves (rows = the tile's 32 edges)
    {
        const int64_t word_row = e0 >> 6;
        const int half = (int)((e0 >> 5) & 1);
        if (Cs <= 32 && 2 * Cv <= 32) {
            // narrow layer: every word has at most 32 columns in use, so lanes 0..31 carry the low half of one (plane, word)
            // entry and lanes 32..63 the low half of the next one: five rounds of 32 ballots instead of ten
            for (int pi = wave; pi < NW; pi += 4) {
                const int ent = 2 * pi + (lane >> 5);                  // entry = plane * NW + word
                const int plane = ent / NW, w = ent - plane * NW;
                const uint32_t mine = (uint32_t)pl[(plane * TE + (lane & 31)) * NW + w];
                const uint32_t colword = svnet_bit_transpose32(mine, lane);     // lane b of either half: its entry's column b over the 32 rows
                uint32_t* dst = plane == 0 ? d.x_sign32 : d.x_nz32;
                dst[((word_row * NCOL) + w * 64 + (lane & 31)) * 2 + half] = colword;
            }
        } else
        // lanes 0..31 hold the low and lanes 32..63 the high 32 bits of row (lane & 31): one ballot yields two columns
        for (int item = wave; item < 2 * NW; item += 4) {
            const int plane = item / NW, w = item - plane * NW;
            const uint64_t roww = pl[(plane * TE + (lane & 31)) * NW + w];
            const uint32_t mine = lane < 32 ? (uint32_t)roww : (uint32_t)(roww >> 32);
            const uint32_t colword = svnet_bit_transpose32(mine, lane);         // (32 ballots + selects per item before)
            uint32_t* dst = plane == 0 ? d.x_sign32 : d.x_nz32;
            dst[((word_row * NCOL) + w * 64 + lane) * 2 + half] = colword;
        }
    }

    PHASE_MARK(1);   // plane transposition
    // phase C's global operands (NC2 > 0): requested between phase B's MFMAs and its epilogue (see there), consumed in phase C
    constexpr int PC_NCH = NC2 > 0 ? NC2 / 4 : 1;
    float pc_vst[8][2], pc_vi[PC_NCH], pc_zj[3], pc_zi[6];
    // ================= phase B: dx_b = dnl . sign(W1), masked by the STE plane =================
    {
        const int r = lane & 31, h = lane >> 5;
        f32x16 acc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
#pragma unroll
        for (int ks0 = 0; ks0 < NKS; ks0 += NKB) {
            if (!HOIST_B) SVNET_LOAD_BFR(ks0);       // Os = 128: two groups of four k-steps (96 fragment registers would cost a wave per SIMD)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const int ks = ks0 + kb;
                if (ks < nks) {
                    const uint16_t* a_ = dnb + r * DNB + ks * 16 + 8 * h;                          // 16-byte aligned: DNB % 8 == 0
                    const bf16x8 fh = *reinterpret_cast<const bf16x8*>(a_), fm = *reinterpret_cast<const bf16x8*>(a_ + TE * DNB);
#if SVNET_DN_PIECES >= 3
                    const bf16x8 fl = *reinterpret_cast<const bf16x8*>(a_ + 2 * TE * DNB);
#endif
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        if (cts[q] >= 0) {  // wave-uniform
                            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bfr[kb][q], acc[q], 0, 0, 0);
                            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fm, bfr[kb][q], acc[q], 0, 0, 0);
#if SVNET_DN_PIECES >= 3
                            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, bfr[kb][q], acc[q], 0, 0, 0);
#endif
                        }
                    }
                }
            }
        }
        // Phase C's operands are requested HERE for the wide layers, not at the start of phase C: its first two passes are ~100
        // instructions, too short to cover an L2 round trip, while the epilogue below is 300 - 450 (and phase B's weight fragments are
        // dead by now: the registers are free).  conv4 531 -> 521 us, conv3 330 -> 320; at Os = 32 the epilogue is two column tiles per
        // wave and the early requests only lengthened live ranges (254 -> 276 us): there they stay at the start of phase C.
        PHASE_MARK(5);   // phase B: MFMAs issued
        if constexpr (NC2 > 0 && NKS >= 4) SVNET_PHASEC_REQUESTS();
        __syncthreads();   // every wave has consumed dnb: dxl may now overwrite the same LDS bytes
        PHASE_MARK(6);   // phase B: barrier behind the MFMAs
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int ct = cts[q];
            if (ct >= 0) {
                const int col = ct * 32 + r;
                const int wd = ct >> 1, wi = (ct & 1) * 32 + r;                       // word, index inside the word
                const bool in_use = wi < (wd < 2 ? Cs : 2 * Cv);
                const int ccol = (wd < 2 ? wd * Cs : 2 * Cs + (wd - 2) * 2 * Cv) + wi;   // compact LDS column
                // STE bits of this lane's column over the tile's 32 rows: the (rows x 32 columns) bit block of the column tile, transposed
                // across the lanes (35 instructions instead of an LDS word read + 64-bit shift + test for each of the 16 elements)
                const uint32_t roww = reinterpret_cast<const uint32_t*>(pl)[((2 * TE + r) * NW + wd) * 2 + (ct & 1)];
                const int cmask = (int)(svnet_bit_transpose32(roww, lane) >> (4 * h));
                float csum = 0.f;                       // dL/dbeta of this column: sum over the tile's rows (rows past E are zero)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rb = (i & 3) + 8 * (i >> 2), row = rb + 4 * h;
                    const uint32_t keep = (uint32_t)__builtin_amdgcn_sbfe(cmask, rb, 1);            // 0 or ~0
                    const float v = __uint_as_float(__float_as_uint(acc[q][i]) & keep);
                    if (in_use) dxl[row * DXS + ccol] = v;
                    csum += v;
                }
                const float other = __uint_as_float(lane_half_swap(__float_as_uint(csum)));
                if (h == 0 && in_use) { const float t = csum + other; if (t != 0.f) ATOMIC_ADD(&d.dbeta_perm[(tile_lin & (SVNET_DBETA_SLICES - 1)) * NCOL + col], t); }
            }
        }
    }
    __syncthreads();
    PHASE_MARK(2);   // phase B
    if (MODE == 3) { PHASE_FLUSH(); return; }

    if constexpr (NC2 > 0) {
        // ================= phase C, 8 rows per wave, every global access a whole row =================
        // The one-edge-per-iteration form below spends ~160 vector + ~150 scalar instructions per EDGE on 42..64 lanes (address
        // arithmetic, cursor, nine wave-wide sums, a request ring) and is a latency chain of 8 iterations per wave.  Here every wave
        // owns rows 8w .. 8w+7 of the tile and runs four short passes over them:
        //   1. scalar part, lanes = scalar channels: message s-part stored row by row, centre sums per point;
        //   2. the neighbours' v rows, requested at the very start with one coalesced load per row, are parked in the rows' scalar
        //      columns of the dx tile (consumed by pass 1);
        //   3. Vector2Scalar backward with lane = (edge el < 8, channel half ch < 2, axis q < 3 of 4): the lane owns channel pairs
        //      c = ch*NCH .. of axis q of its edge - g_q[c2] = dx[e][2Cs + q*2Cv + c2] (its OWN group of the row), ve_q, z[q][0..2];
        //      the other two groups' g come from its quad through quad_perm DPP reads:
        //        dve[e][q][c2] = sum_jz g_jz[c2] * z[q][jz]       dz[e][q][jz] = sum_c2 g_jz[c2] * ve_q[c2]   (in-lane sums + one row_shr:4)
        //      ~20 vector instructions per channel pair for EIGHT edges; results go back into the consumed columns of the tile;
        //   4. lanes = elements of the message row: the Vector2Scalar part of the message stored row by row, centre sums per point.
        constexpr int NCH = NC2 / 4;                       // channel pairs per lane: Cv <= 2 * NCH
        constexpr int RW = 8;                              // rows per wave
        const int64_t R = msg_stride(Cs, Cv, d.Ov);
        const uint32_t Nu = (uint32_t)d.N, uCv = (uint32_t)Cv;
        const int row0 = RW * wave;
        const int n3 = 3 * Cv;
        float* zs = reinterpret_cast<float*>(pl);          // [TE][9] dL/dz of every row (the plane words are consumed: phase B is over)
        EdgeCursor cur;                                    // position of row0 (uniform)
        cursor_init(d, tp, e0, row0, cur);

        // ---- the neighbour rows v_j (lane = element of the [3][Cv] row) and the per-lane operands of pass 3: requested before phase
        // B's epilogue (wide layers) or here
        if constexpr (NKS < 4) SVNET_PHASEC_REQUESTS();
        float (&vst)[RW][2] = pc_vst;
        // ---- per-lane operands of pass 3 (requested there as well)
        const int el = lane >> 3, ch = (lane >> 2) & 1, q = lane & 3, qq = min(q, 2);
        const int r = row0 + el;
        const int64_t e = e0 + r;
        const bool in_range = e < E;
        const int jloc = __builtin_amdgcn_ds_bpermute(4 * el, jv8);
        const bool valid = in_range && (uint32_t)jloc < Nu;
        if (in_range && !valid && d.debug && (lane & 7) == 0) {
            if (atomicAdd(reinterpret_cast<unsigned long long*>(d.debug), 1ull) == 0ull) { d.debug[1] = e; d.debug[2] = jloc; d.debug[3] = d.N; }
        }
        const int c0 = ch * NCH;                           // first channel pair of this lane
        float (&vi)[NCH] = pc_vi;
        const float zj0 = pc_zj[0], zj1 = pc_zj[1], zj2 = pc_zj[2];
        const float zi0 = pc_zi[0], zi1 = pc_zi[1], zi2 = pc_zi[2], zi3 = pc_zi[3], zi4 = pc_zi[4], zi5 = pc_zi[5];

        PHASE_MARK(10);   // phase C: set-up (cursor, requests of the narrow layers)
        // ---- pass 1: scalar part, lanes = scalar channels
        {
            const bool s_lane = lane < Cs;
            const int sl = min(lane, Cs - 1);
            uint32_t cur_p = 0xFFFFFFFFu, cur_b = 0xFFFFFFFFu;
            float g0c = 0.f, g1c = 0.f, cs_sum = 0.f;
            EdgeCursor c2 = cur;
            for (int rr = 0; rr < RW; ++rr) {
                if (c2.e >= E) break;                              // (uniform)
                if (c2.gp != cur_p) {
                    if (cur_p != 0xFFFFFFFFu && s_lane) ATOMIC_ADD(&d.ds_acc[cur_p * (uint32_t)Cs + (uint32_t)lane], cs_sum);
                    cur_p = c2.gp;
                    cs_sum = 0.f;
                    if (c2.b != cur_b) {
                        cur_b = c2.b;
                        if (cur_b == tp.b0) { g0c = gc_pre0; g1c = gc_pre1; }      // (the tile's first cloud: requested at the kernel's start)
                        else { g0c = d.gconst[cur_b * 2u * (uint32_t)Cs + sl]; g1c = d.gconst[cur_b * 2u * (uint32_t)Cs + Cs + sl]; }
                    }
                }
                const float* row = dxl + (row0 + rr) * DXS;
                const float d0 = row[sl] + g0c;
                if (s_lane) {
                    st_f32_sbase(d.msg + c2.e * R, 4u * (uint32_t)lane, d0);
                    cs_sum += (row[Cs + sl] + g1c) - d0;
                }
                cursor_next(d, c2);
            }
            if (cur_p != 0xFFFFFFFFu && s_lane) ATOMIC_ADD(&d.ds_acc[cur_p * (uint32_t)Cs + (uint32_t)lane], cs_sum);
        }
        PHASE_MARK(7);   // phase C pass 1
        // ---- pass 2: the neighbours' v rows into the (consumed) scalar columns of their tile rows: [3][Cv] at column 0 (3 Cv <= 2 Cs)
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            float* row = dxl + (row0 + rr) * DXS;
            if (lane < n3) row[lane] = vst[rr][0];
            if (NC2 > 44 && lane + 64 < n3) row[lane + 64] = vst[rr][1];       // (3 Cv > 64: Cv = 22 .. 24 only)
        }
        // ---- pass 3: Vector2Scalar backward
        {
            const float z0 = zj0 + (zi3 - zi0), z1 = zj1 + (zi4 - zi1), z2 = zj2 + (zi5 - zi2);    // z[q][jz] = Zp_j - Zp_i + Zq_i
            float* rowp = dxl + r * DXS;
            float* gmine = rowp + 2 * Cs + qq * 2 * Cv;            // this lane's group of the row's Vector2Scalar columns
            const float* vjl = rowp + qq * Cv;                     // axis q of the parked neighbour row
            float dz0 = 0.f, dz1 = 0.f, dz2 = 0.f;
            constexpr int QB0 = 0x00, QB1 = 0x55, QB2 = 0xAA;      // quad_perm broadcasts of lane 0 / 1 / 2 of the quad
            // the message part dve[c] (difference channel) replaces the consumed column c, the centre contribution
            // dve[Cv + c] - dve[c] the consumed column Cv + c (only this lane reads them; its quad gets them through DPP)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = c0 + i;
                const bool con = c < Cv;
                const int cc = min(c, Cv - 1);
                const float gd = (con && valid) ? gmine[cc] : 0.f, gc = (con && valid) ? gmine[Cv + cc] : 0.f;
                const float gd0 = dpp_get<QB0>(gd), gd1 = dpp_get<QB1>(gd), gd2 = dpp_get<QB2>(gd);
                const float gc0 = dpp_get<QB0>(gc), gc1 = dpp_get<QB1>(gc), gc2 = dpp_get<QB2>(gc);
                const float ved = vjl[cc] - vi[i], vec = vi[i];
                const float dved = gd0 * z0 + gd1 * z1 + gd2 * z2;                          // dL/dve of the difference channel -> the neighbour
                const float dvc = (gc0 - gd0) * z0 + (gc1 - gd1) * z1 + (gc2 - gd2) * z2;   // centre: dve[Cv + c] - dve[c]
                dz0 += gd0 * ved + gc0 * vec; dz1 += gd1 * ved + gc1 * vec; dz2 += gd2 * ved + gc2 * vec;
                if (q < 3 && con) { gmine[c] = dved; gmine[Cv + c] = dvc; }
            }
            constexpr int DPP_ROW_SHR4 = 0x114;                    // the two channel halves of an edge sit 4 lanes apart: the upper one collects
            dz0 += dpp_get<DPP_ROW_SHR4>(dz0); dz1 += dpp_get<DPP_ROW_SHR4>(dz1); dz2 += dpp_get<DPP_ROW_SHR4>(dz2);
            if (ch == 1 && q < 3) { zs[r * 9 + q * 3 + 0] = dz0; zs[r * 9 + q * 3 + 1] = dz1; zs[r * 9 + q * 3 + 2] = dz2; }
        }
        PHASE_MARK(9);   // phase C passes 2 + 3
        // ---- pass 4: lanes = elements L of the message row's Vector2Scalar part [dve (3 x Cv) | dz (9)]; centre sums [dv | dz] per point
        {
            const int nout = n3 + 9;
            for (int L0 = 0; L0 < nout; L0 += 64) {
                const int L = L0 + lane;
                const bool on = L < nout;
                const int Ld = on ? L : 0;
                const bool isv = Ld < n3;
                const int grp = isv ? Ld / Cv : 0;
                const int colm = 2 * Cs + grp * 2 * Cv + (Ld - grp * Cv);     // column of dve element (grp, c); its centre twin is Cv further
                EdgeCursor c2 = cur;
                uint32_t cur_p = 0xFFFFFFFFu;
                float acc = 0.f;
#define SVNET_FLUSH_V(p)                                                                                           \
    do {                                                                                                           \
        if (on) {                                                                                                  \
            if (isv) ATOMIC_ADD(&d.dv_acc[(p) * 3u * uCv + (uint32_t)L], acc);                                     \
            else ATOMIC_ADD(&d.dzc[(p) * 9u + (uint32_t)(L - n3)], acc);                                           \
        }                                                                                                          \
    } while (0)
                for (int rr = 0; rr < RW; ++rr) {
                    if (c2.e >= E) break;                          // (uniform)
                    if (c2.gp != cur_p) {
                        if (cur_p != 0xFFFFFFFFu) SVNET_FLUSH_V(cur_p);
                        cur_p = c2.gp;
                        acc = 0.f;
                    }
                    const float* row = dxl + (row0 + rr) * DXS;
                    const float mv_ = isv ? row[colm] : zs[(row0 + rr) * 9 + (Ld - n3)];
                    acc += isv ? row[colm + Cv] : mv_;
                    if (on) st_f32_sbase(d.msg + c2.e * R + Cs, 4u * (uint32_t)L, mv_);
                    cursor_next(d, c2);
                }
                if (cur_p != 0xFFFFFFFFu) SVNET_FLUSH_V(cur_p);
#undef SVNET_FLUSH_V
            }
        }
        PHASE_MARK(4);
        PHASE_FLUSH();
        return;
    }

    // (the s part - msg[e][c] = dx[e][c] + gconst0[c], ds_acc[i,c] += sum over the point's edges of dx[e][Cs+c] + gconst1[c] - msg[e][c] -
    //  used to be a phase of its own over the whole tile (three barriers, LDS atomics): it now rides in phase C2's edge loop,
    //  lanes = scalar channels, centre sums in a register per point)
    const int64_t R = msg_stride(Cs, Cv, d.Ov);
    PHASE_MARK(3);   // phase C1
    // ================= phase C2 (lanes = vector channels, one edge per wave iteration): v2s backward =================
    {
        uint32_t cur_p = 0xFFFFFFFFu;               // "no point yet"
        uint32_t loaded_p = 0xFFFFFFFFu;            // point whose PointIn is pending
        PointIn pend;
        float vi0 = 0.f, vi1 = 0.f, vi2 = 0.f, zc[9];   // the current point's v (masked per lane class below) and Zq_i - Zp_i
#pragma unroll
        for (int q = 0; q < 9; ++q) zc[q] = 0.f;
        const uint32_t uCv = (uint32_t)Cv;
        float* const mrow0 = d.msg + ew * R + Cs;   // message row of this wave's first edge (the only 64-bit product)
        float cvd0 = 0.f, cvd1 = 0.f, cvd2 = 0.f;   // centre part of dv (diff lanes carry -sum, centre lanes +sum)
        const bool s_lane = lane < Cs;
        const int sl = min(lane, Cs - 1);
        float cs_sum = 0.f;                          // centre part of ds of the current point (lanes = scalar channels)
        EdgeCursor cur;
        cursor_init(d, tp, e0, wave * (TE / 4), cur);
        uint32_t cur_b = min(cur.b, (uint32_t)(d.B - 1));   // cloud whose gate constants are loaded
        float g0c = d.gconst[cur_b * 2u * (uint32_t)Cs + sl], g1c = d.gconst[cur_b * 2u * (uint32_t)Cs + Cs + sl];
        float* const srow0 = d.msg + ew * R;
        float czq = 0.f, czq8 = 0.f;                // centre sums of dL/dz: packed (group g of 8 lanes: entry bitreverse3(g)), entry 8
        const int zq_idx = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2);   // bitreverse3(lane >> 3)
        const bool zq_writer = (lane & 7) == 0;
        const uint32_t bo_d0 = 4u * (uint32_t)lane, bo_d1 = 4u * (uint32_t)(Cv + lane), bo_d2 = 4u * (uint32_t)(2 * Cv + lane),
                       bo_zq = 4u * (uint32_t)(3 * Cv + zq_idx), bo_z8 = 4u * (uint32_t)(3 * Cv + 8);   // byte offsets inside a message row

// (a macro: a by-reference lambda forces the per-point accumulators into scratch memory)
#define SVNET_FLUSH_POINT(p)                                                                                      \
    do {                                                                                                          \
        if (v2_lane) {                                                                                            \
            float* a_ = d.dv_acc + ((p) * 3u * uCv + (uint32_t)cm);                                                \
            ATOMIC_ADD(a_, cvd0);                                                                                  \
            ATOMIC_ADD(a_ + uCv, cvd1);                                                                            \
            ATOMIC_ADD(a_ + 2u * uCv, cvd2);                                                                       \
        }                                                                                                         \
        if (s_lane) ATOMIC_ADD(&d.ds_acc[(p) * (uint32_t)Cs + (uint32_t)lane], cs_sum);                            \
        if (zq_writer) ATOMIC_ADD(&d.dzc[(p) * 9u + (uint32_t)zq_idx], czq);                                       \
        if (lane == 63) ATOMIC_ADD(&d.dzc[(p) * 9u + 8u], czq8);                                                   \
    } while (0)

        // edge rows are requested two iterations ahead (a ring of four with compile-time slots: no register copies)
        EdgeIn q[4];
        load_edge(d, cur, __builtin_amdgcn_readlane(jv8, 0), E, lane, v2_lane, cm, q[0], pend, loaded_p);
        if (loaded_p != 0xFFFFFFFFu && cur.t + 1 == (int)d.k) {
            // the wave's second edge already starts another point: take the first point's operands now (this waits for them)
            cur_p = loaded_p;
            vi0 = pend.vi0; vi1 = pend.vi1; vi2 = pend.vi2;
#pragma unroll
            for (int q9 = 0; q9 < 9; ++q9) zc[q9] = pend.zi[(q9 / 3) * 6 + 3 + q9 % 3] - pend.zi[(q9 / 3) * 6 + q9 % 3];
        }
        cursor_next(d, cur);
        load_edge(d, cur, __builtin_amdgcn_readlane(jv8, 1), E, lane, v2_lane, cm, q[1], pend, loaded_p);
#pragma unroll 1
        for (int r4 = 0; r4 < TE / 4; r4 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r4 + u;
            const int r = wave * (TE / 4) + rr;
            const EdgeIn& in = q[u];
            // the current edge's point operands are taken BEFORE the requests of this iteration go out (they may re-use `pend`)
            if (in.in_range && in.gp != cur_p) {
                if (cur_p != 0xFFFFFFFFu) SVNET_FLUSH_POINT(cur_p);
                cur_p = in.gp;
                cvd0 = cvd1 = cvd2 = 0.f;
                czq = czq8 = 0.f;
                cs_sum = 0.f;
                if (in.b != cur_b) {   // the tile straddles two clouds: the other cloud's gate-path constants
                    cur_b = in.b;
                    g0c = d.gconst[cur_b * 2u * (uint32_t)Cs + sl]; g1c = d.gconst[cur_b * 2u * (uint32_t)Cs + Cs + sl];
                }
                vi0 = pend.vi0; vi1 = pend.vi1; vi2 = pend.vi2;
#pragma unroll
                for (int q9 = 0; q9 < 9; ++q9) zc[q9] = pend.zi[(q9 / 3) * 6 + 3 + q9 % 3] - pend.zi[(q9 / 3) * 6 + q9 % 3];
            }
            // (the last two requests of the wave repeat its last row: nothing consumes them)
            cursor_next(d, cur);
            load_edge(d, cur, __builtin_amdgcn_readlane(jv8, min(rr + 2, TE / 4 - 1)), rr + 2 < TE / 4 ? E : 0, lane, v2_lane, cm,
                      q[(u + 2) & 3], pend, loaded_p);
            const float* row = dxl + r * DXS;
            if (in.in_range) {   // ---- s part (wave-uniform branch; no load-destination registers inside)
                const float d0 = row[sl] + g0c;
                if (s_lane) {
                    st_f32_sbase(srow0 + (uint32_t)rr * (uint32_t)R, 4u * (uint32_t)lane, d0);
                    cs_sum += (row[Cs + sl] + g1c) - d0;
                }
            }
            const bool live = v2_lane && in.valid;   // rows past E / dropped edges contribute zeros and store nothing
            const float g0 = live ? row[2 * Cs + lane] : 0.f, g1 = live ? row[2 * Cs + 2 * Cv + lane] : 0.f,
                        g2 = live ? row[2 * Cs + 4 * Cv + lane] : 0.f;
            const float z0 = in.zj[0] + zc[0], z1 = in.zj[1] + zc[1], z2 = in.zj[2] + zc[2], z3 = in.zj[3] + zc[3], z4 = in.zj[4] + zc[4],
                        z5 = in.zj[5] + zc[5], z6 = in.zj[6] + zc[6], z7 = in.zj[7] + zc[7], z8 = in.zj[8] + zc[8];
            // v2s backward: s_v[c2][jz] = sum_d ve[d][c2] * z[d][jz]
            const float ve0 = diff_lane ? (in.vj0 - vi0) : (v2_lane ? vi0 : 0.f);
            const float ve1 = diff_lane ? (in.vj1 - vi1) : (v2_lane ? vi1 : 0.f);
            const float ve2 = diff_lane ? (in.vj2 - vi2) : (v2_lane ? vi2 : 0.f);
            const float dve0 = g0 * z0 + g1 * z1 + g2 * z2;
            const float dve1 = g0 * z3 + g1 * z4 + g2 * z5;
            const float dve2 = g0 * z6 + g1 * z7 + g2 * z8;
            // dL/dz[d][jz] = sum over the lanes of g_jz * ve_d: nine wave-wide sums, eight of them packed
            const float pz[8] = {g0 * ve0, g1 * ve0, g2 * ve0, g0 * ve1, g1 * ve1, g2 * ve1, g0 * ve2, g1 * ve2};
            const float dzp = wave_sum8_packed(pz, lane);
            const float dz8 = wave_sum_last(g2 * ve2);
            czq += dzp;
            czq8 += dz8;
            if (diff_lane) { cvd0 -= dve0; cvd1 -= dve1; cvd2 -= dve2; }
            else if (v2_lane) { cvd0 += dve0; cvd1 += dve1; cvd2 += dve2; }
            // ---- the neighbour's share: plain stores into the edge's message row
            if (in.valid) {
                float* m = mrow0 + (uint32_t)rr * (uint32_t)R;   // wave-uniform: SGPR-base stores with 32-bit lane byte offsets
                if (diff_lane) { st_f32_sbase(m, bo_d0, dve0); st_f32_sbase(m, bo_d1, dve1); st_f32_sbase(m, bo_d2, dve2); }
                if (zq_writer) st_f32_sbase(m, bo_zq, dzp);
                if (lane == 63) st_f32_sbase(m, bo_z8, dz8);
            }
        }
        }
        if (cur_p != 0xFFFFFFFFu) SVNET_FLUSH_POINT(cur_p);
#undef SVNET_FLUSH_POINT
#undef SVNET_LOAD_BFR
    }
    PHASE_MARK(4);   // phase C2 (wave 0 of the workgroup)
    PHASE_FLUSH();
}

}  // namespace

extern "C" int svnet_edgeblock_wbt_bf16(const uint64_t* w_sign, const uint64_t* w_nz, int64_t Os, int64_t Cs, int64_t Cv, uint16_t* wbt,
                                        uint32_t* w_dense, void* stream) {
    SVNET_REQUIRE(w_sign && w_nz && wbt && Os > 0 && Os % 8 == 0, SVNET_E_ARG, "svnet_edgeblock_wbt_bf16: bad arguments (Os must be a multiple of 8)");
    SVNET_REQUIRE(!w_dense || (Cs > 0 && Cs <= 64 && Cv > 0 && 2 * Cv <= 64), SVNET_E_ARG, "svnet_edgeblock_wbt_bf16: w_dense needs the layer's Cs <= 64, 2 Cv <= 64");
    hipLaunchKernelGGL(edgeblock_wbt_kernel, dim3((unsigned)svnet_cdiv(NCOL * ((Os + 15) / 16 * 16), 256)), dim3(256), 0, (hipStream_t)stream, w_sign, w_nz,
                       (int)Os, (int)Cs, (int)Cv, wbt, w_dense);
    SVNET_CHECK_LAUNCH("edgeblock_wbt_kernel");
    return SVNET_OK;
}

extern "C" int svnet_edgeblock_bwd_prelude_f32(const float* gs, const float* gv, const int32_t* n_max, const int32_t* n_min,
                                               const float* mv, const float* mvn, const float* coef, const float* scale1,
                                               const float* gate, int64_t P, int64_t N, int64_t Os, int64_t Ov, float slope,
                                               float* gy, float* red, float* redv, float* dgate, const float* gs2, int64_t gs2_ld,
                                               const float* gv2, int64_t gv2_ld, float* gv_sum, void* stream) {
    SVNET_REQUIRE((gs || gs2) && (gv || gv2) && n_max && n_min && mv && mvn && coef && scale1 && gate && gy && red && redv && dgate, SVNET_E_ARG,
                  "svnet_edgeblock_bwd_prelude_f32: null pointer");
    SVNET_REQUIRE((!gs2 || gs2_ld >= Os) && (!gv2 || (gv2_ld >= Ov && gv_sum)), SVNET_E_ARG,
                  "svnet_edgeblock_bwd_prelude_f32: second gradient source needs row strides >= the slice and gv_sum");
    SVNET_REQUIRE(P > 0 && N > 0 && Os > 0 && Ov > 0, SVNET_E_ARG, "svnet_edgeblock_bwd_prelude_f32: bad sizes");
    SVNET_REQUIRE(Os <= 128 && Ov <= 64, SVNET_E_UNSUPPORTED, "svnet_edgeblock_bwd_prelude_f32: Os <= 128, Ov <= 64");
    const int64_t rpb = svnet_prelude_rows(N);
    hipLaunchKernelGGL(edgeblock_bwd_prelude_kernel, dim3((unsigned)svnet_cdiv(P, rpb)), dim3(256), 0, (hipStream_t)stream, gs, gv, n_max,
                       n_min, mv, mvn, coef, scale1, gate, P, N, (int)Os, (int)Ov, slope, rpb, gy, red, redv, dgate, gs2, gs2_ld, gv2, gv2_ld,
                       gv_sum);
    SVNET_CHECK_LAUNCH("edgeblock_bwd_prelude_kernel");
    return SVNET_OK;
}

extern "C" int svnet_edgeblock_bwd_coeffs_f32(const float* red, const float* redv, const float* coef, const float* gamma1,
                                              const float* gamma2, int64_t E, int64_t Os, int64_t Ov, int training, const float* scale1,
                                              float* bcoef, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2,
                                              const svnet_gate_bwd_job* gate_job, void* stream) {
    SVNET_REQUIRE(red && redv && coef && gamma1 && gamma2 && bcoef && dgamma1 && dbeta1 && dgamma2 && dbeta2 && E > 0, SVNET_E_ARG,
                  "svnet_edgeblock_bwd_coeffs_f32: bad arguments");
    const int64_t n = Os > Ov ? Os : Ov;
    SVNET_REQUIRE(!gate_job || svnet_gate_bwd_job_ok(gate_job), SVNET_E_ARG, "svnet_edgeblock_bwd_coeffs_f32: bad gate job");
    const int coef_blocks = (int)svnet_cdiv(n, 256);
    const svnet_gate_bwd_job job = gate_job ? *gate_job : svnet_gate_bwd_job{};
    const int chunks = gate_job ? svnet_gate_bwd_chunks(job.Cin, job.H, job.Ov) : 1;
    hipLaunchKernelGGL(edgeblock_bwd_coeffs_kernel, dim3((unsigned)(coef_blocks + (gate_job ? gate_job->B * chunks : 0))), dim3(256), 0,
                       (hipStream_t)stream, red, redv, coef, gamma1, gamma2, E, (int)Os, (int)Ov, training, scale1, bcoef, dgamma1, dbeta1,
                       dgamma2, dbeta2, job, coef_blocks, chunks);
    SVNET_CHECK_LAUNCH("edgeblock_bwd_coeffs_kernel");
    return SVNET_OK;
}

extern "C" int svnet_edgeblock_bwd_f32(const svnet_edgeblock_bwd_desc* desc, void* stream) {
    SVNET_REQUIRE(desc, SVNET_E_ARG, "svnet_edgeblock_bwd_f32: null descriptor");
    const svnet_edgeblock_bwd_desc& d = *desc;
    SVNET_REQUIRE(d.v && d.idx && d.zz && d.ut && d.n16 && d.planes && d.w1bt && d.scale1 && d.slot_max && d.slot_min && d.coef &&
                      d.gate && d.gy && d.bcoef && d.gv && d.gconst && d.x_sign32 && d.x_nz32 && d.ds_acc && d.dv_acc &&
                      d.msg && d.dvc && d.dzc && d.dbeta_perm && d.ub_tab && d.ge_tab,
                  SVNET_E_ARG, "svnet_edgeblock_bwd_f32: null pointer");
    SVNET_REQUIRE(d.B >= 0 && d.N > 0 && d.k >= 2 && d.k <= 64, SVNET_E_ARG, "svnet_edgeblock_bwd_f32: bad sizes (2 <= k <= 64)");
    SVNET_REQUIRE((d.Os & (d.Os - 1)) == 0, SVNET_E_UNSUPPORTED, "svnet_edgeblock_bwd_f32: Os must be a power of two (8..128)");
    SVNET_REQUIRE(d.B * d.N * 384 < ((int64_t)1 << 32), SVNET_E_UNSUPPORTED, "svnet_edgeblock_bwd_f32: more than 11 M points (32-bit row offsets)");
    SVNET_REQUIRE(d.B * d.N * d.k < ((int64_t)1 << 31), SVNET_E_UNSUPPORTED, "svnet_edgeblock_bwd_f32: more than 2^31 edge rows");
    SVNET_REQUIRE(d.Cs > 0 && d.Cs <= 64 && d.Cv > 0 && 2 * d.Cv <= 64 && d.Os > 0 && d.Os <= 128 && d.Os % 8 == 0 && d.Ov > 0 &&
                      d.Ov <= 64, SVNET_E_UNSUPPORTED, "svnet_edgeblock_bwd_f32: channel counts outside Cs<=64, 2Cv<=64, Os<=128 (mult of 8), Ov<=64");
    const int64_t E = d.B * d.N * d.k;
    if (E == 0) return SVNET_OK;
    hipStream_t st = (hipStream_t)stream;
    static const int mode = getenv("SVNET_BWD_MODE") ? atoi(getenv("SVNET_BWD_MODE")) : 0;

    // vector path: wave per point
    VecArgs va;
    va.d = d;
    int wpc = (int)svnet_cdiv(4096, d.B);                 // ~4096 waves in flight (16 per CU)
    if (wpc > d.N) wpc = (int)d.N;
    if (wpc < 1) wpc = 1;
    va.points_per_wave = (int)svnet_cdiv(d.N, wpc);
    va.waves_per_cloud = (int)svnet_cdiv(d.N, va.points_per_wave);
    const unsigned vgrid = (unsigned)svnet_cdiv(d.B * va.waves_per_cloud, 4);
    const bool do_vec = d.parts == 0 || (d.parts & 1), do_tile = d.parts == 0 || (d.parts & 2);
    if (do_vec) {
        if (mode == 1) hipLaunchKernelGGL((edgeblock_bwd_vec_kernel<1>), dim3(vgrid), dim3(256), 0, st, va);
        else hipLaunchKernelGGL((edgeblock_bwd_vec_kernel<0>), dim3(vgrid), dim3(256), 0, st, va);
        SVNET_CHECK_LAUNCH("edgeblock_bwd_vec_kernel");
    }
    if (!do_tile) return SVNET_OK;

    // scalar path: 32-edge tiles
    const int dxs = dx_stride(d.Cs, d.Cv), dnf = (3 * TE * dn_stride(d.Os) + 1) / 2;
    const int dx_floats = TE * dxs > dnf ? TE * dxs : dnf;              // the dL/dn planes alias dxl
    const size_t lds = (size_t)((dx_floats + 3) & ~3) * 4 + (size_t)3 * TE * NW * 8;
    const unsigned grid = (unsigned)svnet_cdiv(E, TE);
#define SVNET_LAUNCH_BWD(MODE)                                                                                              \
    do {                                                                                                                    \
        if (d.Os <= 32) hipLaunchKernelGGL((edgeblock_bwd_kernel<MODE, 2>), dim3(grid), dim3(256), lds, st, d);             \
        else if (d.Os <= 64) hipLaunchKernelGGL((edgeblock_bwd_kernel<MODE, 4>), dim3(grid), dim3(256), lds, st, d);        \
        else hipLaunchKernelGGL((edgeblock_bwd_kernel<MODE, 8>), dim3(grid), dim3(256), lds, st, d);                        \
    } while (0)
#define SVNET_LAUNCH_BWD_C(NKS_)                                                                                            \
    do {                                                                                                                    \
        if (d.Cv <= 10) hipLaunchKernelGGL((edgeblock_bwd_kernel<0, NKS_, 20>), dim3(grid), dim3(256), lds, st, d);         \
        else if (d.Cv <= 12) hipLaunchKernelGGL((edgeblock_bwd_kernel<0, NKS_, 24>), dim3(grid), dim3(256), lds, st, d);    \
        else if (d.Cv <= 21) hipLaunchKernelGGL((edgeblock_bwd_kernel<0, NKS_, 44>), dim3(grid), dim3(256), lds, st, d);    \
        else hipLaunchKernelGGL((edgeblock_bwd_kernel<0, NKS_, 48>), dim3(grid), dim3(256), lds, st, d);                    \
    } while (0)
    static const bool c_old = getenv("SVNET_BWD_C_OLD") != nullptr;      // (diagnostic: the one-edge-per-iteration phase C)
    if (mode == 0 && !c_old && d.Cv >= 3 && d.Cv <= 24 && 3 * d.Cv <= 2 * d.Cs && 3 * d.Cv <= 128) {     // (else the one-edge form)
        if (d.Os <= 32) SVNET_LAUNCH_BWD_C(2); else if (d.Os <= 64) SVNET_LAUNCH_BWD_C(4); else SVNET_LAUNCH_BWD_C(8);
    } else if (mode == 1) SVNET_LAUNCH_BWD(1); else if (mode == 2) SVNET_LAUNCH_BWD(2); else if (mode == 3) SVNET_LAUNCH_BWD(3); else SVNET_LAUNCH_BWD(0);
#undef SVNET_LAUNCH_BWD
#undef SVNET_LAUNCH_BWD_C
    SVNET_CHECK_LAUNCH("edgeblock_bwd_kernel");
    return SVNET_OK;
}
