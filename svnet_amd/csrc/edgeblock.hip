// Fused edge block (tier 2): k-NN gather -> binarized SVBlock -> neighbour pooling in ONE pass over the edges,
// without ever writing an edge tensor to HBM.
//
// Replaces, for a binarized edge layer (conv2/3/4 of sv_dgcnn_cls.py:55-65), the chain
//   get_graph_feature_sv (sv_util.py:90-116) -> SVBlock.forward (sv_layers.py:172-196) -> svpool (sv_util.py:118-132).
// Algebra that makes the single pass possible (re-derived in DESIGN.md §4.2):
//   * linear maps applied to v_e = [v_j - v_i, v_i] collapse to per-POINT products:
//       z(e)  = Zp[j] - Zp[i] + Zq[i]            (Vector2Scalar frame, 3x3)
//       v'(e) = U[j]  - U[i]  + T[i]             (linear2, 3 x Ov)
//     so the edge loop only gathers rows of small point tables (L2 / Infinity-Cache resident);
//   * pre-BN scalar outputs are scale*n with integer n (ternary popcount), BatchNorm+LeakyReLU is monotone per
//     channel, so max_k commutes with it: only max_k n, min_k n (and their slots) and the exact integer sums
//     sum n, sum n^2 are needed;
//   * VectorBN is affine in (v', v'/|v'|): mean_k out = gate * (Av * mean_k v' + Bv * mean_k v'/n').
// One wave per point; lanes are channels.  The five ternary words of an edge row (s_j-s_i | s_i | s_v[:,0..2]) are
// produced by wave ballots in a lane-friendly bit order; linear1's sign planes are permuted to that order once.
#include <limits.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gate_mlp.h"
#include "apply_knn.h"

namespace {

constexpr float VEPS = 1e-6f;
constexpr int NW = 5;  // ternary words per edge row in fused bit order

__device__ __forceinline__ int tdot(uint64_t xs, uint64_t xz, uint64_t ws, uint64_t wz) {
    const uint64_t m = xz & wz;
    return __popcll(m) - 2 * __popcll(m & (xs ^ ws));
}

// feature index of (word, bit) in the reference's K1 ordering [s_j-s_i (Cs) | s_i (Cs) | s_v (2Cv x 3)], or -1
__device__ __forceinline__ int fused_feature(int w, int b, int Cs, int Cv) {
    if (w == 0) return b < Cs ? b : -1;
    if (w == 1) return b < Cs ? Cs + b : -1;
    return b < 2 * Cv ? 2 * Cs + b * 3 + (w - 2) : -1;
}

// one wave per (output channel, word): lane b loads the weight of bit b, two ballots make the plane words
__global__ __launch_bounds__(256) void edgeblock_prepare_kernel(const float* __restrict__ W, const float* __restrict__ beta, int Os, int Cs,
                                                                int Cv, uint64_t* __restrict__ w_sign, uint64_t* __restrict__ w_nz,
                                                                float* __restrict__ beta_perm) {
    const int K1 = 2 * Cs + 6 * Cv;
    const int lane = threadIdx.x & 63;
    const int item = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);      // (o, w) pairs, then the NW beta words
    if (item < Os * NW) {
        const int o = item / NW, w = item - o * NW;
        const int f = fused_feature(w, lane, Cs, Cv);
        const float v = f >= 0 ? W[(int64_t)o * K1 + f] : 0.f;
        const uint64_t sg = __ballot(v > 0.f), nz = __ballot(v != 0.f);
        if (lane == 0) { w_sign[item] = sg; w_nz[item] = nz; }
    } else if (item < Os * NW + NW) {
        const int w = item - Os * NW;
        const int f = fused_feature(w, lane, Cs, Cv);
        beta_perm[w * 64 + lane] = f >= 0 ? beta[f] : 0.f;
    }
}

struct FwdArgs {
    svnet_edgeblock_desc d;
    int waves_per_cloud, points_per_wave;
};

// OP = scalar outputs per lane (Os <= 64*OP)
__device__ __forceinline__ int tdot(uint32_t xs, uint32_t xz, uint32_t ws, uint32_t wz) {
    const uint32_t m = xz & wz;
    return __popc(m) - 2 * __popc(m & (xs ^ ws));
}
// The same product as two running popcounts, pm += popc(m), pd += popc(m & (xs ^ ws)): v_bcnt_u32_b32 adds into its third operand
// for free, so a row of words costs one "pm - 2*pd" instead of a subtract-and-add per word.
__device__ __forceinline__ void tacc(uint64_t xs, uint64_t xz, uint64_t ws, uint64_t wz, int& pm, int& pd) {
    const uint64_t m = xz & wz;
    pm += __popcll(m);
    pd += __popcll(m & (xs ^ ws));
}
__device__ __forceinline__ void tacc(uint32_t xs, uint32_t xz, uint32_t ws, uint32_t wz, int& pm, int& pd) {
    const uint32_t m = xz & wz;
    pm += __popc(m);
    pd += __popc(m & (xs ^ ws));
}

// DENSE weights (no exact zero in W1: *w_dense, set by svnet_edgeblock_prepare_f32): the mask of a product is the edge's own non-zero plane,
// so popc(m) is ONE wave-uniform count per edge (scalar unit) and a word costs xor + and + bcnt instead of and + bcnt + xor + and + bcnt:
// 40 % fewer instructions in the half of the kernel that is popcounts.  Same integer, bit for bit.
__device__ __forceinline__ void tacc_dense(uint64_t xs, uint64_t xz, uint64_t ws, int& pd) { pd += __popcll(xz & (xs ^ ws)); }
__device__ __forceinline__ void tacc_dense(uint32_t xs, uint32_t xz, uint32_t ws, int& pd) { pd += __popc(xz & (xs ^ ws)); }

__device__ __forceinline__ bool edge_weights_dense(const uint32_t* __restrict__ w_dense) {      // (wave-uniform: a scalar load)
    return w_dense && *w_dense == 1u;
}

// NARROW: every word has at most 32 columns in use (Cs <= 32 and 2 Cv <= 32): the popcount products run on the low halves only
// The kernel proper.  Every table comes in as a __restrict__ parameter (the kernel below just unpacks the descriptor): with
// the aliasing question settled, the wave-uniform reads (the zz rows) become scalar loads instead of vector loads + readlanes.
template <int OP, bool NARROW, bool DENSE>
__device__ __forceinline__ void edgeblock_fwd_body(const FwdArgs& fa, const float* __restrict__ ts, const float* __restrict__ tv,
                                                   const int64_t* __restrict__ tidx, const float* __restrict__ tzz,
                                                   const float* __restrict__ tut, int16_t* __restrict__ o_n16,
                                                   uint64_t* __restrict__ o_planes, int32_t* __restrict__ o_nmax,
                                                   int32_t* __restrict__ o_nmin, uint8_t* __restrict__ o_smax,
                                                   uint8_t* __restrict__ o_smin, float* __restrict__ o_mv, float* __restrict__ o_mvn) {
    typedef typename std::conditional<NARROW, uint32_t, uint64_t>::type word_t;
    const svnet_edgeblock_desc& d = fa.d;
    const int lane = threadIdx.x & 63;
    // XCD-aware order: workgroups b and b+8 share an XCD, so XCD x walks the clouds x, x+8, ... one after the other and
    // a cloud's point tables stay in that XCD's L2 while its edges are processed
    int64_t blk = blockIdx.x;
    if ((d.B & 7) == 0 && (fa.waves_per_cloud & 3) == 0) {
        const int64_t bpc = fa.waves_per_cloud >> 2, xcd = blk & 7, slot = blk >> 3;
        blk = ((slot / bpc) * 8 + xcd) * bpc + (slot % bpc);
    }
    const int64_t wave_g = blk * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // uniform: scalar loads / SGPR addressing
    const int64_t bq = wave_g / fa.waves_per_cloud;
    const int64_t b = bq < d.B ? bq : d.B - 1;                 // idle waves keep valid addresses and reach the barriers
    const int wi = (int)(wave_g - bq * fa.waves_per_cloud);
    const int p_begin = wi * fa.points_per_wave;
    const int p_end = bq < d.B ? min((int)d.N, p_begin + fa.points_per_wave) : p_begin;
    const int Cs = d.Cs, Cv = d.Cv, Os = d.Os, Ov = d.Ov, k = (int)d.k;

    // my output channels' weight words
    word_t wsg[OP][NW], wnz[OP][NW];
#pragma unroll
    for (int op = 0; op < OP; ++op) {
        const int o = lane + 64 * op;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            wsg[op][w] = (word_t)((o < Os) ? d.w_sign[o * NW + w] : 0ull);
            wnz[op][w] = DENSE ? (word_t)~(word_t)0 : (word_t)((o < Os) ? d.w_nz[o * NW + w] : 0ull);
        }
    }
    const float bd = d.beta_perm[lane], bc = d.beta_perm[64 + lane];
    float bv[3];
#pragma unroll
    for (int jz = 0; jz < 3; ++jz) bv[jz] = d.beta_perm[128 + 64 * jz + lane];

    const bool s_lane = lane < Cs;
    const bool v2_lane = lane < 2 * Cv;
    const bool diff_lane = lane < Cv;
    const int cm = diff_lane ? lane : lane - Cv;
    const bool o_lane = lane < Ov;

    const bool save = o_planes != nullptr;   // training: keep n and the ternary / STE planes of every edge for the backward

    long long sn[OP], sn2[OP];
#pragma unroll
    for (int op = 0; op < OP; ++op) sn[op] = sn2[op] = 0;
    double sv1 = 0.0, sv2 = 0.0;
    float gs_diff = 0.f, gs_cen = 0.f;

    // Neighbour ids: lane t of one coalesced load holds idx[p][t] (k <= 64) and v_readlane hands it to the scalar unit, so the
    // gathers of an edge never wait on a load of their own index; the next point's ids are requested a whole point ahead.
    const int lk = min(lane, k - 1);
    int jv_next = (p_begin < p_end) ? reinterpret_cast<const int*>(tidx)[2 * ((b * d.N + p_begin) * k + lk)] : 0;    // (low dword of the int64 id)
    for (int p = p_begin; p < p_end; ++p) {
        const int64_t gp = b * d.N + p;
        const int jv = jv_next;
        if (p + 1 < p_end) jv_next = reinterpret_cast<const int*>(tidx)[2 * ((gp + 1) * k + lk)];
        // the point's OWN operands: every load unconditional (clamped lane index, masked where it is used) and requested before the first
        // use.  With exec-masked lanes each `cond ? load : 0` became a branch around its load, every axis its own basic block ending in
        // `s_waitcnt vmcnt(0)` for the difference it forms at once: four dependent L2 round trips at the start of every point.
        const float s_raw = ts[gp * Cs + min(lane, Cs - 1)];
        float vi_raw[3], zr_raw[3][6], tu_raw[3][2];
        {
            const int cmc = v2_lane ? cm : 0, loc = min(lane, Ov - 1);
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
                vi_raw[dd] = tv[(gp * 3 + dd) * Cv + cmc];
                const float* zrow = tzz + (gp * 3 + dd) * 6;
#pragma unroll
                for (int jz = 0; jz < 6; ++jz) zr_raw[dd][jz] = zrow[jz];
                tu_raw[dd][0] = tut[(gp * 3 + dd) * 2 * Ov + Ov + loc];
                tu_raw[dd][1] = tut[(gp * 3 + dd) * 2 * Ov + loc];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const float s_i = s_lane ? s_raw : 0.f;
        gs_cen += s_i;
        const float tc = s_i + bc;
        const uint64_t csg = __ballot(s_lane && tc > 0.f), cnz = __ballot(s_lane && tc != 0.f);
        const uint64_t cst = save ? __ballot(s_lane && fabsf(tc) <= 1.2f) : 0ull;
        int base[OP];
#pragma unroll
        for (int op = 0; op < OP; ++op) base[op] = tdot((word_t)csg, (word_t)cnz, wsg[op][1], wnz[op][1]);

        float vi[3], zi[3][3], ub[3];
#pragma unroll
        for (int dd = 0; dd < 3; ++dd) {
            vi[dd] = v2_lane ? vi_raw[dd] : 0.f;
#pragma unroll
            for (int jz = 0; jz < 3; ++jz) zi[dd][jz] = zr_raw[dd][3 + jz] - zr_raw[dd][jz];  // Zq_i - Zp_i
            ub[dd] = o_lane ? tu_raw[dd][0] - tu_raw[dd][1] : 0.f;  // T_i - U_i
        }
        int nmax[OP], nmin[OP], smax[OP], smin[OP];
#pragma unroll
        for (int op = 0; op < OP; ++op) { nmax[op] = INT_MIN; nmin[op] = INT_MAX; smax[op] = 0; smin[op] = 0; }
        float av[3] = {0.f, 0.f, 0.f}, avn[3] = {0.f, 0.f, 0.f};

        // neighbour rows are loaded one edge ahead (all lanes, clamped channel index: no exec-masked branches) into two
        // alternating register sets (the loop is unrolled by two, so no set is ever copied).  The zz row is wave-uniform; its
        // address carries an opaque zero so that it stays a plain (broadcast) vector load instead of load + 18 readlanes.
        const int ls = min(lane, Cs - 1), ld = min(lane, Cv - 1), lo = min(lane, Ov - 1);
        const int opaque0 = (int)__builtin_amdgcn_mbcnt_lo(0u, 0u);
        const uint32_t bs0 = 4u * (uint32_t)ls, bd0 = 4u * (uint32_t)ld, bd1 = 4u * (uint32_t)(ld + Cv), bd2 = 4u * (uint32_t)(ld + 2 * Cv),
                       bo0 = 4u * (uint32_t)lo, bo1 = 4u * (uint32_t)(lo + 2 * Ov), bo2 = 4u * (uint32_t)(lo + 4 * Ov);
        struct Nbr { float sj, vj0, vj1, vj2, u0, u1, u2, z[9]; };
        Nbr na = {}, nb = {};
#define SVNET_LOAD_NBR(N_, T)                                                               \
    do {                                                                                    \
        const int64_t gj_ = b * d.N + __builtin_amdgcn_readlane(jv, (T));                   \
        const float* ps_ = ts + gj_ * Cs;        /* wave-uniform row bases + 32-bit lane byte offsets: SGPR-base loads */ \
        const float* pv_ = tv + gj_ * 3 * Cv;                                               \
        const float* pu_ = tut + gj_ * 6 * Ov;                                              \
        N_.sj = ld_f32_sbase(ps_, bs0);                                                     \
        N_.vj0 = ld_f32_sbase(pv_, bd0); N_.vj1 = ld_f32_sbase(pv_, bd1); N_.vj2 = ld_f32_sbase(pv_, bd2); \
        N_.u0 = ld_f32_sbase(pu_, bo0); N_.u1 = ld_f32_sbase(pu_, bo1); N_.u2 = ld_f32_sbase(pu_, bo2); \
        const float* zr_ = tzz + gj_ * 18 + opaque0;                                        \
        N_.z[0] = zr_[0]; N_.z[1] = zr_[1]; N_.z[2] = zr_[2]; N_.z[3] = zr_[6]; N_.z[4] = zr_[7]; N_.z[5] = zr_[8];        \
        N_.z[6] = zr_[12]; N_.z[7] = zr_[13]; N_.z[8] = zr_[14];                            \
    } while (0)
#define SVNET_WL(W, L)                                                                                           \
    do {                                                                                                             \
        asm("v_writelane_b32 %0, %1, " #L : "+v"(vlo) : "s"((int)(uint32_t)(W)));                                    \
        if (!NARROW) asm("v_writelane_b32 %0, %1, " #L : "+v"(vhi) : "s"((int)(uint32_t)((W) >> 32)));               \
    } while (0)
#define SVNET_EDGE(CUR, NXT, T)                                                                                              \
    do {                                                                                                                     \
        const int t = (T);                                                                                                   \
        SVNET_LOAD_NBR(NXT, min(t + 1, k - 1));   /* unconditional (the last edge re-requests itself): a branch here makes */ \
                                                  /* the waitcnt pass drain the queue, the next row's loads included       */ \
        const float sd = s_lane ? (CUR.sj - s_i) : 0.f;                                                                      \
        gs_diff += sd;                                                                                                       \
        const float td = sd + bd;                                                                                            \
        const uint64_t dsg = __ballot(s_lane && td > 0.f), dnz = __ballot(s_lane && td != 0.f);                              \
        float ve[3], z[3][3];                                                                                                \
        ve[0] = diff_lane ? (CUR.vj0 - vi[0]) : vi[0];                                                                       \
        ve[1] = diff_lane ? (CUR.vj1 - vi[1]) : vi[1];                                                                       \
        ve[2] = diff_lane ? (CUR.vj2 - vi[2]) : vi[2];                                                                       \
        _Pragma("unroll") for (int dd = 0; dd < 3; ++dd)                                                                     \
            _Pragma("unroll") for (int jz = 0; jz < 3; ++jz) z[dd][jz] = CUR.z[dd * 3 + jz] + zi[dd][jz];                    \
        uint64_t vsg[3], vnz[3], vst[3];                                                                                     \
        _Pragma("unroll") for (int jz = 0; jz < 3; ++jz) {                                                                   \
            const float tvv = ve[0] * z[0][jz] + ve[1] * z[1][jz] + ve[2] * z[2][jz] + bv[jz];                               \
            vsg[jz] = __ballot(v2_lane && tvv > 0.f);                                                                        \
            vnz[jz] = __ballot(v2_lane && tvv != 0.f);                                                                       \
            vst[jz] = save ? __ballot(v2_lane && fabsf(tvv) <= 1.2f) : 0ull;                                                 \
        }                                                                                                                    \
        const int64_t e = gp * k + t;                                                                                        \
        if (save) { /* wave-uniform.  planes[e][plane][word]: lane 5*plane + word holds one 64-bit word of the edge row */  \
            const uint64_t dst = __ballot(s_lane && fabsf(td) <= 1.2f);                                                      \
            /* v_writelane drops each (wave-uniform) ballot word straight into its lane: no lane masks, no selects */       \
            int vlo = 0, vhi = 0;                                                                                            \
            SVNET_WL(dsg, 0);  SVNET_WL(csg, 1);  SVNET_WL(vsg[0], 2);  SVNET_WL(vsg[1], 3);  SVNET_WL(vsg[2], 4);           \
            SVNET_WL(dnz, 5);  SVNET_WL(cnz, 6);  SVNET_WL(vnz[0], 7);  SVNET_WL(vnz[1], 8);  SVNET_WL(vnz[2], 9);           \
            SVNET_WL(dst, 10); SVNET_WL(cst, 11); SVNET_WL(vst[0], 12); SVNET_WL(vst[1], 13); SVNET_WL(vst[2], 14);          \
            if (lane < 3 * NW) o_planes[e * (3 * NW) + lane] = ((uint64_t)(uint32_t)vhi << 32) | (uint32_t)vlo;             \
        }                                                                                                                    \
        /* (DENSE: the edge's own count of non-zero features - wave-uniform, on the scalar unit) */                          \
        const int pme = DENSE ? (int)(__popcll((uint64_t)(word_t)dnz) + __popcll((uint64_t)(word_t)vnz[0]) + __popcll((uint64_t)(word_t)vnz[1]) + \
                                      __popcll((uint64_t)(word_t)vnz[2])) : 0;                                               \
        _Pragma("unroll") for (int op = 0; op < OP; ++op) {                                                                  \
            int pm_ = base[op] + pme, pd_ = 0;                                                                               \
            if (DENSE) {                                                                                                     \
                tacc_dense((word_t)dsg, (word_t)dnz, wsg[op][0], pd_);                                                       \
                _Pragma("unroll") for (int jz = 0; jz < 3; ++jz) tacc_dense((word_t)vsg[jz], (word_t)vnz[jz], wsg[op][2 + jz], pd_); \
            } else {                                                                                                         \
                tacc((word_t)dsg, (word_t)dnz, wsg[op][0], wnz[op][0], pm_, pd_);                                            \
                _Pragma("unroll") for (int jz = 0; jz < 3; ++jz) tacc((word_t)vsg[jz], (word_t)vnz[jz], wsg[op][2 + jz], wnz[op][2 + jz], pm_, pd_); \
            }                                                                                                                \
            const int n = pm_ - 2 * pd_;                                                                                     \
            if (n > nmax[op]) { nmax[op] = n; smax[op] = t; }                                                                \
            if (n < nmin[op]) { nmin[op] = n; smin[op] = t; }                                                                \
            sn[op] += n;                                                                                                     \
            sn2[op] += n * n;                                                                                                \
            if (save && lane + 64 * op < Os) o_n16[e * Os + lane + 64 * op] = (int16_t)n;   /* |n| <= 320 */                 \
        }                                                                                                                    \
        if (o_lane) {                                                                                                        \
            const float vp0 = CUR.u0 + ub[0], vp1 = CUR.u1 + ub[1], vp2 = CUR.u2 + ub[2];                                    \
            const float nn = fast_sqrt(vp0 * vp0 + vp1 * vp1 + vp2 * vp2) + VEPS;                                            \
            const float inv = fast_rcp(nn);                                                                                  \
            av[0] += vp0; av[1] += vp1; av[2] += vp2;                                                                        \
            avn[0] += vp0 * inv; avn[1] += vp1 * inv; avn[2] += vp2 * inv;                                                   \
            sv1 += (double)nn;                                                                                               \
            sv2 += (double)nn * (double)nn;                                                                                  \
        }                                                                                                                    \
    } while (0)
        SVNET_LOAD_NBR(na, 0);
        // pairs without a skippable half (a path on which a slot's loads are never consumed costs a full drain per edge);
        // an odd k ends with one more edge outside the loop
        int t2 = 0;
        for (; t2 + 1 < k; t2 += 2) {
            SVNET_EDGE(na, nb, t2);
            SVNET_EDGE(nb, na, t2 + 1);
        }
        if (t2 < k) SVNET_EDGE(na, nb, t2);
#undef SVNET_EDGE
#undef SVNET_WL
#undef SVNET_LOAD_NBR
        const float invk = 1.f / (float)k;
#pragma unroll
        for (int op = 0; op < OP; ++op) {
            const int o = lane + 64 * op;
            if (o < Os) {
                o_nmax[gp * Os + o] = nmax[op];
                o_nmin[gp * Os + o] = nmin[op];
                o_smax[gp * Os + o] = (uint8_t)smax[op];
                o_smin[gp * Os + o] = (uint8_t)smin[op];
            }
        }
        if (o_lane) {
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
                o_mv[(gp * 3 + dd) * Ov + lane] = av[dd] * invk;
                o_mvn[(gp * 3 + dd) * Ov + lane] = avn[dd] * invk;
            }
        }
    }
    // ---- batch statistics: all waves of the grid add into the same 2*Os + 2*Ov addresses, so the workgroup's four
    // waves are combined in LDS first (same-address atomics serialise at the memory side)
    __shared__ unsigned long long red_n[2 * 128];
    __shared__ double red_v[2 * 64];
    if (d.stat_n) {
        for (int i = threadIdx.x; i < 2 * Os; i += blockDim.x) red_n[i] = 0ull;
        for (int i = threadIdx.x; i < 2 * Ov; i += blockDim.x) red_v[i] = 0.0;
        __syncthreads();
        if (p_begin < p_end) {
#pragma unroll
            for (int op = 0; op < OP; ++op) {
                const int o = lane + 64 * op;
                if (o < Os) {
                    atomicAdd(&red_n[o], (unsigned long long)sn[op]);
                    atomicAdd(&red_n[Os + o], (unsigned long long)sn2[op]);
                }
            }
            if (o_lane) {
                atomicAdd(&red_v[lane], sv1);
                atomicAdd(&red_v[Ov + lane], sv2);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * Os; i += blockDim.x)
            if (red_n[i] != 0ull) atomicAdd(reinterpret_cast<unsigned long long*>(d.stat_n) + (blockIdx.x & (SVNET_RED_SLICES - 1)) * 2 * Os + i, red_n[i]);
        for (int i = threadIdx.x; i < 2 * Ov; i += blockDim.x)
            if (red_v[i] != 0.0) atomicAdd(&d.stat_v[(blockIdx.x & (SVNET_RED_SLICES - 1)) * 2 * Ov + i], red_v[i]);
    }
    if (p_begin < p_end && s_lane) {
        // fp64 atomics: the sum does not depend (to fp32 precision) on the order in which the grid's waves arrive, so the
        // gate -- and through it every sign() downstream -- is reproducible from run to run
        atomicAdd(&d.gate_sum[b * 2 * Cs + lane], (double)gs_diff);
        atomicAdd(&d.gate_sum[b * 2 * Cs + Cs + lane], (double)gs_cen * (double)k);
    }
}

// register budget: 4 waves per SIMD (<= 128 VGPRs) up to Os = 64, 3 (<= 168) above: the kernel sat 1 and 3 registers over
template <int OP, bool NARROW>
__global__ __launch_bounds__(256, (OP == 1 ? 4 : 3)) void edgeblock_fwd_kernel(FwdArgs fa) {
    const svnet_edgeblock_desc& d = fa.d;
    // (a wave-uniform branch between the two instantiations: the weights of a layer hold no exact zero in practice)
    if (edge_weights_dense(d.w_dense))
        edgeblock_fwd_body<OP, NARROW, true>(fa, d.s, d.v, d.idx, d.zz, d.ut, d.n16, d.planes, d.n_max, d.n_min, d.slot_max, d.slot_min, d.mv, d.mvn);
    else
        edgeblock_fwd_body<OP, NARROW, false>(fa, d.s, d.v, d.idx, d.zz, d.ut, d.n16, d.planes, d.n_max, d.n_min, d.slot_max, d.slot_min, d.mv, d.mvn);
}

// ---------------------------------------------------------------------------------------------- two edges per wave iteration
// Narrow layers (Cs <= 32, 2Cv <= 32, Ov <= 32, Os <= 64: conv2 / conv3 of the classifiers) fill at most half of a wave's lanes
// in the kernel above.  Here lanes 0..31 work on edge slot 2i and lanes 32..63 on slot 2i+1 of the same point: one ballot yields
// the plane words of BOTH edges (low / high half), every per-edge instruction serves two edges, and the halves are combined
// once per point (arg-max / arg-min with the lower slot winning ties, sums added).  Lane l of either half owns the output channels
// l + 32*q (q < OP2), the scalar channel l, the vector channel l and the v2s channel l.  Same arithmetic per edge as the kernel
// above (bit-identical results).
__device__ __forceinline__ uint32_t half_swap_u32(uint32_t x) {      // value held by lane (l ^ 32)
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (threadIdx.x & 32) ? r[0] : r[1];
}
__device__ __forceinline__ float half_swap_f32(float x) { return __uint_as_float(half_swap_u32(__float_as_uint(x))); }

template <int OP2, bool DENSE>
__device__ __forceinline__ void edgeblock_fwd2_body(const FwdArgs& fa) {
    const svnet_edgeblock_desc& d = fa.d;
    const float* __restrict__ ts = d.s; const float* __restrict__ tv = d.v; const int64_t* __restrict__ tidx = d.idx;
    const float* __restrict__ tzz = d.zz; const float* __restrict__ tut = d.ut;
    int16_t* __restrict__ o_n16 = d.n16; uint64_t* __restrict__ o_planes = d.planes;
    const int lane = threadIdx.x & 63, l = lane & 31;
    const bool hi = lane >= 32;
    int64_t blk = blockIdx.x;
    if ((d.B & 7) == 0 && (fa.waves_per_cloud & 3) == 0) {      // XCD-aware order, as above
        const int64_t bpc = fa.waves_per_cloud >> 2, xcd = blk & 7, slot = blk >> 3;
        blk = ((slot / bpc) * 8 + xcd) * bpc + (slot % bpc);
    }
    const int64_t wave_g = blk * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t bq = wave_g / fa.waves_per_cloud;
    const int64_t b = bq < d.B ? bq : d.B - 1;                 // idle waves keep valid addresses and reach the barriers
    const int wi = (int)(wave_g - bq * fa.waves_per_cloud);
    const int p_begin = wi * fa.points_per_wave;
    const int p_end = bq < d.B ? min((int)d.N, p_begin + fa.points_per_wave) : p_begin;
    const int Cs = d.Cs, Cv = d.Cv, Os = d.Os, Ov = d.Ov, k = (int)d.k;

    uint32_t wsg[OP2][NW], wnz[OP2][NW];
#pragma unroll
    for (int q = 0; q < OP2; ++q) {
        const int o = l + 32 * q;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            wsg[q][w] = (o < Os) ? (uint32_t)d.w_sign[o * NW + w] : 0u;
            wnz[q][w] = DENSE ? 0xFFFFFFFFu : ((o < Os) ? (uint32_t)d.w_nz[o * NW + w] : 0u);
        }
    }
    const float bd = d.beta_perm[l], bc = d.beta_perm[64 + l];
    float bv[3];
#pragma unroll
    for (int jz = 0; jz < 3; ++jz) bv[jz] = d.beta_perm[128 + 64 * jz + l];

    const bool s_lane = l < Cs, v2_lane = l < 2 * Cv, diff_lane = l < Cv, o_lane = l < Ov;
    const int cm = diff_lane ? l : l - Cv;
    const bool save = o_planes != nullptr;

    long long sn[OP2], sn2[OP2];
#pragma unroll
    for (int q = 0; q < OP2; ++q) sn[q] = sn2[q] = 0;
    double sv1 = 0.0, sv2 = 0.0;
    float gs_diff = 0.f, gs_cen = 0.f;

    const int lk = min(lane, k - 1);
    int jv_next = (p_begin < p_end) ? reinterpret_cast<const int*>(tidx)[2 * ((b * d.N + p_begin) * k + lk)] : 0;    // (low dword of the int64 id)
    const uint32_t ls = (uint32_t)min(l, Cs - 1), ldv = (uint32_t)min(l, Cv - 1), lo = (uint32_t)min(l, Ov - 1);
    const uint32_t uCs = (uint32_t)Cs, uCv = (uint32_t)Cv, uOv = (uint32_t)Ov;
    const uint32_t cloud0 = (uint32_t)(b * d.N);
    const int npairs = (k + 1) >> 1;
    for (int p = p_begin; p < p_end; ++p) {
        const int64_t gp = b * d.N + p;
        const int jv = jv_next;
        if (p + 1 < p_end) jv_next = reinterpret_cast<const int*>(tidx)[2 * ((gp + 1) * k + lk)];
        // (the point's own operands: unconditional, clamped loads requested before their first use - see edgeblock_fwd_kernel)
        const float s_raw = ts[gp * Cs + ls];
        float vi_raw[3], zr_raw[3][6], tu_raw[3][2];
        {
            const int cmc = v2_lane ? cm : 0;
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
                vi_raw[dd] = tv[(gp * 3 + dd) * Cv + cmc];
                const float* zrow = tzz + (gp * 3 + dd) * 6;
#pragma unroll
                for (int jz = 0; jz < 6; ++jz) zr_raw[dd][jz] = zrow[jz];
                tu_raw[dd][0] = tut[(gp * 3 + dd) * 2 * Ov + Ov + (int)lo];
                tu_raw[dd][1] = tut[(gp * 3 + dd) * 2 * Ov + (int)lo];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const float s_i = s_lane ? s_raw : 0.f;
        if (!hi) gs_cen += s_i;
        const float tc = s_i + bc;
        const uint64_t csg = __ballot(s_lane && tc > 0.f), cnz = __ballot(s_lane && tc != 0.f);     // both halves identical
        const uint64_t cst = save ? __ballot(s_lane && fabsf(tc) <= 1.2f) : 0ull;
        int base[OP2];
#pragma unroll
        for (int q = 0; q < OP2; ++q) base[q] = tdot((uint32_t)csg, (uint32_t)cnz, wsg[q][1], wnz[q][1]);

        float vi[3], zi[3][3], ub[3];
#pragma unroll
        for (int dd = 0; dd < 3; ++dd) {
            vi[dd] = v2_lane ? vi_raw[dd] : 0.f;
#pragma unroll
            for (int jz = 0; jz < 3; ++jz) zi[dd][jz] = zr_raw[dd][3 + jz] - zr_raw[dd][jz];  // Zq_i - Zp_i
            ub[dd] = o_lane ? tu_raw[dd][0] - tu_raw[dd][1] : 0.f;  // T_i - U_i
        }
        int nmax[OP2], nmin[OP2], smax[OP2], smin[OP2];
#pragma unroll
        for (int q = 0; q < OP2; ++q) { nmax[q] = INT_MIN; nmin[q] = INT_MAX; smax[q] = 0; smin[q] = 0; }
        float av[3] = {0.f, 0.f, 0.f}, avn[3] = {0.f, 0.f, 0.f};

        // this half's neighbour row of pair i (slot 2i + hi, clamped to k-1: an odd k ends with a masked copy of its last edge);
        // rows are requested one pair ahead into two alternating register sets; 32-bit element offsets from SGPR table bases
        struct Nbr { float sj, vj0, vj1, vj2, u0, u1, u2, z[9]; };
        Nbr na = {}, nb = {};
#define SVNET_LOAD_NBR2(N_, I)                                                                               \
    do {                                                                                                     \
        const int ta_ = min(2 * (I), k - 1), tb_ = min(2 * (I) + 1, k - 1);                                  \
        const int ja_ = __builtin_amdgcn_readlane(jv, ta_), jb_ = __builtin_amdgcn_readlane(jv, tb_);        \
        const uint32_t gj_ = cloud0 + (uint32_t)(hi ? jb_ : ja_);                                            \
        const uint32_t os_ = 4u * (gj_ * uCs + ls), ov_ = 4u * (gj_ * 3u * uCv + ldv), ou_ = 4u * (gj_ * 6u * uOv + lo), \
                       oz_ = 4u * (gj_ * 18u);                                                               \
        N_.sj = ld_f32_sbase(ts, os_);                                                                       \
        N_.vj0 = ld_f32_sbase(tv, ov_); N_.vj1 = ld_f32_sbase(tv, ov_ + 4u * uCv); N_.vj2 = ld_f32_sbase(tv, ov_ + 8u * uCv); \
        N_.u0 = ld_f32_sbase(tut, ou_); N_.u1 = ld_f32_sbase(tut, ou_ + 8u * uOv); N_.u2 = ld_f32_sbase(tut, ou_ + 16u * uOv); \
        N_.z[0] = ld_f32_sbase(tzz, oz_); N_.z[1] = ld_f32_sbase(tzz, oz_ + 4u); N_.z[2] = ld_f32_sbase(tzz, oz_ + 8u);        \
        N_.z[3] = ld_f32_sbase(tzz, oz_ + 24u); N_.z[4] = ld_f32_sbase(tzz, oz_ + 28u); N_.z[5] = ld_f32_sbase(tzz, oz_ + 32u); \
        N_.z[6] = ld_f32_sbase(tzz, oz_ + 48u); N_.z[7] = ld_f32_sbase(tzz, oz_ + 52u); N_.z[8] = ld_f32_sbase(tzz, oz_ + 56u); \
    } while (0)
#define SVNET_HALF(W) (hi ? (uint32_t)((W) >> 32) : (uint32_t)(W))
#define SVNET_WL2(W, L)                                                                                      \
    do {                                                                                                     \
        asm("v_writelane_b32 %0, %1, " #L : "+v"(vlo) : "s"((int)(uint32_t)(W)));                            \
        asm("v_writelane_b32 %0, %1, 15+" #L : "+v"(vlo) : "s"((int)(uint32_t)((W) >> 32)));                 \
    } while (0)
#define SVNET_PAIR(CUR, NXT, I)                                                                              \
    do {                                                                                                     \
        const int i_ = (I);                                                                                  \
        SVNET_LOAD_NBR2(NXT, min(i_ + 1, npairs - 1));      /* unconditional: the last pair re-requests itself */ \
        const int t = 2 * i_ + (hi ? 1 : 0);                                                                 \
        const bool ok = t < k;                               /* only the high half of an odd k's last pair is not */ \
        const float sd = s_lane ? (CUR.sj - s_i) : 0.f;                                                      \
        gs_diff += ok ? sd : 0.f;                                                                            \
        const float td = sd + bd;                                                                            \
        const uint64_t dsg = __ballot(s_lane && td > 0.f), dnz = __ballot(s_lane && td != 0.f);              \
        float ve[3], z[3][3];                                                                                \
        ve[0] = diff_lane ? (CUR.vj0 - vi[0]) : vi[0];                                                       \
        ve[1] = diff_lane ? (CUR.vj1 - vi[1]) : vi[1];                                                       \
        ve[2] = diff_lane ? (CUR.vj2 - vi[2]) : vi[2];                                                       \
        _Pragma("unroll") for (int dd = 0; dd < 3; ++dd)                                                     \
            _Pragma("unroll") for (int jz = 0; jz < 3; ++jz) z[dd][jz] = CUR.z[dd * 3 + jz] + zi[dd][jz];    \
        uint64_t vsg[3], vnz[3], vst[3];                                                                     \
        _Pragma("unroll") for (int jz = 0; jz < 3; ++jz) {                                                   \
            const float tvv = ve[0] * z[0][jz] + ve[1] * z[1][jz] + ve[2] * z[2][jz] + bv[jz];               \
            vsg[jz] = __ballot(v2_lane && tvv > 0.f);                                                        \
            vnz[jz] = __ballot(v2_lane && tvv != 0.f);                                                       \
            vst[jz] = save ? __ballot(v2_lane && fabsf(tvv) <= 1.2f) : 0ull;                                 \
        }                                                                                                    \
        const int64_t e0 = gp * k + 2 * i_;                  /* edge row of the low half; the high half's is e0 + 1 */ \
        if (save) { /* planes[e][plane][word]: lanes 0..14 hold the low half's 15 words, lanes 15..29 the high half's */ \
            const uint64_t dst = __ballot(s_lane && fabsf(td) <= 1.2f);                                      \
            int vlo = 0;                                                                                     \
            SVNET_WL2(dsg, 0);  SVNET_WL2(csg, 1);  SVNET_WL2(vsg[0], 2);  SVNET_WL2(vsg[1], 3);  SVNET_WL2(vsg[2], 4);      \
            SVNET_WL2(dnz, 5);  SVNET_WL2(cnz, 6);  SVNET_WL2(vnz[0], 7);  SVNET_WL2(vnz[1], 8);  SVNET_WL2(vnz[2], 9);      \
            SVNET_WL2(dst, 10); SVNET_WL2(cst, 11); SVNET_WL2(vst[0], 12); SVNET_WL2(vst[1], 13); SVNET_WL2(vst[2], 14);     \
            if (lane < ((2 * i_ + 1 < k) ? 30 : 15)) o_planes[e0 * (3 * NW) + lane] = (uint64_t)(uint32_t)vlo; \
        }                                                                                                    \
        const uint32_t m_dsg = SVNET_HALF(dsg), m_dnz = SVNET_HALF(dnz);                                      \
        uint32_t m_vsg[3], m_vnz[3];                                                                         \
        _Pragma("unroll") for (int jz = 0; jz < 3; ++jz) { m_vsg[jz] = SVNET_HALF(vsg[jz]); m_vnz[jz] = SVNET_HALF(vnz[jz]); } \
        /* (DENSE weights: the half's own count of non-zero features, once per edge instead of an and + bcnt per word and channel) */ \
        const int pme = DENSE ? (__popc(m_dnz) + __popc(m_vnz[0]) + __popc(m_vnz[1]) + __popc(m_vnz[2])) : 0;   \
        _Pragma("unroll") for (int q = 0; q < OP2; ++q) {                                                    \
            int pm_ = base[q] + pme, pd_ = 0;                                                                \
            if (DENSE) {                                                                                     \
                tacc_dense(m_dsg, m_dnz, wsg[q][0], pd_);                                                    \
                _Pragma("unroll") for (int jz = 0; jz < 3; ++jz) tacc_dense(m_vsg[jz], m_vnz[jz], wsg[q][2 + jz], pd_); \
            } else {                                                                                         \
                tacc(m_dsg, m_dnz, wsg[q][0], wnz[q][0], pm_, pd_);                                          \
                _Pragma("unroll") for (int jz = 0; jz < 3; ++jz) tacc(m_vsg[jz], m_vnz[jz], wsg[q][2 + jz], wnz[q][2 + jz], pm_, pd_); \
            }                                                                                                \
            const int n = pm_ - 2 * pd_;                                                                     \
            if (ok) {                                                                                        \
                if (n > nmax[q]) { nmax[q] = n; smax[q] = t; }                                               \
                if (n < nmin[q]) { nmin[q] = n; smin[q] = t; }                                               \
                sn[q] += n;                                                                                  \
                sn2[q] += n * n;                                                                             \
                if (save && l + 32 * q < Os) o_n16[(e0 + (hi ? 1 : 0)) * Os + l + 32 * q] = (int16_t)n;      \
            }                                                                                                \
        }                                                                                                    \
        if (o_lane && ok) {                                                                                  \
            const float vp0 = CUR.u0 + ub[0], vp1 = CUR.u1 + ub[1], vp2 = CUR.u2 + ub[2];                    \
            const float nn = fast_sqrt(vp0 * vp0 + vp1 * vp1 + vp2 * vp2) + VEPS;                            \
            const float inv = fast_rcp(nn);                                                                  \
            av[0] += vp0; av[1] += vp1; av[2] += vp2;                                                        \
            avn[0] += vp0 * inv; avn[1] += vp1 * inv; avn[2] += vp2 * inv;                                   \
            sv1 += (double)nn;                                                                               \
            sv2 += (double)nn * (double)nn;                                                                  \
        }                                                                                                    \
    } while (0)
        SVNET_LOAD_NBR2(na, 0);
        int i2 = 0;
        for (; i2 + 1 < npairs; i2 += 2) {
            SVNET_PAIR(na, nb, i2);
            SVNET_PAIR(nb, na, i2 + 1);
        }
        if (i2 < npairs) SVNET_PAIR(na, nb, i2);
#undef SVNET_PAIR
#undef SVNET_WL2
#undef SVNET_HALF
#undef SVNET_LOAD_NBR2
        // ---- the two halves of the point: extreme sums (ties -> the lower slot, i.e. the first index) and vector means
        const float invk = 1.f / (float)k;
#pragma unroll
        for (int q = 0; q < OP2; ++q) {
            const int on = (int)half_swap_u32((uint32_t)nmax[q]), os = (int)half_swap_u32((uint32_t)smax[q]);
            if (on > nmax[q] || (on == nmax[q] && os < smax[q])) { nmax[q] = on; smax[q] = os; }
            const int un = (int)half_swap_u32((uint32_t)nmin[q]), us = (int)half_swap_u32((uint32_t)smin[q]);
            if (un < nmin[q] || (un == nmin[q] && us < smin[q])) { nmin[q] = un; smin[q] = us; }
            const int o = l + 32 * q;
            if (!hi && o < Os) {
                d.n_max[gp * Os + o] = nmax[q];
                d.n_min[gp * Os + o] = nmin[q];
                d.slot_max[gp * Os + o] = (uint8_t)smax[q];
                d.slot_min[gp * Os + o] = (uint8_t)smin[q];
            }
        }
#pragma unroll
        for (int dd = 0; dd < 3; ++dd) {
            const float a = av[dd] + half_swap_f32(av[dd]), an = avn[dd] + half_swap_f32(avn[dd]);
            if (!hi && o_lane) {
                d.mv[(gp * 3 + dd) * Ov + l] = a * invk;
                d.mvn[(gp * 3 + dd) * Ov + l] = an * invk;
            }
        }
    }
    // ---- batch statistics (as above; both halves add into their channel's slot)
    __shared__ unsigned long long red_n[2 * 64];
    __shared__ double red_v[2 * 32];
    if (d.stat_n) {
        for (int i = threadIdx.x; i < 2 * Os; i += blockDim.x) red_n[i] = 0ull;
        for (int i = threadIdx.x; i < 2 * Ov; i += blockDim.x) red_v[i] = 0.0;
        __syncthreads();
        if (p_begin < p_end) {
#pragma unroll
            for (int q = 0; q < OP2; ++q) {
                const int o = l + 32 * q;
                if (o < Os) {
                    atomicAdd(&red_n[o], (unsigned long long)sn[q]);
                    atomicAdd(&red_n[Os + o], (unsigned long long)sn2[q]);
                }
            }
            if (o_lane) {
                atomicAdd(&red_v[l], sv1);
                atomicAdd(&red_v[Ov + l], sv2);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * Os; i += blockDim.x)
            if (red_n[i] != 0ull) atomicAdd(reinterpret_cast<unsigned long long*>(d.stat_n) + (blockIdx.x & (SVNET_RED_SLICES - 1)) * 2 * Os + i, red_n[i]);
        for (int i = threadIdx.x; i < 2 * Ov; i += blockDim.x)
            if (red_v[i] != 0.0) atomicAdd(&d.stat_v[(blockIdx.x & (SVNET_RED_SLICES - 1)) * 2 * Ov + i], red_v[i]);
    }
    if (p_begin < p_end && s_lane) {
        atomicAdd(&d.gate_sum[b * 2 * Cs + l], (double)gs_diff);
        if (!hi) atomicAdd(&d.gate_sum[b * 2 * Cs + Cs + l], (double)gs_cen * (double)k);
    }
}

template <int OP2>
__global__ __launch_bounds__(256, 4) void edgeblock_fwd2_kernel(FwdArgs fa) {
    if (edge_weights_dense(fa.d.w_dense)) edgeblock_fwd2_body<OP2, true>(fa);      // (wave-uniform: see edgeblock_fwd_kernel)
    else edgeblock_fwd2_body<OP2, false>(fa);
}

// Per-channel affine forms from the batch (or running) statistics.
//   scalar: y = A1*n + B1 with A1 = gamma*scale*invstd_y,  B1 = beta - gamma*mean_y*invstd_y   (y_pre = scale*n)
//   vector: q(n') = Av + Bv/n' with Av = gamma'*invstd', Bv = beta' - gamma'*mean'*invstd'
// coef layout: [A1 (Os) | B1 (Os) | mean_y (Os) | invstd_y (Os) | Av (Ov) | Bv (Ov) | mean' (Ov) | invstd' (Ov)]
struct EdgeCoefArgs {
    const long long* stat_n; const double* stat_v; int64_t E; int Os, Ov;
    const float* scale1; const float* g1; const float* b1; float* rm1; float* rv1;
    const float* g2; const float* b2; float* rm2; float* rv2;
    int training; float eps, momentum;
};
// channel c of both coefficient sets into `out` (the coef layout; global memory or a workgroup's LDS copy).  commit: this caller also
// updates the running statistics (exactly one workgroup of a launch does).  One body for the coefficient kernel and the tail kernel.
__device__ __forceinline__ void edge_coefs_channel(const EdgeCoefArgs& a, int c, bool commit, float* out) {
    const int Os = a.Os, Ov = a.Ov;
    const int64_t E = a.E;
    float* A1 = out; float* B1 = out + Os; float* MY = out + 2 * Os; float* IY = out + 3 * Os;
    float* Av = out + 4 * Os; float* Bv = Av + Ov; float* MV = Av + 2 * Ov; float* IV = Av + 3 * Ov;
    if (c < Os) {
        float mean, invstd;
        if (a.training) {
            const double sc = (double)a.scale1[c];
            long long sn1 = 0, sn2 = 0;                        // the forward kernel's slices (exact integers: any order)
            for (int sl = 0; sl < SVNET_RED_SLICES; ++sl) { sn1 += a.stat_n[sl * 2 * Os + c]; sn2 += a.stat_n[sl * 2 * Os + Os + c]; }
            const double mn = (double)sn1 / (double)E;
            double var_n = (double)sn2 / (double)E - mn * mn;
            if (var_n < 0.0) var_n = 0.0;
            const double m = sc * mn, var = sc * sc * var_n;
            mean = (float)m;
            invstd = (float)(1.0 / sqrt(var + (double)a.eps));
            if (commit && a.rm1) a.rm1[c] = (1.f - a.momentum) * a.rm1[c] + a.momentum * mean;
            if (commit && a.rv1) a.rv1[c] = (1.f - a.momentum) * a.rv1[c] + a.momentum * (float)(E > 1 ? var * ((double)E / (double)(E - 1)) : var);
        } else {
            mean = a.rm1[c];
            invstd = 1.f / sqrtf(a.rv1[c] + a.eps);
        }
        A1[c] = a.g1[c] * a.scale1[c] * invstd;
        B1[c] = a.b1[c] - a.g1[c] * mean * invstd;
        MY[c] = mean;
        IY[c] = invstd;
    }
    if (c < Ov) {
        float mean, invstd;
        if (a.training) {
            double sv1 = 0.0, sv2 = 0.0;
            for (int sl = 0; sl < SVNET_RED_SLICES; ++sl) { sv1 += a.stat_v[sl * 2 * Ov + c]; sv2 += a.stat_v[sl * 2 * Ov + Ov + c]; }
            const double m = sv1 / (double)E;
            double var = sv2 / (double)E - m * m;
            if (var < 0.0) var = 0.0;
            mean = (float)m;
            invstd = (float)(1.0 / sqrt(var + (double)a.eps));
            if (commit && a.rm2) a.rm2[c] = (1.f - a.momentum) * a.rm2[c] + a.momentum * mean;
            if (commit && a.rv2) a.rv2[c] = (1.f - a.momentum) * a.rv2[c] + a.momentum * (float)(E > 1 ? var * ((double)E / (double)(E - 1)) : var);
        } else {
            mean = a.rm2[c];
            invstd = 1.f / sqrtf(a.rv2[c] + a.eps);
        }
        Av[c] = a.g2[c] * invstd;
        Bv[c] = a.b2[c] - a.g2[c] * mean * invstd;
        MV[c] = mean;
        IV[c] = invstd;
    }
}

__global__ void edgeblock_coeffs_kernel(EdgeCoefArgs a, float* __restrict__ coef, long long* __restrict__ nbt1, long long* __restrict__ nbt2,
                                        svnet_gate_fwd_job job, int coef_blocks) {
    if ((int)blockIdx.x >= coef_blocks) { svnet_gate_fwd_block(job, (int)blockIdx.x - coef_blocks); return; }   // the gate MLP beside the coefficients
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && a.training) {
        if (nbt1) *nbt1 += 1;
        if (nbt2) *nbt2 += 1;
    }
    edge_coefs_channel(a, c, true, coef);
}

// Pooled outputs: s_out = lrelu(A1 * (A1 >= 0 ? n_max : n_min) + B1); v_out = gate * (Av*mv + Bv*mvn).
// (one functor for both apply kernels below: the same expressions, so the same contraction - their outputs are bit-identical,
//  tests/test_hip_fused.py)
struct EdgeApplyMath {
    const int32_t* __restrict__ n_max; const int32_t* __restrict__ n_min;
    const float* __restrict__ mv; const float* __restrict__ mvn;
    const float* __restrict__ A1; const float* __restrict__ B1; const float* __restrict__ Av; const float* __restrict__ Bv;
    const float* __restrict__ gate;
    int Os, Ov;
    float slope;
    __device__ __forceinline__ float s(int64_t p, int o) const {
        const float a = A1[o];
        const float y = a * (float)(a >= 0.f ? n_max[p * Os + o] : n_min[p * Os + o]) + B1[o];
        return y > 0.f ? y : y * slope;
    }
    __device__ __forceinline__ float v(int64_t p, int64_t b, int q, int c) const {
        const int64_t e = p * 3 * Ov + q;
        return gate[b * Ov + c] * (Av[c] * mv[e] + Bv[c] * mvn[e]);
    }
};

__global__ __launch_bounds__(256) void edgeblock_apply_kernel(const int32_t* __restrict__ n_max, const int32_t* __restrict__ n_min,
                                                              const float* __restrict__ mv, const float* __restrict__ mvn,
                                                              const float* __restrict__ coef, const float* __restrict__ gate,
                                                              int64_t P, int64_t N, int Os, int Ov, float slope,
                                                              float* __restrict__ s_out, float* __restrict__ v_out, float* __restrict__ s_cat,
                                                              int64_t s_ld, float* __restrict__ v_cat, int64_t v_ld) {
    const EdgeApplyMath m = {n_max, n_min, mv, mvn, coef, coef + Os, coef + 4 * Os, coef + 4 * Os + Ov, gate, Os, Ov, slope};
    // a wave per point row: lanes over the Os scalar channels, then over the 3*Ov vector entries - no per-element divisions (the flat
    // e -> (e % Os, q % Ov, q / 3Ov, p / N) form spent four 64-bit divisions on every output)
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t p = wave0; p < P; p += nwaves) {
        const int64_t b = p / N;
        for (int o = lane; o < Os; o += 64) {
            const float z = m.s(p, o);
            s_out[p * Os + o] = z;
            if (s_cat) s_cat[p * s_ld + o] = z;            // (the level's column slice of the pyramid's concatenation, written in place)
        }
        for (int q = lane; q < 3 * Ov; q += 64) {
            const int dd = q >= 2 * Ov ? 2 : (q >= Ov ? 1 : 0), c = q - dd * Ov;
            const float z = m.v(p, b, q, c);
            v_out[p * 3 * Ov + q] = z;
            if (v_cat) v_cat[(p * 3 + dd) * v_ld + c] = z;
        }
    }
}

// ... and the same pass preparing the k-NN table of its output (apply_knn.h)
__global__ __launch_bounds__(256) void edgeblock_apply_knn_kernel(const int32_t* __restrict__ n_max, const int32_t* __restrict__ n_min,
                                                                  const float* __restrict__ mv, const float* __restrict__ mvn,
                                                                  const float* __restrict__ coef, const float* __restrict__ gate,
                                                                  int64_t P, int64_t N, int Os, int Ov, float slope,
                                                                  float* __restrict__ s_out, float* __restrict__ v_out,
                                                                  float* __restrict__ s_cat, int64_t s_ld, float* __restrict__ v_cat,
                                                                  int64_t v_ld, float* __restrict__ xT, float* __restrict__ xx, int64_t Cpad) {
    extern __shared__ float apply_knn_rows[];
    const EdgeApplyMath m = {n_max, n_min, mv, mvn, coef, coef + Os, coef + 4 * Os, coef + 4 * Os + Ov, gate, Os, Ov, slope};
    apply_knn_tiles<APPLY_KNN_TP>(m, P, N, Os, Ov, s_out, v_out, s_cat, s_ld, v_cat, v_ld, xT, xx, Cpad, apply_knn_rows);
}

// ---- coefficients + gate MLP + apply (+ the next k-NN's table) in one launch (svnet_hip.h: svnet_block_tail_desc).  A workgroup = one
// tile of APPLY_KNN_TP points of cloud b: its 256 threads derive the coefficients into LDS (thread c: channel c of both sets, as the
// coefficient kernel's threads do), run cloud b's gate MLP (every workgroup of the cloud writes the same h / gin / gate values), then the
// apply pass reads both from there.  Workgroup 0 alone commits: coef, running statistics, counters.
__global__ __launch_bounds__(256) void edgeblock_tail_kernel(EdgeCoefArgs ca, float* __restrict__ coef, long long* __restrict__ nbt1,
                                                             long long* __restrict__ nbt2, svnet_gate_fwd_job job,
                                                             const int32_t* __restrict__ n_max, const int32_t* __restrict__ n_min,
                                                             const float* __restrict__ mv, const float* __restrict__ mvn, int64_t P, int64_t N,
                                                             float slope, float* __restrict__ s_out, float* __restrict__ v_out,
                                                             float* __restrict__ s_cat, int64_t s_ld, float* __restrict__ v_cat, int64_t v_ld,
                                                             float* __restrict__ xT, float* __restrict__ xx, int64_t Cpad) {
    extern __shared__ float tail_lds[];                                  // [coef: 4 Os + 4 Ov (rounded to 4) | the tile's rows]
    const int Os = ca.Os, Ov = ca.Ov;
    const int ncoef = (4 * Os + 4 * Ov + 3) & ~3;
    const bool first = blockIdx.x == 0;
    const int64_t b = ((int64_t)blockIdx.x * APPLY_KNN_TP) / N;
    if (first && threadIdx.x == 0 && ca.training) {
        if (nbt1) *nbt1 += 1;
        if (nbt2) *nbt2 += 1;
    }
    edge_coefs_channel(ca, (int)threadIdx.x, first, tail_lds);
    svnet_gate_fwd_block(job, (int)b);
    __syncthreads();                                                     // the coefficients in LDS, the cloud's gate in global memory
    if (first)
        for (int i = threadIdx.x; i < 4 * Os + 4 * Ov; i += blockDim.x) coef[i] = tail_lds[i];
    const EdgeApplyMath m = {n_max, n_min, mv, mvn, tail_lds, tail_lds + Os, tail_lds + 4 * Os, tail_lds + 4 * Os + Ov, job.gate, Os, Ov, slope};
    apply_knn_tiles<APPLY_KNN_TP>(m, P, N, Os, Ov, s_out, v_out, s_cat, s_ld, v_cat, v_ld, xT, xx, Cpad, tail_lds + ncoef);
}

}  // namespace

extern "C" int svnet_edgeblock_prepare_f32(const float* W, const float* beta, int64_t Os, int64_t Cs, int64_t Cv, uint64_t* w_sign,
                                           uint64_t* w_nz, float* beta_perm, void* stream) {
    SVNET_REQUIRE(W && beta && w_sign && w_nz && beta_perm, SVNET_E_ARG, "svnet_edgeblock_prepare_f32: null pointer");
    SVNET_REQUIRE(Cs > 0 && Cs <= 64 && Cv > 0 && 2 * Cv <= 64 && Os > 0, SVNET_E_UNSUPPORTED, "svnet_edgeblock_prepare_f32: needs Cs <= 64, 2*Cv <= 64");
    hipLaunchKernelGGL(edgeblock_prepare_kernel, dim3((unsigned)svnet_cdiv((Os * NW + NW) * 64, 256)), dim3(256), 0, (hipStream_t)stream, W,
                       beta, (int)Os, (int)Cs, (int)Cv, w_sign, w_nz, beta_perm);
    SVNET_CHECK_LAUNCH("edgeblock_prepare_kernel");
    return SVNET_OK;
}

extern "C" int svnet_edgeblock_fwd_f32(const svnet_edgeblock_desc* desc, void* stream) {
    SVNET_REQUIRE(desc, SVNET_E_ARG, "svnet_edgeblock_fwd_f32: null descriptor");
    const svnet_edgeblock_desc& d = *desc;
    SVNET_REQUIRE(d.s && d.v && d.idx && d.zz && d.ut && d.w_sign && d.w_nz && d.beta_perm && d.n_max && d.n_min && d.slot_max &&
                      d.slot_min && d.mv && d.mvn && d.gate_sum, SVNET_E_ARG, "svnet_edgeblock_fwd_f32: null pointer");
    SVNET_REQUIRE((d.stat_n == nullptr) == (d.stat_v == nullptr), SVNET_E_ARG, "svnet_edgeblock_fwd_f32: pass both stat buffers or none");
    SVNET_REQUIRE((d.n16 == nullptr) == (d.planes == nullptr), SVNET_E_ARG, "svnet_edgeblock_fwd_f32: pass both n16 and planes or none");
    SVNET_REQUIRE(d.B >= 0 && d.N > 0 && d.k > 0, SVNET_E_ARG, "svnet_edgeblock_fwd_f32: bad sizes");
    SVNET_REQUIRE(d.k <= 64, SVNET_E_UNSUPPORTED, "svnet_edgeblock_fwd_f32: k=%lld > 64", (long long)d.k);
    SVNET_REQUIRE(d.Cs > 0 && d.Cs <= 64 && d.Cv > 0 && 2 * d.Cv <= 64 && d.Os > 0 && d.Os <= 128 && d.Ov > 0 && d.Ov <= 64,
                  SVNET_E_UNSUPPORTED, "svnet_edgeblock_fwd_f32: channel counts outside Cs<=64, 2Cv<=64, Os<=128, Ov<=64");
    if (d.B == 0) return SVNET_OK;
    FwdArgs fa;
    fa.d = d;
    int wpc = (int)svnet_cdiv(4096, d.B);                 // ~4096 waves in flight (16 per CU)
    if (wpc > d.N) wpc = (int)d.N;
    if (wpc < 1) wpc = 1;
    fa.points_per_wave = (int)svnet_cdiv(d.N, wpc);
    fa.waves_per_cloud = (int)svnet_cdiv(d.N, fa.points_per_wave);
    const int64_t waves = d.B * fa.waves_per_cloud;
    const unsigned grid = (unsigned)svnet_cdiv(waves, 4);
    const bool narrow = d.Cs <= 32 && 2 * d.Cv <= 32;
    static const bool no_pairs = getenv("SVNET_FWD_NO_PAIRS") != nullptr;      // (diagnostic: the one-edge-per-iteration kernel for narrow layers too)
    static const bool pairs64 = getenv("SVNET_FWD_PAIRS64") != nullptr;
    if (narrow && d.Ov <= 32 && (d.Os <= 32 || (pairs64 && d.Os <= 64)) && !no_pairs && d.B * d.N * 6 * (int64_t)d.Ov < ((int64_t)1 << 30)) {
        // two edges per wave iteration (32-bit element offsets: the largest table, ut, has B*N*6*Ov floats).  Measured at B=32,
        // N=1024, k=20: Os = 32 (conv2) 196 -> 159 us; Os = 64 (conv3: two output channels per lane, so the popcount part does
        // not shrink, and 72 B of spills at 4 waves per SIMD) 197 -> 195 us - conv3 stays on the one-edge kernel
        if (d.Os <= 32) hipLaunchKernelGGL((edgeblock_fwd2_kernel<1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, fa);
        else hipLaunchKernelGGL((edgeblock_fwd2_kernel<2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, fa);
        SVNET_CHECK_LAUNCH("edgeblock_fwd2_kernel");
        return SVNET_OK;
    }
    if (d.Os <= 64) {
        if (narrow) hipLaunchKernelGGL((edgeblock_fwd_kernel<1, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, fa);
        else hipLaunchKernelGGL((edgeblock_fwd_kernel<1, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, fa);
    } else {
        if (narrow) hipLaunchKernelGGL((edgeblock_fwd_kernel<2, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, fa);
        else hipLaunchKernelGGL((edgeblock_fwd_kernel<2, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, fa);
    }
    SVNET_CHECK_LAUNCH("edgeblock_fwd_kernel");
    return SVNET_OK;
}

extern "C" int svnet_edgeblock_coeffs_f32(const int64_t* stat_n, const double* stat_v, int64_t E, int64_t Os, int64_t Ov,
                                          const float* scale1, const float* gamma1, const float* beta1, float* running_mean1,
                                          float* running_var1, const float* gamma2, const float* beta2, float* running_mean2,
                                          float* running_var2, int training, float eps, float momentum, float* coef,
                                          int64_t* num_batches_tracked1, int64_t* num_batches_tracked2, const svnet_gate_fwd_job* gate_job, void* stream) {
    SVNET_REQUIRE(scale1 && gamma1 && beta1 && gamma2 && beta2 && coef && E > 0 && Os > 0 && Ov > 0, SVNET_E_ARG, "svnet_edgeblock_coeffs_f32: bad arguments");
    SVNET_REQUIRE(training ? (stat_n && stat_v) : (running_mean1 && running_var1 && running_mean2 && running_var2), SVNET_E_ARG,
                  "svnet_edgeblock_coeffs_f32: missing statistics");
    const int64_t n = Os > Ov ? Os : Ov;
    SVNET_REQUIRE(!gate_job || svnet_gate_fwd_job_ok(gate_job), SVNET_E_ARG, "svnet_edgeblock_coeffs_f32: bad gate job");
    const int coef_blocks = (int)svnet_cdiv(n, 256);
    const svnet_gate_fwd_job job = gate_job ? *gate_job : svnet_gate_fwd_job{};
    const EdgeCoefArgs ca = {reinterpret_cast<const long long*>(stat_n), stat_v, E, (int)Os, (int)Ov, scale1, gamma1, beta1, running_mean1,
                             running_var1, gamma2, beta2, running_mean2, running_var2, training, eps, momentum};
    hipLaunchKernelGGL(edgeblock_coeffs_kernel, dim3((unsigned)(coef_blocks + (gate_job ? gate_job->B : 0))), dim3(256), 0, (hipStream_t)stream,
                       ca, coef, reinterpret_cast<long long*>(num_batches_tracked1), reinterpret_cast<long long*>(num_batches_tracked2), job, coef_blocks);
    SVNET_CHECK_LAUNCH("edgeblock_coeffs_kernel");
    return SVNET_OK;
}

extern "C" int svnet_edgeblock_apply_f32(const int32_t* n_max, const int32_t* n_min, const float* mv, const float* mvn,
                                         const float* coef, const float* gate, int64_t P, int64_t N, int64_t Os, int64_t Ov,
                                         float slope, float* s_out, float* v_out, float* s_cat, int64_t s_ld, float* v_cat, int64_t v_ld,
                                         void* stream) {
    SVNET_REQUIRE(n_max && n_min && mv && mvn && coef && gate && s_out && v_out && P >= 0 && N > 0, SVNET_E_ARG, "svnet_edgeblock_apply_f32: bad arguments");
    SVNET_REQUIRE((!s_cat || s_ld >= Os) && (!v_cat || v_ld >= Ov), SVNET_E_ARG, "svnet_edgeblock_apply_f32: concatenation row shorter than the slice");
    if (P == 0) return SVNET_OK;
    hipLaunchKernelGGL(edgeblock_apply_kernel, dim3(svnet_grid(P * 64, 256, 256 * 8)), dim3(256), 0, (hipStream_t)stream, n_max, n_min,
                       mv, mvn, coef, gate, P, N, (int)Os, (int)Ov, slope, s_out, v_out, s_cat, s_ld, v_cat, v_ld);
    SVNET_CHECK_LAUNCH("edgeblock_apply_kernel");
    return SVNET_OK;
}

extern "C" int svnet_edgeblock_apply_knn_f32(const int32_t* n_max, const int32_t* n_min, const float* mv, const float* mvn,
                                             const float* coef, const float* gate, int64_t P, int64_t N, int64_t Os, int64_t Ov,
                                             float slope, float* s_out, float* v_out, float* s_cat, int64_t s_ld, float* v_cat,
                                             int64_t v_ld, void* knn_workspace, size_t knn_workspace_bytes, void* stream) {
    SVNET_REQUIRE(n_max && n_min && mv && mvn && coef && gate && s_out && v_out && knn_workspace && P > 0 && N > 0 && P % N == 0, SVNET_E_ARG,
                  "svnet_edgeblock_apply_knn_f32: bad arguments");
    SVNET_REQUIRE((!s_cat || s_ld >= Os) && (!v_cat || v_ld >= Ov), SVNET_E_ARG, "svnet_edgeblock_apply_knn_f32: concatenation row shorter than the slice");
    int64_t Cpad = 0;
    SVNET_REQUIRE(apply_knn_supported(P, N, Os, Ov, &Cpad), SVNET_E_UNSUPPORTED,
                  "svnet_edgeblock_apply_knn_f32: N=%lld, Os=%lld, Ov=%lld not supported (ask svnet_knn_table_fusable first)", (long long)N,
                  (long long)Os, (long long)Ov);
    SVNET_REQUIRE(knn_workspace_bytes >= svnet_knn_workspace_bytes(P / N, N, Os + 3 * Ov), SVNET_E_WORKSPACE,
                  "svnet_edgeblock_apply_knn_f32: k-NN workspace too small");
    float* xT = (float*)knn_workspace;
    float* xx = xT + P * ((Os + 3 * Ov + 7) / 8 * 8);
    hipLaunchKernelGGL(edgeblock_apply_knn_kernel, dim3((unsigned)(P / APPLY_KNN_TP)), dim3(256), apply_knn_lds_bytes(Os, Ov), (hipStream_t)stream,
                       n_max, n_min, mv, mvn, coef, gate, P, N, (int)Os, (int)Ov, slope, s_out, v_out, s_cat, s_ld, v_cat, v_ld, xT, xx, Cpad);
    SVNET_CHECK_LAUNCH("edgeblock_apply_knn_kernel");
    return SVNET_OK;
}

extern "C" int svnet_block_tail_supported(int64_t P, int64_t N, int64_t Os, int64_t Ov, int with_knn_table) {
    if (!(P > 0 && N > 0 && P % N == 0 && N % APPLY_KNN_TP == 0 && Os > 0 && Os <= 256 && Ov > 0 && Ov <= 256)) return 0;
    int64_t Cpad = 0;
    return (!with_knn_table || apply_knn_supported(P, N, Os, Ov, &Cpad)) ? 1 : 0;
}

// the checks both tail entry points share; returns the launch's dynamic LDS bytes through *lds and the table pointers
static int block_tail_check(const svnet_block_tail_desc& d, const char* who, size_t* lds, float** xT, float** xx, int64_t* Cpad) {
    SVNET_REQUIRE(d.hi && d.lo && d.mv && d.mvn && d.coef && d.s_out && d.v_out && d.gamma1 && d.beta1 && d.gamma2 && d.beta2, SVNET_E_ARG,
                  "%s: null pointer", who);
    SVNET_REQUIRE(d.training ? (d.stat1 && d.stat_v) : (d.running_mean1 && d.running_var1 && d.running_mean2 && d.running_var2), SVNET_E_ARG,
                  "%s: missing statistics", who);
    SVNET_REQUIRE(svnet_gate_fwd_job_ok(&d.gate) && d.gate.Ov == d.Ov && d.gate.B * d.N == d.P, SVNET_E_ARG, "%s: bad gate job", who);
    SVNET_REQUIRE((!d.s_cat || d.s_ld >= d.Os) && (!d.v_cat || d.v_ld >= d.Ov), SVNET_E_ARG, "%s: concatenation row shorter than the slice", who);
    SVNET_REQUIRE(svnet_block_tail_supported(d.P, d.N, d.Os, d.Ov, d.knn_workspace != nullptr), SVNET_E_UNSUPPORTED,
                  "%s: P=%lld N=%lld Os=%lld Ov=%lld not supported (svnet_block_tail_supported)", who, (long long)d.P, (long long)d.N,
                  (long long)d.Os, (long long)d.Ov);
    *xT = nullptr; *xx = nullptr; *Cpad = 0;
    if (d.knn_workspace) {
        SVNET_REQUIRE(d.knn_workspace_bytes >= svnet_knn_workspace_bytes(d.P / d.N, d.N, d.Os + 3 * d.Ov), SVNET_E_WORKSPACE,
                      "%s: k-NN workspace too small", who);
        apply_knn_supported(d.P, d.N, d.Os, d.Ov, Cpad);
        *xT = (float*)d.knn_workspace;
        *xx = *xT + d.P * ((d.Os + 3 * d.Ov + 7) / 8 * 8);
    }
    *lds = (size_t)((4 * d.Os + 4 * d.Ov + 3) & ~(int64_t)3) * sizeof(float) + apply_knn_lds_bytes(d.Os, d.Ov);
    return SVNET_OK;
}

extern "C" int svnet_edgeblock_tail_f32(const svnet_block_tail_desc* desc, void* stream) {
    SVNET_REQUIRE(desc, SVNET_E_ARG, "svnet_edgeblock_tail_f32: null descriptor");
    const svnet_block_tail_desc& d = *desc;
    SVNET_REQUIRE(d.scale1, SVNET_E_ARG, "svnet_edgeblock_tail_f32: scale1 is required");
    size_t lds; float* xT; float* xx; int64_t Cpad;
    const int rc = block_tail_check(d, "svnet_edgeblock_tail_f32", &lds, &xT, &xx, &Cpad);
    if (rc != SVNET_OK) return rc;
    const EdgeCoefArgs ca = {reinterpret_cast<const long long*>(d.stat1), d.stat_v, d.E, (int)d.Os, (int)d.Ov, d.scale1, d.gamma1, d.beta1,
                             d.running_mean1, d.running_var1, d.gamma2, d.beta2, d.running_mean2, d.running_var2, d.training, d.eps, d.momentum};
    hipLaunchKernelGGL(edgeblock_tail_kernel, dim3((unsigned)(d.P / APPLY_KNN_TP)), dim3(256), lds, (hipStream_t)stream, ca, d.coef,
                       reinterpret_cast<long long*>(d.num_batches_tracked1), reinterpret_cast<long long*>(d.num_batches_tracked2), d.gate,
                       (const int32_t*)d.hi, (const int32_t*)d.lo, d.mv, d.mvn, d.P, d.N, d.slope, d.s_out, d.v_out, d.s_cat, d.s_ld, d.v_cat,
                       d.v_ld, xT, xx, Cpad);
    SVNET_CHECK_LAUNCH("edgeblock_tail_kernel");
    return SVNET_OK;
}
