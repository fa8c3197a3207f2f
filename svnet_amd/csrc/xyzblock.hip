// Fused FIRST edge layer (full precision): xyz k-NN gather -> Vector2Scalar -> SVBlock(fp) -> neighbour pooling in one
// pass over the edges, forward and backward, without materialising an edge tensor.
//
// Replaces, for the first layer of the DGCNN models (sv_dgcnn_cls.py:49-53, sv_dgcnn_partseg.py:85-89), the chain
//   get_graph_feature (sv_util.py:28-62) -> init_scalar = Vector2Scalar(2,3) (sv_layers.py:111-129)
//   -> conv1 = SVBlock((6,2),(Os,Ov)) (sv_layers.py:172-196, never binarized) -> svpool (sv_util.py:118-132),
// and (NC = 3) for the first layer of the PointNet models (sv_pointnet_cls.py:35-40, sv_pointnet_partseg.py:56-58) the same chain on
//   get_graph_feature_cross (sv_util.py:64-88) -> Vector2Scalar(3,3) -> conv_pos = SVBlock((9,3),(Os,Ov)).
// An edge row is a function of six floats (x_i, x_j): v_e = [x_j - x_i, x_i] (3x2), s = v2s(v_e; W0) (6),
// s_v = v2s(v_e; Wz) (6), y = W1 [s, s_v] (Os), v' = v_e W2^T (3 x Ov).  One wave per point, lanes are output
// channels; the 12 input features are wave-uniform.  As in edgeblock.hip, max_k commutes with the monotone
// BatchNorm+LeakyReLU (keep max_k y, min_k y and their slots) and VectorBN is affine in (v', v'/|v'|).
// The input coordinates need no gradient, so the backward is a pure reduction into the (tiny) parameter gradients:
// per-lane register accumulators, flushed once per wave with float atomics.
#include <float.h>

#include "common.h"
#include "prelude.h"
#include "gate_mlp.h"
#include "apply_knn.h"

namespace {

constexpr float VEPS = 1e-6f;

struct XyzFwdArgs {
    svnet_xyzblock_desc d;
    int waves_per_cloud, points_per_wave;
};

// v_e and the 6 NC scalar features of one edge (all wave-uniform values, computed redundantly by every lane).
// NC = vector channels of the edge feature: 2 = get_graph_feature [x_j - x_i | x_i] (the DGCNN callers), 3 = get_graph_feature_cross
// [x_j - x_i | x_i | x_j x x_i] (sv_util.py:64-88: the PointNet callers' conv_pos, sv_pointnet_cls.py:35-40).
template <int NC>
struct EdgeFeat {
    float ve[3][NC];
    float f[6 * NC];      // [s (c2*3+jz) | s_v (c2*3+jz)]
};

template <int NC>
__device__ __forceinline__ void edge_features(const float xi[3], const float xj[3], const float (&w0)[3][NC], const float (&wz)[3][NC],
                                              EdgeFeat<NC>& e) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        e.ve[d][0] = xj[d] - xi[d];
        e.ve[d][1] = xi[d];
    }
    if (NC == 3) {      // torch.cross(x_j, x_i) (sv_util.py:84)
        e.ve[0][NC - 1] = xj[1] * xi[2] - xj[2] * xi[1];
        e.ve[1][NC - 1] = xj[2] * xi[0] - xj[0] * xi[2];
        e.ve[2][NC - 1] = xj[0] * xi[1] - xj[1] * xi[0];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float (&w)[3][NC] = h == 0 ? w0 : wz;
        float z[3][3];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int jz = 0; jz < 3; ++jz) {
                float acc = e.ve[d][0] * w[jz][0];
#pragma unroll
                for (int c = 1; c < NC; ++c) acc += e.ve[d][c] * w[jz][c];
                z[d][jz] = acc;
            }
#pragma unroll
        for (int c2 = 0; c2 < NC; ++c2)
#pragma unroll
            for (int jz = 0; jz < 3; ++jz)
                e.f[h * 3 * NC + c2 * 3 + jz] = e.ve[0][c2] * z[0][jz] + e.ve[1][c2] * z[1][jz] + e.ve[2][c2] * z[2][jz];
    }
}

template <int NC>
__device__ __forceinline__ void load_small(const float* __restrict__ p, float (&w)[3][NC]) {
#pragma unroll
    for (int jz = 0; jz < 3; ++jz)
#pragma unroll
        for (int c = 0; c < NC; ++c) w[jz][c] = p[jz * NC + c];
}

// LDS row of one edge, written by the lane that owns the edge and read (broadcast) by the lanes that own the output channels:
// [f (6 NC) | ve[d][c] (3 NC) | backward only: Q_e[c2][c], c2 <= c (NC (NC + 1) / 2)], padded to whole float4s.
template <int NC, bool BWD>
struct FeatRow {
    static constexpr int NF = 6 * NC, NV = 3 * NC, NE = BWD ? NC * (NC + 1) / 2 : 0, NP = (NF + NV + NE + 3) & ~3;
};

// value held by lane (l ^ 32)
__device__ __forceinline__ float xyz_swap32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

// One wave per point.  An edge row is a function of six floats, so its features are computed ONCE, by the lane that owns the edge
// (lane t = neighbour slot t: phase 1, ~80 instructions per POINT, the lane's own neighbour id - no cross-lane read), and left in a
// per-wave LDS row; the output channels then walk the point's edges with lanes = channels (phase 2), EPI = 2 edges per iteration when
// the layer has <= 32 channels (lanes 0-31 edge 2i, lanes 32-63 edge 2i+1; every lane of a half reads its edge's row as broadcast
// float4s).  (Rounds 1-2 computed the wave-uniform features redundantly in all 64 lanes for every edge: ~80 of the ~130 vector
// instructions per edge, at half the lanes.)  Neighbour ids are requested two points ahead, neighbour coordinates one point ahead.
template <int NC, int EPI>
__global__ __launch_bounds__(256) void xyzblock_fwd_kernel(XyzFwdArgs fa) {
    using FR = FeatRow<NC, false>;
    constexpr int NF = 6 * NC, NG = 3 * NC, NV = 3 * NC, NP = FR::NP, HL = 64 / EPI;
    __shared__ __attribute__((aligned(16))) float feat_s[4][64 * NP];
    const svnet_xyzblock_desc& d = fa.d;
    const int lane = threadIdx.x & 63;
    const int wave_l = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* feat = feat_s[wave_l];
    const int64_t wave_g = (int64_t)blockIdx.x * 4 + wave_l;
    const int64_t bq = wave_g / fa.waves_per_cloud;
    const int64_t b = bq < d.B ? bq : d.B - 1;                 // idle waves keep valid addresses and reach the barriers
    const int wi = (int)(wave_g - bq * fa.waves_per_cloud);
    const int p_begin = wi * fa.points_per_wave;
    const int p_end = bq < d.B ? min((int)d.N, p_begin + fa.points_per_wave) : p_begin;
    const int Os = d.Os, Ov = d.Ov, k = (int)d.k;
    const int64_t N = d.N;
    const float* xb = d.x + b * 3 * N;

    float w0[3][NC], wz[3][NC];
    load_small<NC>(d.w0, w0);
    load_small<NC>(d.wz, wz);
    const int o = lane & (HL - 1), half = lane / HL;           // output channel of this lane, which edge of the iteration
    const bool o_lane = o < Os, v_lane = o < Ov;
    float w1[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) w1[f] = o_lane ? d.w1[o * NF + f] : 0.f;
    float w2[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) w2[c] = v_lane ? d.w2[o * NC + c] : 0.f;

    double sy1 = 0.0, sy2 = 0.0, sv1 = 0.0, sv2 = 0.0;
    float gsum[NG];                                            // per EDGE lane (summed over the wave at the end)
#pragma unroll
    for (int f = 0; f < NG; ++f) gsum[f] = 0.f;

    const int lk = min(lane, k - 1);
    const bool e_lane = lane < k;
    const int p_last = max(p_end - 1, 0);
    const int64_t gp0 = b * N;
    int jn1 = (int)d.idx[(gp0 + min(p_begin + 1, p_last)) * k + lk];         // ids of point p + 1
    float xjn[3];
    {
        const int j0 = (int)d.idx[(gp0 + min(p_begin, p_last)) * k + lk];
        xjn[0] = xb[j0]; xjn[1] = xb[N + j0]; xjn[2] = xb[2 * N + j0];
    }
    for (int p = p_begin; p < p_end; ++p) {
        const int64_t gp = gp0 + p;
        const float xj[3] = {xjn[0], xjn[1], xjn[2]};
        {   // unconditional requests from clamped addresses (a branch around a request makes the waitcnt pass drain it)
            const int jq = jn1;
            jn1 = (int)d.idx[(gp0 + min(p + 2, p_last)) * k + lk];
            xjn[0] = xb[jq]; xjn[1] = xb[N + jq]; xjn[2] = xb[2 * N + jq];
        }
        const float xi[3] = {xb[p], xb[N + p], xb[2 * N + p]};
        // ---- phase 1: lane = edge
        {
            EdgeFeat<NC> e;
            edge_features<NC>(xi, xj, w0, wz, e);
            float r[NP];
#pragma unroll
            for (int f = 0; f < NF; ++f) r[f] = e.f[f];
#pragma unroll
            for (int dd = 0; dd < 3; ++dd)
#pragma unroll
                for (int c = 0; c < NC; ++c) r[NF + dd * NC + c] = e.ve[dd][c];
#pragma unroll
            for (int i = NF + NV; i < NP; ++i) r[i] = 0.f;
            if (e_lane) {
#pragma unroll
                for (int f = 0; f < NG; ++f) gsum[f] += e.f[f];
                float4* row = reinterpret_cast<float4*>(feat + lane * NP);
#pragma unroll
                for (int i = 0; i < NP / 4; ++i) row[i] = make_float4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
            }
        }
        __builtin_amdgcn_wave_barrier();          // (one wave: its LDS instructions execute in order; this only pins the compiler's order)
        // ---- phase 2: lane = output channel
        float ymax = -FLT_MAX, ymin = FLT_MAX;
        int smax = 0, smin = 0;
        float av[3] = {0.f, 0.f, 0.f}, avn[3] = {0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < k; t0 += EPI) {
            const int t = t0 + half;
            const bool valid = t < k;
            const float4* row = reinterpret_cast<const float4*>(feat + min(t, k - 1) * NP);
            float r[NP];
#pragma unroll
            for (int i = 0; i < NP / 4; ++i) { const float4 q = row[i]; r[4 * i] = q.x; r[4 * i + 1] = q.y; r[4 * i + 2] = q.z; r[4 * i + 3] = q.w; }
            float y = 0.f;
#pragma unroll
            for (int f = 0; f < NF; ++f) y = fmaf(w1[f], r[f], y);
            if (valid && y > ymax) { ymax = y; smax = t; }
            if (valid && y < ymin) { ymin = y; smin = t; }
            const double yd = valid ? (double)y : 0.0;
            sy1 += yd;
            sy2 += yd * yd;
            float vp[3];
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
                float acc = w2[0] * r[NF + dd * NC];
#pragma unroll
                for (int c = 1; c < NC; ++c) acc += w2[c] * r[NF + dd * NC + c];
                vp[dd] = acc;
            }
            const float nn = fast_sqrt(vp[0] * vp[0] + vp[1] * vp[1] + vp[2] * vp[2]) + VEPS;
            const float inv = valid ? fast_rcp(nn) : 0.f;
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) { const float vv = valid ? vp[dd] : 0.f; av[dd] += vv; avn[dd] += vv * inv; }
            const double nd = valid ? (double)nn : 0.0;
            sv1 += nd;
            sv2 += nd * nd;
        }
        __builtin_amdgcn_wave_barrier();          // the next point's rows overwrite these
        if (EPI == 2) {   // the two halves hold the even / the odd slots: first occurrence of the extremum = larger value, then lower slot
            const float omax = xyz_swap32(ymax), omin = xyz_swap32(ymin);
            const int osmax = __float_as_int(xyz_swap32(__int_as_float(smax))), osmin = __float_as_int(xyz_swap32(__int_as_float(smin)));
            if (omax > ymax || (omax == ymax && osmax < smax)) { ymax = omax; smax = osmax; }
            if (omin < ymin || (omin == ymin && osmin < smin)) { ymin = omin; smin = osmin; }
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) { av[dd] += xyz_swap32(av[dd]); avn[dd] += xyz_swap32(avn[dd]); }
        }
        const float invk = 1.f / (float)k;
        if (o_lane && half == 0) {
            d.y_max[gp * Os + o] = ymax;
            d.y_min[gp * Os + o] = ymin;
            d.slot_max[gp * Os + o] = (uint8_t)smax;
            d.slot_min[gp * Os + o] = (uint8_t)smin;
        }
        if (v_lane && half == 0) {
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
                d.mv[(gp * 3 + dd) * Ov + o] = av[dd] * invk;
                d.mvn[(gp * 3 + dd) * Ov + o] = avn[dd] * invk;
            }
        }
    }
    // ---- batch statistics: combine the workgroup's waves in LDS before touching the (grid-shared) global sums
    __shared__ double red_s[2 * 64 + 2 * 64];
    if (d.stat_y) {
        for (int i = threadIdx.x; i < 2 * Os + 2 * Ov; i += blockDim.x) red_s[i] = 0.0;
        __syncthreads();
        if (p_begin < p_end) {
            if (o_lane) { atomicAdd(&red_s[o], sy1); atomicAdd(&red_s[Os + o], sy2); }
            if (v_lane) { atomicAdd(&red_s[2 * Os + o], sv1); atomicAdd(&red_s[2 * Os + Ov + o], sv2); }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * Os + 2 * Ov; i += blockDim.x) {
            const double v = red_s[i];
            const int sl = (int)(blockIdx.x & (SVNET_RED_SLICES - 1));
            if (v != 0.0) atomicAdd(i < 2 * Os ? &d.stat_y[sl * 2 * Os + i] : &d.stat_v[sl * 2 * Ov + i - 2 * Os], v);
        }
    }
    if (p_begin < p_end) {
        float val = 0.f;
#pragma unroll
        for (int f = 0; f < NG; ++f) {
            const float tot = wave_sum(gsum[f]);
            val = (lane == f) ? tot : val;
        }
        if (lane < NG) atomicAdd(&d.gate_sum[b * NG + lane], (double)val);      // fp64: order-independent to fp32 precision
    }
}

// coef = [A1 | B1 | mean_y | invstd_y (Os each) | Av | Bv | mean_n' | invstd_n' (Ov each)]  (same layout as edgeblock)
struct XyzCoefArgs {
    const double* stat_y; const double* stat_v; int64_t E; int Os, Ov;
    const float* g1; const float* b1; float* rm1; float* rv1; const float* g2; const float* b2; float* rm2; float* rv2;
    int training; float eps, momentum;
};
// channel c of both coefficient sets into `out` (global memory or a workgroup's LDS copy); commit: this caller also updates the running
// statistics.  One body for the coefficient kernel and the tail kernel.
__device__ __forceinline__ void xyz_coefs_channel(const XyzCoefArgs& a, int c, bool commit, float* coef) {
    const int Os = a.Os, Ov = a.Ov;
    const int64_t E = a.E;
    for (int part = 0; part < 2; ++part) {
        const int C = part == 0 ? Os : Ov;
        if (c >= C) continue;
        const double* st = part == 0 ? a.stat_y : a.stat_v;
        const float* g = part == 0 ? a.g1 : a.g2;
        const float* bb = part == 0 ? a.b1 : a.b2;
        float* rm = part == 0 ? a.rm1 : a.rm2;
        float* rv = part == 0 ? a.rv1 : a.rv2;
        float* out = part == 0 ? coef : coef + 4 * Os;
        float mean, invstd;
        if (a.training) {
            double s1 = 0.0, s2 = 0.0;                      // the forward kernel's slices, added in a fixed order
            for (int sl = 0; sl < SVNET_RED_SLICES; ++sl) { s1 += st[sl * 2 * C + c]; s2 += st[sl * 2 * C + C + c]; }
            const double m = s1 / (double)E;
            double var = s2 / (double)E - m * m;
            if (var < 0.0) var = 0.0;
            mean = (float)m;
            invstd = (float)(1.0 / sqrt(var + (double)a.eps));
            if (commit && rm) rm[c] = (1.f - a.momentum) * rm[c] + a.momentum * mean;
            if (commit && rv) rv[c] = (1.f - a.momentum) * rv[c] + a.momentum * (float)(E > 1 ? var * ((double)E / (double)(E - 1)) : var);
        } else {
            mean = rm[c];
            invstd = 1.f / sqrtf(rv[c] + a.eps);
        }
        out[c] = g[c] * invstd;
        out[C + c] = bb[c] - g[c] * mean * invstd;
        out[2 * C + c] = mean;
        out[3 * C + c] = invstd;
    }
}

__global__ void xyzblock_coeffs_kernel(XyzCoefArgs a, float* __restrict__ coef, long long* __restrict__ nbt1, long long* __restrict__ nbt2,
                                       svnet_gate_fwd_job job, int coef_blocks) {
    if ((int)blockIdx.x >= coef_blocks) { svnet_gate_fwd_block(job, (int)blockIdx.x - coef_blocks); return; }   // the gate MLP beside the coefficients
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && a.training) {
        if (nbt1) *nbt1 += 1;
        if (nbt2) *nbt2 += 1;
    }
    xyz_coefs_channel(a, c, true, coef);
}

// (one functor for both apply kernels below: bit-identical outputs)
struct XyzApplyMath {
    const float* __restrict__ y_max; const float* __restrict__ y_min;
    const float* __restrict__ mv; const float* __restrict__ mvn;
    const float* __restrict__ A1; const float* __restrict__ B1; const float* __restrict__ Av; const float* __restrict__ Bv;
    const float* __restrict__ gate;
    int Os, Ov;
    float slope;
    __device__ __forceinline__ float s(int64_t p, int o) const {
        const float a = A1[o];
        const float y = a * (a >= 0.f ? y_max[p * Os + o] : y_min[p * Os + o]) + B1[o];
        return y > 0.f ? y : y * slope;
    }
    __device__ __forceinline__ float v(int64_t p, int64_t b, int q, int c) const {
        const int64_t e = p * 3 * Ov + q;
        // (the rounding sequence is pinned - gate * fma(Bv, mvn, Av * mv), what the compiler made of this line in rounds 1-3: the STRICT
        //  parity case pseg_fp_b32 sits on an unreplayed knife edge - a ReLU kink of the reference's own arithmetic - and a 1-ulp change of
        //  58 % of these outputs moves its worst gradient error from 9e-5 to 2e-3; the other contraction and the uncontracted torch order both do)
        return __fmul_rn(gate[b * Ov + c], __fmaf_rn(Bv[c], mvn[e], __fmul_rn(Av[c], mv[e])));
    }
};

__global__ __launch_bounds__(256) void xyzblock_apply_kernel(const float* __restrict__ y_max, const float* __restrict__ y_min,
                                                             const float* __restrict__ mv, const float* __restrict__ mvn,
                                                             const float* __restrict__ coef, const float* __restrict__ gate,
                                                             int64_t P, int64_t N, int Os, int Ov, float slope,
                                                             float* __restrict__ s_out, float* __restrict__ v_out, float* __restrict__ s_cat,
                                                             int64_t s_ld, float* __restrict__ v_cat, int64_t v_ld) {
    const XyzApplyMath m = {y_max, y_min, mv, mvn, coef, coef + Os, coef + 4 * Os, coef + 4 * Os + Ov, gate, Os, Ov, slope};
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
__global__ __launch_bounds__(256) void xyzblock_apply_knn_kernel(const float* __restrict__ y_max, const float* __restrict__ y_min,
                                                                 const float* __restrict__ mv, const float* __restrict__ mvn,
                                                                 const float* __restrict__ coef, const float* __restrict__ gate,
                                                                 int64_t P, int64_t N, int Os, int Ov, float slope,
                                                                 float* __restrict__ s_out, float* __restrict__ v_out,
                                                                 float* __restrict__ s_cat, int64_t s_ld, float* __restrict__ v_cat,
                                                                 int64_t v_ld, float* __restrict__ xT, float* __restrict__ xx, int64_t Cpad) {
    extern __shared__ float apply_knn_rows[];
    const XyzApplyMath m = {y_max, y_min, mv, mvn, coef, coef + Os, coef + 4 * Os, coef + 4 * Os + Ov, gate, Os, Ov, slope};
    apply_knn_tiles<APPLY_KNN_TP>(m, P, N, Os, Ov, s_out, v_out, s_cat, s_ld, v_cat, v_ld, xT, xx, Cpad, apply_knn_rows);
}

// ---- coefficients + gate MLP + apply (+ the next k-NN's table) in one launch: as edgeblock_tail_kernel (edgeblock.hip)
__global__ __launch_bounds__(256) void xyzblock_tail_kernel(XyzCoefArgs ca, float* __restrict__ coef, long long* __restrict__ nbt1,
                                                            long long* __restrict__ nbt2, svnet_gate_fwd_job job,
                                                            const float* __restrict__ y_max, const float* __restrict__ y_min,
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
    xyz_coefs_channel(ca, (int)threadIdx.x, first, tail_lds);
    svnet_gate_fwd_block(job, (int)b);
    __syncthreads();                                                     // the coefficients in LDS, the cloud's gate in global memory
    if (first)
        for (int i = threadIdx.x; i < 4 * Os + 4 * Ov; i += blockDim.x) coef[i] = tail_lds[i];
    const XyzApplyMath m = {y_max, y_min, mv, mvn, tail_lds, tail_lds + Os, tail_lds + 4 * Os, tail_lds + 4 * Os + Ov, job.gate, Os, Ov, slope};
    apply_knn_tiles<APPLY_KNN_TP>(m, P, N, Os, Ov, s_out, v_out, s_cat, s_ld, v_cat, v_ld, xT, xx, Cpad, tail_lds + ncoef);
}

// ---------------------------------------------------------------------------------------------- backward
// prelude: gy = Gs*lrelu'(y*), red = [sum gy | sum gy*xhat*], dgate, redv = [sum Gv*gate*mv | sum Gv*gate*mvn]  (prelude.h)
__global__ __launch_bounds__(256) void xyzblock_bwd_prelude_kernel(
    const float* __restrict__ gs, const float* __restrict__ gv, const float* __restrict__ y_max, const float* __restrict__ y_min,
    const float* __restrict__ mv, const float* __restrict__ mvn, const float* __restrict__ coef, const float* __restrict__ gate, int64_t P,
    int64_t N, int Os, int Ov, float slope, int64_t rows_per_block, float* __restrict__ gy, float* __restrict__ red,
    float* __restrict__ redv, float* __restrict__ dgate, const float* __restrict__ gs2, int64_t gs2_ld, const float* __restrict__ gv2,
    int64_t gv2_ld, float* __restrict__ gv_sum) {
    svnet_prelude_body<float>(gs, gv, y_max, y_min, mv, mvn, coef, nullptr, gate, P, N, Os, Ov, slope, rows_per_block, gy, red, redv, dgate, gs2,
                              gs2_ld, gv2, gv2_ld, gv_sum);
}

// edge pass: parameter gradients only.  gw layout: [W1 (Os*6NC) | W2 (Ov*NC) | W0 (3NC) | Wz (3NC)], accumulated with atomics into the
// slices of a sliced accumulator (SVNET_SLICED_LEN).
// Same two phases per point as the forward: the edge's features, v_e and Q_e = v_e^T v_e by the lane that owns the edge, then
// lanes = output channels over the point's edges (EPI = 2 edges per iteration for <= 32 channels).
template <int NC, int EPI>
__global__ __launch_bounds__(256) void xyzblock_bwd_kernel(svnet_xyzblock_bwd_desc d, int waves_per_cloud, int points_per_wave) {
    using FR = FeatRow<NC, true>;
    constexpr int NF = 6 * NC, NG = 3 * NC, NV = 3 * NC, NE = FR::NE, NP = FR::NP, HL = 64 / EPI;
    __shared__ __attribute__((aligned(16))) float feat_s[4][64 * NP];
    const int lane = threadIdx.x & 63;
    const int wave_l = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* feat = feat_s[wave_l];
    const int64_t wave_g = (int64_t)blockIdx.x * 4 + wave_l;
    const int64_t bq = wave_g / waves_per_cloud;
    const int64_t b = bq < d.B ? bq : d.B - 1;
    const int wi = (int)(wave_g - bq * waves_per_cloud);
    const int p_begin = wi * points_per_wave;
    const int p_end = (bq < d.B) ? min((int)d.N, p_begin + points_per_wave) : p_begin;   // idle waves still reach the barriers
    const bool live = p_begin < p_end;
    const int Os = d.Os, Ov = d.Ov, k = (int)d.k;
    const int64_t N = d.N;
    const float* xb = d.x + b * 3 * N;

    float w0[3][NC], wz[3][NC];
    load_small<NC>(d.w0, w0);
    load_small<NC>(d.wz, wz);
    const int o = lane & (HL - 1), half = lane / HL;
    const bool o_lane = o < Os, v_lane = o < Ov;
    const int lo = min(o, Os - 1), lv = min(o, Ov - 1);
    float w1[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) w1[f] = o_lane ? d.w1[o * NF + f] : 0.f;
    float w2[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) w2[c] = v_lane ? d.w2[o * NC + c] : 0.f;

    const float* coef = d.coef;
    const float a1 = coef[lo], my = coef[2 * Os + lo], iy = coef[3 * Os + lo];
    const float avc = coef[4 * Os + lv], bvc = coef[4 * Os + Ov + lv];
    const float m1 = d.bcoef[lo], m2 = d.bcoef[Os + lo], cs = d.bcoef[2 * Os + lo];
    const float c0 = d.bcoef[3 * Os + lv], c1 = d.bcoef[3 * Os + Ov + lv];
    const float invk = 1.f / (float)k;
    const float gt = d.gate[b * Ov + lv] * invk;
    float gc[NG];
#pragma unroll
    for (int f = 0; f < NG; ++f) gc[f] = d.gconst[b * NG + f];
    const uint8_t* slot_tab = (a1 >= 0.f) ? d.slot_max : d.slot_min;

    float gw1[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) gw1[f] = 0.f;
    float gw2[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) gw2[c] = 0.f;
    // The v2s weight gradients are linear in dL/dfeature = sum_o dyp[o] W1[o][:] (+ gate constants), so instead of a
    // wave reduction per edge each lane keeps  M[c2][c] = sum_e dyp[e,o] * Q_e[c2][c],  Q_e = sum_d ve[d][c2] ve[d][c],
    // and the reduction over the output channels happens ONCE per wave at the end.
    float mm[NE], qq[NE];                      // M per lane (output channel o); Q per EDGE lane (summed over the wave at the end); upper triangles
#pragma unroll
    for (int i = 0; i < NE; ++i) { mm[i] = 0.f; qq[i] = 0.f; }

    const int lk = min(lane, k - 1);
    const bool e_lane = lane < k;
    const int p_last = max(p_end - 1, 0);
    const int64_t gp0 = b * N;
    int jn1 = (int)d.idx[(gp0 + min(p_begin + 1, p_last)) * k + lk];         // ids of point p + 1
    float xjn[3];
    {
        const int j0 = (int)d.idx[(gp0 + min(p_begin, p_last)) * k + lk];
        xjn[0] = xb[j0]; xjn[1] = xb[N + j0]; xjn[2] = xb[2 * N + j0];
    }
    // per-point operands of the channel lanes, one point ahead
    int slot_n; float gy_n, gv_n[3];
#define SVNET_XYZ_LOAD_POINT(PP)                                                                            \
    do {                                                                                                    \
        const int64_t gq_ = gp0 + (PP);                                                                     \
        slot_n = slot_tab[gq_ * Os + lo];                                                                   \
        gy_n = d.gy[gq_ * Os + lo];                                                                         \
        gv_n[0] = d.gv[(gq_ * 3 + 0) * Ov + lv]; gv_n[1] = d.gv[(gq_ * 3 + 1) * Ov + lv]; gv_n[2] = d.gv[(gq_ * 3 + 2) * Ov + lv]; \
    } while (0)
    SVNET_XYZ_LOAD_POINT(min(p_begin, p_last));
    for (int p = p_begin; p < p_end; ++p) {
        const float xj[3] = {xjn[0], xjn[1], xjn[2]};
        const int slot = slot_n;
        const float gyv = gy_n, gv0 = gv_n[0] * gt, gv1 = gv_n[1] * gt, gv2 = gv_n[2] * gt;
        {   // unconditional requests from clamped addresses
            const int jq = jn1;
            jn1 = (int)d.idx[(gp0 + min(p + 2, p_last)) * k + lk];
            xjn[0] = xb[jq]; xjn[1] = xb[N + jq]; xjn[2] = xb[2 * N + jq];
            SVNET_XYZ_LOAD_POINT(min(p + 1, p_last));
        }
        const float xi[3] = {xb[p], xb[N + p], xb[2 * N + p]};
        // ---- phase 1: lane = edge
        {
            EdgeFeat<NC> e;
            edge_features<NC>(xi, xj, w0, wz, e);
            float r[NP];
#pragma unroll
            for (int f = 0; f < NF; ++f) r[f] = e.f[f];
#pragma unroll
            for (int dd = 0; dd < 3; ++dd)
#pragma unroll
                for (int c = 0; c < NC; ++c) r[NF + dd * NC + c] = e.ve[dd][c];
            int q = 0;
#pragma unroll
            for (int c2 = 0; c2 < NC; ++c2)
#pragma unroll
                for (int c = c2; c < NC; ++c) {
                    r[NF + NV + q] = e.ve[0][c2] * e.ve[0][c] + e.ve[1][c2] * e.ve[1][c] + e.ve[2][c2] * e.ve[2][c];
                    ++q;
                }
#pragma unroll
            for (int i = NF + NV + NE; i < NP; ++i) r[i] = 0.f;
            if (e_lane) {
#pragma unroll
                for (int i = 0; i < NE; ++i) qq[i] += r[NF + NV + i];
                float4* row = reinterpret_cast<float4*>(feat + lane * NP);
#pragma unroll
                for (int i = 0; i < NP / 4; ++i) row[i] = make_float4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- phase 2: lane = output channel
        for (int t0 = 0; t0 < k; t0 += EPI) {
            const int t = t0 + half;
            const bool valid = t < k;
            const float4* row = reinterpret_cast<const float4*>(feat + min(t, k - 1) * NP);
            float r[NP];
#pragma unroll
            for (int i = 0; i < NP / 4; ++i) { const float4 q4 = row[i]; r[4 * i] = q4.x; r[4 * i + 1] = q4.y; r[4 * i + 2] = q4.z; r[4 * i + 3] = q4.w; }
            // ---- scalar path
            float y = 0.f;
#pragma unroll
            for (int f = 0; f < NF; ++f) y = fmaf(w1[f], r[f], y);
            const float g = (slot == t) ? gyv : 0.f;
            const float xh = (y - my) * iy;
            const float dyp = (o_lane && valid) ? cs * (g - m1 - xh * m2) : 0.f;
#pragma unroll
            for (int f = 0; f < NF; ++f) gw1[f] = fmaf(dyp, r[f], gw1[f]);
#pragma unroll
            for (int i = 0; i < NE; ++i) mm[i] = fmaf(dyp, r[NF + NV + i], mm[i]);
            // ---- vector path
            float vp[3];
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
                float acc = w2[0] * r[NF + dd * NC];
#pragma unroll
                for (int c = 1; c < NC; ++c) acc += w2[c] * r[NF + dd * NC + c];
                vp[dd] = acc;
            }
            const float nv = fast_sqrt(vp[0] * vp[0] + vp[1] * vp[1] + vp[2] * vp[2]);
            const float nn = nv + VEPS;
            const float rn = fast_rcp(nn);
            const float qv = avc + bvc * rn;
            const float gdot = gv0 * vp[0] + gv1 * vp[1] + gv2 * vp[2];
            const float dnn = -gdot * bvc * rn * rn + c0 + c1 * nn;
            const float kk = nv > 0.f ? dnn * fast_rcp(nv) : 0.f;
            const float gvv[3] = {gv0, gv1, gv2};
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
                const float dvp = (v_lane && valid) ? (gvv[dd] * qv + kk * vp[dd]) : 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c) gw2[c] = fmaf(dvp, r[NF + dd * NC + c], gw2[c]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#undef SVNET_XYZ_LOAD_POINT
    // ---- every wave of the grid adds into the same few hundred addresses: combine the workgroup's four waves in LDS
    // first and issue one set of global atomics per workgroup (same-address float atomics serialise at the memory side)
    __shared__ float red[64 * NF + 64 * NC + NF];
    const int GW = Os * NF + Ov * NC + NF;
    for (int i = threadIdx.x; i < GW; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    if (live) {
        if (o_lane) {
#pragma unroll
            for (int f = 0; f < NF; ++f) atomicAdd(&red[o * NF + f], gw1[f]);
        }
        if (v_lane) {
#pragma unroll
            for (int c = 0; c < NC; ++c) atomicAdd(&red[Os * NF + o * NC + c], gw2[c]);
        }
        // dW[jz][c] = sum_c2 ( sum_o W1[o][h*3NC + c2*3 + jz] * M_o[c2][c]  +  gc[c2*3+jz] * Qsum[c2][c] (frame 0 only) ),  M, Q symmetric
        float mf[NC][NC], qf[NC][NC];
        {
            int q = 0;
#pragma unroll
            for (int c2 = 0; c2 < NC; ++c2)
#pragma unroll
                for (int c = c2; c < NC; ++c) {
                    const float qt = wave_sum(qq[q]);          // the edge lanes' partial sums -> the wave's total (uniform)
                    mf[c2][c] = mm[q]; mf[c][c2] = mm[q];
                    qf[c2][c] = qt; qf[c][c2] = qt;
                    ++q;
                }
        }
        float mine = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jz = 0; jz < 3; ++jz)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float part = 0.f;
#pragma unroll
                    for (int c2 = 0; c2 < NC; ++c2) part += w1[h * NG + c2 * 3 + jz] * mf[c2][c];
                    float tot = wave_sum(part);
                    if (h == 0) {
#pragma unroll
                        for (int c2 = 0; c2 < NC; ++c2) tot += gc[c2 * 3 + jz] * qf[c2][c];
                    }
                    mine = (lane == h * NG + jz * NC + c) ? tot : mine;
                }
        if (lane < NF) atomicAdd(&red[Os * NF + Ov * NC + lane], mine);
    }
    __syncthreads();
    // (gw: a SLICED accumulator, SVNET_SLICED_LEN(GW): two thousand workgroups adding to the same ~400 addresses are served one after the
    //  other at the memory side; the caller adds the slices up, svnet_slices_sum_f32)
    float* gsl = svnet_slice_ptr(d.gw, GW);
    for (int i = threadIdx.x; i < GW; i += blockDim.x) {
        const float v = red[i];
        if (v != 0.f) atomicAdd(&gsl[i], v);
    }
}

inline void wave_geometry(int64_t B, int64_t N, int& wpc, int& ppw) {
    wpc = (int)svnet_cdiv(8192, B);
    if (wpc > N) wpc = (int)N;
    if (wpc < 1) wpc = 1;
    ppw = (int)svnet_cdiv(N, wpc);
    wpc = (int)svnet_cdiv(N, ppw);
}

}  // namespace

extern "C" int svnet_xyzblock_fwd_f32(const svnet_xyzblock_desc* desc, void* stream) {
    SVNET_REQUIRE(desc, SVNET_E_ARG, "svnet_xyzblock_fwd_f32: null descriptor");
    const svnet_xyzblock_desc& d = *desc;
    SVNET_REQUIRE(d.x && d.idx && d.w0 && d.wz && d.w1 && d.w2 && d.y_max && d.y_min && d.slot_max && d.slot_min && d.mv && d.mvn &&
                      d.gate_sum, SVNET_E_ARG, "svnet_xyzblock_fwd_f32: null pointer");
    SVNET_REQUIRE((d.stat_y == nullptr) == (d.stat_v == nullptr), SVNET_E_ARG, "svnet_xyzblock_fwd_f32: pass both stat buffers or none");
    SVNET_REQUIRE(d.B >= 0 && d.N > 0 && d.k > 0 && d.k <= 64, SVNET_E_ARG, "svnet_xyzblock_fwd_f32: bad sizes (k <= 64)");
    SVNET_REQUIRE(d.Os > 0 && d.Os <= 64 && d.Ov > 0 && d.Ov <= 64, SVNET_E_UNSUPPORTED, "svnet_xyzblock_fwd_f32: needs Os <= 64, Ov <= 64");
    if (d.B == 0) return SVNET_OK;
    XyzFwdArgs fa;
    fa.d = d;
    wave_geometry(d.B, d.N, fa.waves_per_cloud, fa.points_per_wave);
    const unsigned grid = (unsigned)svnet_cdiv(d.B * fa.waves_per_cloud, 4);
    SVNET_REQUIRE(d.nc == 0 || d.nc == 2 || d.nc == 3, SVNET_E_UNSUPPORTED, "svnet_xyzblock_fwd_f32: nc must be 2 (plain) or 3 (cross)");
    const bool two = d.Os <= 32 && d.Ov <= 32;                  // two edges per wave iteration
    if (d.nc == 3) {
        if (two) hipLaunchKernelGGL((xyzblock_fwd_kernel<3, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, fa);
        else hipLaunchKernelGGL((xyzblock_fwd_kernel<3, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, fa);
    } else {
        if (two) hipLaunchKernelGGL((xyzblock_fwd_kernel<2, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, fa);
        else hipLaunchKernelGGL((xyzblock_fwd_kernel<2, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, fa);
    }
    SVNET_CHECK_LAUNCH("xyzblock_fwd_kernel");
    return SVNET_OK;
}

extern "C" int svnet_xyzblock_coeffs_f32(const double* stat_y, const double* stat_v, int64_t E, int64_t Os, int64_t Ov,
                                         const float* gamma1, const float* beta1, float* running_mean1, float* running_var1,
                                         const float* gamma2, const float* beta2, float* running_mean2, float* running_var2,
                                         int training, float eps, float momentum, float* coef, int64_t* num_batches_tracked1,
                                         int64_t* num_batches_tracked2, const svnet_gate_fwd_job* gate_job, void* stream) {
    SVNET_REQUIRE(gamma1 && beta1 && gamma2 && beta2 && coef && E > 0 && Os > 0 && Ov > 0, SVNET_E_ARG, "svnet_xyzblock_coeffs_f32: bad arguments");
    SVNET_REQUIRE(training ? (stat_y && stat_v) : (running_mean1 && running_var1 && running_mean2 && running_var2), SVNET_E_ARG,
                  "svnet_xyzblock_coeffs_f32: missing statistics");
    const int64_t n = Os > Ov ? Os : Ov;
    SVNET_REQUIRE(!gate_job || svnet_gate_fwd_job_ok(gate_job), SVNET_E_ARG, "svnet_xyzblock_coeffs_f32: bad gate job");
    const int coef_blocks = (int)svnet_cdiv(n, 256);
    const svnet_gate_fwd_job job = gate_job ? *gate_job : svnet_gate_fwd_job{};
    const XyzCoefArgs ca = {stat_y, stat_v, E, (int)Os, (int)Ov, gamma1, beta1, running_mean1, running_var1, gamma2, beta2, running_mean2,
                            running_var2, training, eps, momentum};
    hipLaunchKernelGGL(xyzblock_coeffs_kernel, dim3((unsigned)(coef_blocks + (gate_job ? gate_job->B : 0))), dim3(256), 0, (hipStream_t)stream, ca,
                       coef, reinterpret_cast<long long*>(num_batches_tracked1), reinterpret_cast<long long*>(num_batches_tracked2), job, coef_blocks);
    SVNET_CHECK_LAUNCH("xyzblock_coeffs_kernel");
    return SVNET_OK;
}

extern "C" int svnet_xyzblock_apply_f32(const float* y_max, const float* y_min, const float* mv, const float* mvn, const float* coef,
                                        const float* gate, int64_t P, int64_t N, int64_t Os, int64_t Ov, float slope, float* s_out,
                                        float* v_out, float* s_cat, int64_t s_ld, float* v_cat, int64_t v_ld, void* stream) {
    SVNET_REQUIRE(y_max && y_min && mv && mvn && coef && gate && s_out && v_out && P >= 0 && N > 0, SVNET_E_ARG, "svnet_xyzblock_apply_f32: bad arguments");
    SVNET_REQUIRE((!s_cat || s_ld >= Os) && (!v_cat || v_ld >= Ov), SVNET_E_ARG, "svnet_xyzblock_apply_f32: concatenation row shorter than the slice");
    if (P == 0) return SVNET_OK;
    hipLaunchKernelGGL(xyzblock_apply_kernel, dim3(svnet_grid(P * 64, 256, 256 * 8)), dim3(256), 0, (hipStream_t)stream, y_max, y_min, mv,
                       mvn, coef, gate, P, N, (int)Os, (int)Ov, slope, s_out, v_out, s_cat, s_ld, v_cat, v_ld);
    SVNET_CHECK_LAUNCH("xyzblock_apply_kernel");
    return SVNET_OK;
}

extern "C" int svnet_xyzblock_apply_knn_f32(const float* y_max, const float* y_min, const float* mv, const float* mvn, const float* coef,
                                            const float* gate, int64_t P, int64_t N, int64_t Os, int64_t Ov, float slope, float* s_out,
                                            float* v_out, float* s_cat, int64_t s_ld, float* v_cat, int64_t v_ld, void* knn_workspace,
                                            size_t knn_workspace_bytes, void* stream) {
    SVNET_REQUIRE(y_max && y_min && mv && mvn && coef && gate && s_out && v_out && knn_workspace && P > 0 && N > 0 && P % N == 0, SVNET_E_ARG,
                  "svnet_xyzblock_apply_knn_f32: bad arguments");
    SVNET_REQUIRE((!s_cat || s_ld >= Os) && (!v_cat || v_ld >= Ov), SVNET_E_ARG, "svnet_xyzblock_apply_knn_f32: concatenation row shorter than the slice");
    int64_t Cpad = 0;
    SVNET_REQUIRE(apply_knn_supported(P, N, Os, Ov, &Cpad), SVNET_E_UNSUPPORTED,
                  "svnet_xyzblock_apply_knn_f32: N=%lld, Os=%lld, Ov=%lld not supported (ask svnet_knn_table_fusable first)", (long long)N,
                  (long long)Os, (long long)Ov);
    SVNET_REQUIRE(knn_workspace_bytes >= svnet_knn_workspace_bytes(P / N, N, Os + 3 * Ov), SVNET_E_WORKSPACE,
                  "svnet_xyzblock_apply_knn_f32: k-NN workspace too small");
    float* xT = (float*)knn_workspace;
    float* xx = xT + P * ((Os + 3 * Ov + 7) / 8 * 8);
    hipLaunchKernelGGL(xyzblock_apply_knn_kernel, dim3((unsigned)(P / APPLY_KNN_TP)), dim3(256), apply_knn_lds_bytes(Os, Ov), (hipStream_t)stream,
                       y_max, y_min, mv, mvn, coef, gate, P, N, (int)Os, (int)Ov, slope, s_out, v_out, s_cat, s_ld, v_cat, v_ld, xT, xx, Cpad);
    SVNET_CHECK_LAUNCH("xyzblock_apply_knn_kernel");
    return SVNET_OK;
}

extern "C" int svnet_xyzblock_tail_f32(const svnet_block_tail_desc* desc, void* stream) {
    SVNET_REQUIRE(desc, SVNET_E_ARG, "svnet_xyzblock_tail_f32: null descriptor");
    const svnet_block_tail_desc& d = *desc;
    const char* who = "svnet_xyzblock_tail_f32";
    SVNET_REQUIRE(d.hi && d.lo && d.mv && d.mvn && d.coef && d.s_out && d.v_out && d.gamma1 && d.beta1 && d.gamma2 && d.beta2, SVNET_E_ARG,
                  "%s: null pointer", who);
    SVNET_REQUIRE(d.training ? (d.stat1 && d.stat_v) : (d.running_mean1 && d.running_var1 && d.running_mean2 && d.running_var2), SVNET_E_ARG,
                  "%s: missing statistics", who);
    SVNET_REQUIRE(svnet_gate_fwd_job_ok(&d.gate) && d.gate.Ov == d.Ov && d.gate.B * d.N == d.P, SVNET_E_ARG, "%s: bad gate job", who);
    SVNET_REQUIRE((!d.s_cat || d.s_ld >= d.Os) && (!d.v_cat || d.v_ld >= d.Ov), SVNET_E_ARG, "%s: concatenation row shorter than the slice", who);
    SVNET_REQUIRE(svnet_block_tail_supported(d.P, d.N, d.Os, d.Ov, d.knn_workspace != nullptr), SVNET_E_UNSUPPORTED,
                  "%s: P=%lld N=%lld Os=%lld Ov=%lld not supported (svnet_block_tail_supported)", who, (long long)d.P, (long long)d.N,
                  (long long)d.Os, (long long)d.Ov);
    float* xT = nullptr; float* xx = nullptr; int64_t Cpad = 0;
    if (d.knn_workspace) {
        SVNET_REQUIRE(d.knn_workspace_bytes >= svnet_knn_workspace_bytes(d.P / d.N, d.N, d.Os + 3 * d.Ov), SVNET_E_WORKSPACE,
                      "%s: k-NN workspace too small", who);
        apply_knn_supported(d.P, d.N, d.Os, d.Ov, &Cpad);
        xT = (float*)d.knn_workspace;
        xx = xT + d.P * ((d.Os + 3 * d.Ov + 7) / 8 * 8);
    }
    const size_t lds = (size_t)((4 * d.Os + 4 * d.Ov + 3) & ~(int64_t)3) * sizeof(float) + apply_knn_lds_bytes(d.Os, d.Ov);
    const XyzCoefArgs ca = {reinterpret_cast<const double*>(d.stat1), d.stat_v, d.E, (int)d.Os, (int)d.Ov, d.gamma1, d.beta1, d.running_mean1,
                            d.running_var1, d.gamma2, d.beta2, d.running_mean2, d.running_var2, d.training, d.eps, d.momentum};
    hipLaunchKernelGGL(xyzblock_tail_kernel, dim3((unsigned)(d.P / APPLY_KNN_TP)), dim3(256), lds, (hipStream_t)stream, ca, d.coef,
                       reinterpret_cast<long long*>(d.num_batches_tracked1), reinterpret_cast<long long*>(d.num_batches_tracked2), d.gate,
                       (const float*)d.hi, (const float*)d.lo, d.mv, d.mvn, d.P, d.N, d.slope, d.s_out, d.v_out, d.s_cat, d.s_ld, d.v_cat,
                       d.v_ld, xT, xx, Cpad);
    SVNET_CHECK_LAUNCH("xyzblock_tail_kernel");
    return SVNET_OK;
}

extern "C" int svnet_xyzblock_bwd_prelude_f32(const float* gs, const float* gv, const float* y_max, const float* y_min, const float* mv,
                                              const float* mvn, const float* coef, const float* gate, int64_t P, int64_t N, int64_t Os,
                                              int64_t Ov, float slope, float* gy, float* red, float* redv, float* dgate, const float* gs2,
                                              int64_t gs2_ld, const float* gv2, int64_t gv2_ld, float* gv_sum, void* stream) {
    SVNET_REQUIRE((gs || gs2) && (gv || gv2) && y_max && y_min && mv && mvn && coef && gate && gy && red && redv && dgate, SVNET_E_ARG,
                  "svnet_xyzblock_bwd_prelude_f32: null pointer");
    SVNET_REQUIRE((!gs2 || gs2_ld >= Os) && (!gv2 || (gv2_ld >= Ov && gv_sum)), SVNET_E_ARG,
                  "svnet_xyzblock_bwd_prelude_f32: second gradient source needs row strides >= the slice and gv_sum");
    SVNET_REQUIRE(P > 0 && N > 0 && Os > 0 && Ov > 0, SVNET_E_ARG, "svnet_xyzblock_bwd_prelude_f32: bad sizes");
    SVNET_REQUIRE(Os <= 128 && Ov <= 64, SVNET_E_UNSUPPORTED, "svnet_xyzblock_bwd_prelude_f32: Os <= 128, Ov <= 64");
    const int64_t rpb = svnet_prelude_rows(N);
    hipLaunchKernelGGL(xyzblock_bwd_prelude_kernel, dim3((unsigned)svnet_cdiv(P, rpb)), dim3(256), 0, (hipStream_t)stream, gs, gv, y_max,
                       y_min, mv, mvn, coef, gate, P, N, (int)Os, (int)Ov, slope, rpb, gy, red, redv, dgate, gs2, gs2_ld, gv2, gv2_ld, gv_sum);
    SVNET_CHECK_LAUNCH("xyzblock_bwd_prelude_kernel");
    return SVNET_OK;
}

extern "C" int svnet_xyzblock_bwd_f32(const svnet_xyzblock_bwd_desc* desc, void* stream) {
    SVNET_REQUIRE(desc, SVNET_E_ARG, "svnet_xyzblock_bwd_f32: null descriptor");
    const svnet_xyzblock_bwd_desc& d = *desc;
    SVNET_REQUIRE(d.x && d.idx && d.w0 && d.wz && d.w1 && d.w2 && d.slot_max && d.slot_min && d.coef && d.bcoef && d.gate && d.gy && d.gv &&
                      d.gconst && d.gw, SVNET_E_ARG, "svnet_xyzblock_bwd_f32: null pointer");
    SVNET_REQUIRE(d.B >= 0 && d.N > 0 && d.k > 0 && d.k <= 64, SVNET_E_ARG, "svnet_xyzblock_bwd_f32: bad sizes (k <= 64)");
    SVNET_REQUIRE(d.Os > 0 && d.Os <= 64 && d.Ov > 0 && d.Ov <= 64, SVNET_E_UNSUPPORTED, "svnet_xyzblock_bwd_f32: needs Os <= 64, Ov <= 64");
    if (d.B == 0) return SVNET_OK;
    int wpc = (int)svnet_cdiv(4096, d.B);                       // ~4096 waves = 1024 workgroups, one flush each
    if (wpc > d.N) wpc = (int)d.N;
    if (wpc < 1) wpc = 1;
    const int ppw = (int)svnet_cdiv(d.N, wpc);
    wpc = (int)svnet_cdiv(d.N, ppw);
    const unsigned grid = (unsigned)svnet_cdiv(d.B * wpc, 4);
    SVNET_REQUIRE(d.nc == 0 || d.nc == 2 || d.nc == 3, SVNET_E_UNSUPPORTED, "svnet_xyzblock_bwd_f32: nc must be 2 (plain) or 3 (cross)");
    const bool two = d.Os <= 32 && d.Ov <= 32;
    if (d.nc == 3) {
        if (two) hipLaunchKernelGGL((xyzblock_bwd_kernel<3, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, d, wpc, ppw);
        else hipLaunchKernelGGL((xyzblock_bwd_kernel<3, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, d, wpc, ppw);
    } else {
        if (two) hipLaunchKernelGGL((xyzblock_bwd_kernel<2, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, d, wpc, ppw);
        else hipLaunchKernelGGL((xyzblock_bwd_kernel<2, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, d, wpc, ppw);
    }
    SVNET_CHECK_LAUNCH("xyzblock_bwd_kernel");
    return SVNET_OK;
}
