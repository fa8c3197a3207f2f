// The vector path's linear layer of an SVBlock on rows, WITH the statistics of the VectorBN that follows it:
//   y[p, a, o] = scale[o] * sum_k v[p, a, k] * sign(W)[o, k]          (sv_layers.py:44-49 with bw only, :192)
//   sums[o] += n, sums[O + o] += n^2,  n = ||y[p, :, o]||_2 + 1e-6     (sv_layers.py:86-102, the batch statistics of bn2)
// The generic rows GEMM cannot form n in its epilogue: its 128-row tiles cut through the points (3 rows each), and a point's three
// rows land in different lanes / registers of the 32x32 accumulator tile.  Here a workgroup takes 32 POINTS at a time and the three
// MFMA row tiles of a wave are the three AXES of the same 32 points (A fragment of axis a, lane r: LDS row 3 r + a), so the three
// components of y[p, :, o] sit in the same lane and register index of three accumulators and n is formed in place - the separate
// statistics pass over y (svnet_colstats_f64 kind 1: 67 MB read at conv5 of the classifier, 17 launches a step in the PointNet callers)
// is gone.  K <= 96 (the whole reduction and the whole weight fit in LDS: no k pipeline - the layer is bound by its 3 P (K + O) floats of
// traffic, the next tile's rows travel in registers under the products), O <= 256.
// Arithmetic: the fp32 activations are split exactly into three bf16 pieces (x = h + m + l, as in gemm_mfma.hip), the +-1 / 0 weights are
// exact in bf16, fp32 accumulation - the rows kernels' recipe.  Measured: faster alone, not in the step - the product
// path uses it only with config.FUSE_VBN_STATS.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ __bf16 vl_bf16_from_bits(uint32_t b) { return __builtin_bit_cast(__bf16, (unsigned short)b); }

struct VlinArgs {
    const float* v; const float* wb; const float* cs; float* y; double* sums;
    int64_t P;
    int K, O, tiles;
};

constexpr int VL_PT = 32;                    // points per tile
constexpr int VL_ROWS = 3 * VL_PT;
constexpr float VL_EPS = 1e-6f;              // VectorBN's EPS (sv_layers.py:94)

// KP: K rounded up to 32, 64 or 96.  8 waves: wave w multiplies the 32 points x 3 axes with column tile w (waves past the last tile only
// help with the staging).  The fp32 rows are split into their three bf16 pieces ONCE, by the thread that stages them ([piece][96][KP + 8]
// bf16 in LDS; lane r reads row 3 r + axis, 16 bytes at a time: rows 52 words apart, 3 x 52 = 28 (mod 32) - conflict-free) - splitting in
// the multiplying waves (the rows kernels' way) did the same ~70 vector instructions per fragment six times over: 66 us at conv5.
template <int KP>
__global__ __launch_bounds__(512, 2) void vlinear_stats_kernel(VlinArgs a) {
    constexpr int BS = KP + 8;               // bf16 per LDS row (weight and activation pieces)
    constexpr int AP = VL_ROWS * BS;         // elements of one piece
    constexpr int NTH = 512;
    constexpr int NU = (VL_ROWS * KP + NTH - 1) / NTH;   // floats of a tile per thread (upper bound: K = KP)
    extern __shared__ __attribute__((aligned(16))) unsigned char vls[];
    uint16_t* As = reinterpret_cast<uint16_t*>(vls);                            // [3][96][BS]
    __bf16* Bs = reinterpret_cast<__bf16*>(vls + 3 * AP * sizeof(uint16_t));    // [32 ceil(O / 32)][BS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int K = a.K, O = a.O;
    const int nct = (O + 31) >> 5;                                              // 32-column tiles
    const int tile_floats = VL_ROWS * K;
    const float inv_k = 1.f / (float)K;

    // once per workgroup: the weight as bf16 [o][k] (zeros past O / K), the padding columns of the activation pieces zeroed
    for (int e = tid; e < 32 * nct * KP; e += NTH) {
        const int o = e / KP, k = e - o * KP;
        const float w = (o < O && k < K) ? a.wb[(size_t)o * K + k] : 0.f;
        Bs[o * BS + k] = vl_bf16_from_bits(__float_as_uint(w) >> 16);
    }
    for (int e = tid; e < 3 * AP; e += NTH) As[e] = 0;
    const int col = 32 * wave + r;
    const bool active = wave < nct;                                             // (wave-uniform)
    const float csv = (col < O) ? (a.cs ? a.cs[col] : 1.f) : 0.f;
    double s1 = 0.0, s2 = 0.0;

    float stg[NU];
    const int64_t total_floats = a.P * 3 * (int64_t)K;
#define SVNET_VL_FETCH(TILE)                                                                                    \
    do {                                                                                                        \
        const int64_t base_ = (int64_t)(TILE) * tile_floats;                                                     \
        _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                                         \
            const int i_ = tid + NTH * u;                                                                        \
            const int64_t g_ = min(base_ + i_, total_floats - 1);               /* (clamped: rows past P are masked below) */ \
            stg[u] = a.v[g_];                                                                                    \
        }                                                                                                       \
    } while (0)

    int tile = blockIdx.x;
    if (tile < a.tiles) SVNET_VL_FETCH(tile);
    __syncthreads();                                                            // the zero fill, the weight
    for (; tile < a.tiles; tile += gridDim.x) {
        const int64_t base = (int64_t)tile * tile_floats;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = tid + NTH * u;                                        // float i of the tile = (row i / K, column i % K)
            const int row = (int)(((float)i + 0.5f) * inv_k);                   // (i < 96 * 96: exact)
            const int k = i - row * K;
            const float x = (base + i < total_floats) ? stg[u] : 0.f;           // (rows past P: zeros)
            const uint32_t hu = __float_as_uint(x) & 0xFFFF0000u;               // x = h + m + l exactly (three bf16 pieces)
            const float r1 = x - __uint_as_float(hu);
            const uint32_t mu = __float_as_uint(r1) & 0xFFFF0000u;
            const float r2 = r1 - __uint_as_float(mu);
            if (i < tile_floats) {
                uint16_t* o_ = As + row * BS + k;
                o_[0] = (uint16_t)(hu >> 16);
                o_[AP] = (uint16_t)(mu >> 16);
                o_[2 * AP] = (uint16_t)(__float_as_uint(r2) >> 16);
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < a.tiles) SVNET_VL_FETCH(tile + gridDim.x);   // the next tile's rows, under this tile's products

        if (active) {
            f32x16 acc[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ax][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KP; ks += 16) {
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(Bs + (32 * wave + r) * BS + ks + 8 * h);
                bf16x8 fa[3][3];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax)
                        fa[pc][ax] = *reinterpret_cast<const bf16x8*>(As + pc * AP + (3 * r + ax) * BS + ks + 8 * h);
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)                                  // (piece outermost: three independent accumulators between dependent products)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) acc[ax] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pc][ax], fb, acc[ax], 0, 0, 0);
            }
            // epilogue: D register e of a tile = point (e & 3) + 8 (e >> 2) + 4 h of the 32, column r
            const int64_t p0 = (int64_t)tile * VL_PT;
            if (col < O) {
                float t1 = 0.f, t2 = 0.f;                                       // this tile's 16 points in fp32, then onto the fp64 sums
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t p = p0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (p < a.P) {
                        const float y0 = acc[0][e] * csv, y1 = acc[1][e] * csv, y2 = acc[2][e] * csv;
                        float* dst = a.y + (p * 3) * O + col;
                        __builtin_nontemporal_store(y0, dst);
                        __builtin_nontemporal_store(y1, dst + O);
                        __builtin_nontemporal_store(y2, dst + 2 * O);
                        const float n = sqrtf(y0 * y0 + y1 * y1 + y2 * y2) + VL_EPS;
                        t1 += n;
                        t2 += n * n;
                    }
                }
                s1 += (double)t1;
                s2 += (double)t2;
            }
        }
        __syncthreads();                                                        // the tile has been read
    }
#undef SVNET_VL_FETCH
    // this workgroup's share of the sums: one fp64 atomic per column and moment onto its slice (svnet_hip.h: sliced accumulators)
    double* sl = svnet_slice_ptr(a.sums, 2 * O);
    const double t1 = s1 + __shfl_xor(s1, 32, 64), t2 = s2 + __shfl_xor(s2, 32, 64);
    if (h == 0 && active && col < O) {
        atomicAdd(&sl[col], t1);
        atomicAdd(&sl[O + col], t2);
    }
}

template <int KP>
int vl_launch(const VlinArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(3 * VL_ROWS + 32 * ((a.O + 31) / 32)) * (KP + 8) * 2;
    bool ok = true;
    SVNET_LDS_OPTIN(ok, 128 * 1024, "vlinear_stats_kernel", (const void*)vlinear_stats_kernel<KP>);
    if (!ok) return SVNET_E_LAUNCH;
    static const int cus = [] { hipDeviceProp_t pr; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }();
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;                               // (160 KB of LDS per CU)
    int grid = cus * per_cu;                                                    // resident workgroups walk the tiles
    if (grid > a.tiles) grid = a.tiles;
    hipLaunchKernelGGL((vlinear_stats_kernel<KP>), dim3((unsigned)grid), dim3(512), lds, st, a);
    return 0;
}

}  // namespace

extern "C" int svnet_vlinear_stats_f32(const float* v, int64_t P, int64_t K, const float* w_b, const float* col_scale, int64_t O,
                                       float* y, double* sums, void* stream) {
    SVNET_REQUIRE(v && w_b && y && sums && P >= 0 && K > 0 && O > 0, SVNET_E_ARG, "svnet_vlinear_stats_f32: bad arguments");
    SVNET_REQUIRE(K <= 96 && O <= 256, SVNET_E_UNSUPPORTED, "svnet_vlinear_stats_f32: K=%lld O=%lld outside K <= 96, O <= 256", (long long)K,
                  (long long)O);
    SVNET_REQUIRE(P * 3 * K < ((int64_t)1 << 40), SVNET_E_UNSUPPORTED, "svnet_vlinear_stats_f32: too many rows");
    if (P == 0) return SVNET_OK;
    VlinArgs a = {v, w_b, col_scale, y, sums, P, (int)K, (int)O, (int)svnet_cdiv(P, VL_PT)};
    hipStream_t st = (hipStream_t)stream;
    if (K <= 32) vl_launch<32>(a, st);
    else if (K <= 64) vl_launch<64>(a, st);
    else vl_launch<96>(a, st);
    SVNET_CHECK_LAUNCH("vlinear_stats_kernel");
    return SVNET_OK;
}
