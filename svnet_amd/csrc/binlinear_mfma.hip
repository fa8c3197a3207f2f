// Binarized dense layer on the matrix cores, for many rows (conv5 of the classifiers: [32768, 505] x [505, 512]).
//
// Replaces  models/sv_layers.py:35-51 (Linear with bw and ba), like binlinear.hip:  y = (sign(x + beta) . sign(W)^T) * scale (+ b).
// The XNOR-popcount form costs ~10 vector instructions per 64-bit word and (row, channel) pair - 50 M wave-instructions on conv5,
// which made binlinear_fwd_kernel VALU-bound at 114 us.  Here the ternary operands are int8 (-1 / 0 / +1) and the products run on
// v_mfma_i32_32x32x32_i8 with exact int32 accumulation (|count| <= K): the same integer count, hence bit-identical outputs.
//   workgroup = 16 waves = 128 rows x 128 / 256 / 512 output channels; K in chunks of 128 columns;
//   stage A: coalesced fp32 row loads (lane q of a half-wave takes columns q, q+32, q+64, q+96 of the chunk), binarized; the four
//            int8 go to LDS as ONE dword at position 4q (the reduction index is permuted consistently: the packed weights use the
//            same order), and their sign / non-zero / STE bits come out of ballots as row-major plane words;
//   planes : (training) the row-major words are transposed 32 x 32 bits at a time (svnet_bit_transpose32) into the ROW-SLICED
//            planes the backward GEMMs read, written as 32-bit halves;
//   stage B: 16-byte copies of the pre-packed int8 weights (svnet_binweight_pack_i8);
//   MFMA   : A and B fragments are 16-byte LDS reads (lane (r, h): row / column r, positions 16h .. 16h+15 of the k-step).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;

constexpr int BM = 128;            // rows per workgroup
constexpr int KC = 128;            // columns per chunk
constexpr int LDA = KC + 16;       // bytes per LDS row (36 dwords: 16-byte reads of 16 consecutive rows hit 64 distinct banks)

// (position of column k = 32 j + q of a 128-column chunk in the permuted reduction order: 4 q + j)
// w_i8p [O][Kp] (Kp = 128 * ceil(K / 128)): sign(W) as int8, every 128-column chunk in the permuted order, zero padded
__global__ __launch_bounds__(256) void binweight_pack_i8_kernel(const float* __restrict__ W, int O, int K, int Kp, int8_t* __restrict__ out) {
    const int total = O * Kp;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int o = e / Kp, p = e - o * Kp;
        const int c = p >> 7, pin = p & 127;
        const int k = c * KC + 32 * (pin & 3) + (pin >> 2);
        const float w = k < K ? W[(int64_t)o * K + k] : 0.f;
        out[e] = (int8_t)((w > 0.f) - (w < 0.f));
    }
}

// Workgroup = 16 waves = 128 rows x (128 NTW) output channels: wave w multiplies rows 32 (w >> 2) .. +31 by columns
// 32 NTW (w & 3) .. of the tile (NTW accumulator tiles: 64 registers at NTW = 4) and binarizes rows 8 w .. 8 w + 7 of every chunk.
// ONE workgroup per CU at four waves per SIMD: the A tile of a row block is binarized once (the 4-wave / 256-column form did it once
// per column group, i.e. twice at O = 512), the next chunk's 16 values per lane are requested right after the barrier that frees the
// LDS tiles and fly across the plane transposition and the MFMAs of the chunk before (the old form waited for four dependent
// quarter-chunk round trips per chunk with nothing else to do: 92 us for conv5's 8.5 GOP, 3.7 % of the int8 matrix rate).
template <int NTW>
__global__ __launch_bounds__(1024, 4) void binlinear_i8_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ beta,
                                                                   const int8_t* __restrict__ w8, const float* __restrict__ scale,
                                                                   const float* __restrict__ bias, int64_t M, int K, int O, int Kp,
                                                                   float* __restrict__ y, uint32_t* __restrict__ x_sign32,
                                                                   uint32_t* __restrict__ x_nz32, uint32_t* __restrict__ x_ste32,
                                                                   double* __restrict__ col_sums, const float* __restrict__ cloud_n,
                                                                   int64_t rows_per_cloud) {
    constexpr int OT = 128 * NTW;                      // output channels per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char bl_lds[];     // [At | Bt | pw]: one array (cdna guide: no second __shared__ object)
    int8_t* At = reinterpret_cast<int8_t*>(bl_lds);                             // [BM][LDA]
    int8_t* Bt = At + BM * LDA;                                                 // [OT][LDA]
    uint64_t* pw = reinterpret_cast<uint64_t*>(Bt + OT * LDA);                  // [plane][row][word] of the current chunk, row-major bits (bit = column)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int o0 = blockIdx.y * OT;
    const bool save = x_sign32 != nullptr && blockIdx.y == 0;      // the column groups of a row tile share the planes: group 0 writes them

    i32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0;

    const int nchunks = Kp / KC;
    float xa[4][4];                                                 // the chunk's raw values of this lane: rows 8 wave + 2 rp + h, columns 32 j + r
    // (wave-uniform row base in SGPRs + 32-bit lane byte offsets: one add per load instead of a 64-bit address each; rows past M are
    //  clamped to the last row - never stored, never counted)
    const int64_t brow = min(m0 + 8 * wave, M - 1);
    const float* xbase = x + brow * ldx;
    uint32_t rowb[4];
#pragma unroll
    for (int rp = 0; rp < 4; ++rp) rowb[rp] = (uint32_t)(min(m0 + 8 * wave + h + 2 * rp, M - 1) - brow) * (uint32_t)ldx * 4u;
#define SVNET_BL_LOAD_A(K0)                                                                       \
    do {                                                                                          \
        uint32_t colb_[4];                                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) colb_[j] = 4u * (uint32_t)min((K0) + 32 * j + r, K - 1); \
        _Pragma("unroll") for (int rp = 0; rp < 4; ++rp)                                          \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) xa[rp][j] = ld_f32_sbase(xbase, rowb[rp] + colb_[j]); \
    } while (0)
    SVNET_BL_LOAD_A(0);
    for (int c = 0; c < nchunks; ++c) {
        const int k0 = c * KC;
        // ---- stage B (issued first: the loads fly while A is binarized): OT columns x 8 pieces of 16 bytes
        uint4 bstage[NTW];
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            const int e = tid + 1024 * u;
            const int n = e >> 3, pc = e & 7;
            const int o = min(o0 + n, O - 1);
            bstage[u] = *reinterpret_cast<const uint4*>(w8 + (int64_t)o * Kp + k0 + 16 * pc);
        }
        // ---- stage A: wave w binarizes rows 8w .. 8w+7, two rows per instruction (lanes 0-31 / 32-63)
        {
            float bt[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bt[j] = beta[min(k0 + 32 * j + r, K - 1)];
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                const int row = 8 * wave + 2 * rp + h;
                uint32_t pk = 0;
                uint64_t bs[4], bz[4], bq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool in = (k0 + 32 * j + r) < K && (m0 + row) < M;
                    const float tv = in ? xa[rp][j] + bt[j] : 0.f;
                    const bool pos = tv > 0.f, neg = tv < 0.f;
                    pk |= (pos ? 0x01u : (neg ? 0xFFu : 0u)) << (8 * j);
                    if (save) {                                      // (wave-uniform: only the plane-writing column group pays for the ballots)
                        bs[j] = __ballot(pos);
                        bz[j] = __ballot(pos || neg);
                        bq[j] = __ballot(in && fabsf(tv) <= 1.2f);
                    } else {
                        bs[j] = bz[j] = bq[j] = 0ull;
                    }
                }
                *reinterpret_cast<uint32_t*>(&At[row * LDA + 4 * r]) = pk;
                if (save && lane == 0) {
                    // ballot j: bits 0-31 = row (2 rp) columns 32 j .. 32 j + 31, bits 32-63 = row (2 rp + 1)
                    const int ra = 8 * wave + 2 * rp;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int sh = 32 * hh;
                        pw[(0 * BM + ra + hh) * 2 + 0] = ((bs[0] >> sh) & 0xFFFFFFFFull) | (((bs[1] >> sh) & 0xFFFFFFFFull) << 32);
                        pw[(0 * BM + ra + hh) * 2 + 1] = ((bs[2] >> sh) & 0xFFFFFFFFull) | (((bs[3] >> sh) & 0xFFFFFFFFull) << 32);
                        pw[(1 * BM + ra + hh) * 2 + 0] = ((bz[0] >> sh) & 0xFFFFFFFFull) | (((bz[1] >> sh) & 0xFFFFFFFFull) << 32);
                        pw[(1 * BM + ra + hh) * 2 + 1] = ((bz[2] >> sh) & 0xFFFFFFFFull) | (((bz[3] >> sh) & 0xFFFFFFFFull) << 32);
                        pw[(2 * BM + ra + hh) * 2 + 0] = ((bq[0] >> sh) & 0xFFFFFFFFull) | (((bq[1] >> sh) & 0xFFFFFFFFull) << 32);
                        pw[(2 * BM + ra + hh) * 2 + 1] = ((bq[2] >> sh) & 0xFFFFFFFFull) | (((bq[3] >> sh) & 0xFFFFFFFFull) << 32);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            const int e = tid + 1024 * u;
            *reinterpret_cast<uint4*>(&Bt[(e >> 3) * LDA + 16 * (e & 7)]) = bstage[u];
        }
        __syncthreads();
        // the next chunk's A values: requested now, consumed after the MFMAs below (unconditional, clamped: a branch around the requests
        // would make the waitcnt pass drain them early; the last chunk re-reads its own columns)
        SVNET_BL_LOAD_A(min(k0 + KC, (nchunks - 1) * KC));
        __builtin_amdgcn_sched_barrier(0);
        // ---- row-sliced planes of this chunk: 3 planes x 2 row blocks (64 rows) x 4 column groups of 32 = 24 transposes of
        // (2 x) 32 x 32 bits; both halves of a wave transpose at once (lanes 0-31: rows 0-31 of the block, lanes 32-63: rows 32-63)
        if (save) {
            for (int item = wave; item < 24; item += 16) {
                const int plane = item >> 3, rb = (item >> 2) & 1, cg = item & 3;
                const uint64_t wrd = pw[(plane * BM + 64 * rb + lane) * 2 + (cg >> 1)];
                const uint32_t mine = (cg & 1) ? (uint32_t)(wrd >> 32) : (uint32_t)wrd;
                const uint32_t colword = svnet_bit_transpose32(mine, lane);      // lane b of a half: column 32 cg + b over the half's 32 rows
                const int k = k0 + 32 * cg + (lane & 31);
                const int64_t blk = (m0 >> 6) + rb;                               // 64-row block of the matrix
                if (k < K && blk * 64 < M) {
                    uint32_t* dst = plane == 0 ? x_sign32 : (plane == 1 ? x_nz32 : x_ste32);
                    dst[(blk * K + k) * 2 + (lane >> 5)] = colword;
                }
            }
        }
        // ---- MFMA: 4 k-steps of 32 positions
#pragma unroll
        for (int s = 0; s < KC / 32; ++s) {
            const i32x4 af = *reinterpret_cast<const i32x4*>(&At[(32 * wr + r) * LDA + 32 * s + 16 * h]);
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const i32x4 bf = *reinterpret_cast<const i32x4*>(&Bt[(32 * (NTW * wc + t) + r) * LDA + 32 * s + 16 * h]);
                acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();      // At / Bt / pw are rewritten by the next chunk
    }
#undef SVNET_BL_LOAD_A
    // ---- (optional) the columns of the layer's input that are constant over a cloud's rows (a broadcast half of a concatenation,
    // sv_dgcnn_partseg.py:115-118 `repeat` + `cat`, sv_pointnet_cls.py:43-52 `expand_as` + `svcat`) are not multiplied row by row: their
    // integer count n_cloud[b, o] - the same XNOR-popcount sum over THEIR columns, computed once per cloud - is added to the per-point
    // count here.  The sum of the two counts IS the count over the full row: outputs and batch statistics identical to the full product.
    if (cloud_n) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int o = min(o0 + 32 * (NTW * wc + t) + r, O - 1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t m = min(m0 + 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h, M - 1);
                acc[t][i] += (int)cloud_n[(m / rows_per_cloud) * O + o];
            }
        }
    }
    // ---- epilogue: y = count * scale + bias  (the expression of binlinear_fwd_kernel: identical outputs)
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int o = o0 + 32 * (NTW * wc + t) + r;
        if (o < O) {
            const float sc = scale[o], bs = bias ? bias[o] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t m = m0 + 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (m < M) y[m * O + o] = (float)acc[t][i] * sc + bs;
            }
        }
    }
    // ---- (optional) column sums of y for the BatchNorm that follows (sv_layers.py:189: bn(linear(x))): the counts are integers, so the
    // workgroup's sums of n and n^2 are exact; sum y = scale*S1 + R*bias, sum y^2 = scale^2*S2 + 2 scale*bias*S1 + R*bias^2 over its R rows,
    // in double, one atomic pair per column and workgroup - the layer's output is not read again for its statistics
    if (col_sums) {
        long long* red = reinterpret_cast<long long*>(bl_lds);      // [2][OT] <= 8 KiB over the A tile (consumed: the loop ended on a barrier)
        for (int i = tid; i < 2 * OT; i += 1024) red[i] = 0;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            int s1 = 0;
            long long s2 = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t m = m0 + 32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h;
                const int c = m < M ? acc[t][i] : 0;
                s1 += c;
                s2 += (long long)(c * c);                            // (|c| <= K <= 46340, checked on the host: c * c < 2^31)
            }
            atomicAdd(reinterpret_cast<unsigned long long*>(&red[32 * (NTW * wc + t) + r]), (unsigned long long)(long long)s1);
            atomicAdd(reinterpret_cast<unsigned long long*>(&red[OT + 32 * (NTW * wc + t) + r]), (unsigned long long)s2);
        }
        __syncthreads();
        const int o = o0 + tid;
        if (tid < OT && o < O) {
            const double sc = (double)scale[o], bs = bias ? (double)bias[o] : 0.0;
            const double S1 = (double)red[tid], S2 = (double)red[OT + tid];
            const double R = (double)min((int64_t)BM, M - m0);
            double* sl = svnet_slice_ptr(col_sums, 2 * O);
            svnet_slice_add(&sl[o], sc * S1 + R * bs);
            svnet_slice_add(&sl[O + o], sc * sc * S2 + 2.0 * sc * bs * S1 + R * bs * bs);
            // (the slices are added up by the BatchNorm kernel that consumes them: svnet_bn_finalize_f32)
        }
    }
}

}  // namespace

extern "C" size_t svnet_binweight_i8_bytes(int64_t O, int64_t K) { return (O > 0 && K > 0) ? (size_t)O * (size_t)((K + KC - 1) / KC * KC) : 0; }

extern "C" int svnet_binweight_pack_i8(const float* W, int64_t O, int64_t K, int8_t* w_i8, void* stream) {
    SVNET_REQUIRE(W && w_i8 && O > 0 && K > 0 && O * ((K + KC - 1) / KC * KC) < ((int64_t)1 << 31), SVNET_E_ARG, "svnet_binweight_pack_i8: bad arguments");
    const int Kp = (int)((K + KC - 1) / KC * KC);
    hipLaunchKernelGGL(binweight_pack_i8_kernel, dim3(svnet_grid(O * Kp, 256)), dim3(256), 0, (hipStream_t)stream, W, (int)O, (int)K, Kp, w_i8);
    SVNET_CHECK_LAUNCH("binweight_pack_i8_kernel");
    return SVNET_OK;
}

static int binlinear_i8_launch(const float* x, int64_t ldx, const float* beta, const int8_t* w_i8, const float* scale,
                               const float* bias, int64_t M, int64_t K, int64_t O, float* y, uint64_t* x_sign, uint64_t* x_nz,
                               uint64_t* x_ste, double* col_sums, const float* cloud_n, int64_t rows_per_cloud, void* stream) {
    SVNET_REQUIRE(x && beta && w_i8 && scale && y, SVNET_E_ARG, "svnet_binlinear_i8_fwd_f32: null pointer");
    SVNET_REQUIRE(!cloud_n || (rows_per_cloud > 0 && M % rows_per_cloud == 0), SVNET_E_ARG,
                  "svnet_binlinear_i8_cloud_fwd_f32: rows_per_cloud must divide M");
    SVNET_REQUIRE(M >= 0 && K > 0 && O > 0 && ldx >= K, SVNET_E_ARG, "svnet_binlinear_i8_fwd_f32: bad sizes");
    const bool any = x_sign || x_nz || x_ste, all = x_sign && x_nz && x_ste;
    SVNET_REQUIRE(!any || all, SVNET_E_ARG, "svnet_binlinear_i8_fwd_f32: pass all three saved planes or none");
    SVNET_REQUIRE(K < (1 << 24) && O < (1 << 24), SVNET_E_UNSUPPORTED, "svnet_binlinear_i8_fwd_f32: K=%lld, O=%lld too large", (long long)K, (long long)O);
    SVNET_REQUIRE(!col_sums || K <= 46340, SVNET_E_UNSUPPORTED, "svnet_binlinear_i8_fwd_f32: column sums need K <= 46340 (got %lld)", (long long)K);
    if (M == 0) return SVNET_OK;
    const int Kp = (int)((K + KC - 1) / KC * KC);
    // 128, 256 or 512 output channels per workgroup (16 waves: one, two or four 32-column tiles per wave)
    const int ntw = O <= 128 ? 1 : (O <= 256 ? 2 : 4);
    const dim3 grid((unsigned)svnet_cdiv(M, BM), (unsigned)svnet_cdiv(O, 128 * ntw));
    const size_t lds = (size_t)BM * LDA + (size_t)128 * ntw * LDA + 3 * BM * 2 * sizeof(uint64_t);
    bool ok = true;                                      // more than 64 KiB of dynamic LDS needs an explicit opt-in (NTW = 4: 96 KiB), per device
    if (ntw == 1) SVNET_LDS_OPTIN(ok, BM * LDA + 128 * LDA + 6144, "binlinear_i8_fwd_kernel<1>", reinterpret_cast<const void*>(&binlinear_i8_fwd_kernel<1>));
    else if (ntw == 2) SVNET_LDS_OPTIN(ok, BM * LDA + 256 * LDA + 6144, "binlinear_i8_fwd_kernel<2>", reinterpret_cast<const void*>(&binlinear_i8_fwd_kernel<2>));
    else SVNET_LDS_OPTIN(ok, BM * LDA + 512 * LDA + 6144, "binlinear_i8_fwd_kernel<4>", reinterpret_cast<const void*>(&binlinear_i8_fwd_kernel<4>));
    if (!ok) return SVNET_E_LAUNCH;
#define SVNET_BL_LAUNCH(NTW_)                                                                                                          \
    hipLaunchKernelGGL((binlinear_i8_fwd_kernel<NTW_>), grid, dim3(1024), lds, (hipStream_t)stream, x, ldx, beta, w_i8, scale, bias, M, (int)K, \
                       (int)O, Kp, y, reinterpret_cast<uint32_t*>(x_sign), reinterpret_cast<uint32_t*>(x_nz), reinterpret_cast<uint32_t*>(x_ste), col_sums, \
                       cloud_n, rows_per_cloud > 0 ? rows_per_cloud : 1)
    if (ntw == 1) SVNET_BL_LAUNCH(1);
    else if (ntw == 2) SVNET_BL_LAUNCH(2);
    else SVNET_BL_LAUNCH(4);
#undef SVNET_BL_LAUNCH
    SVNET_CHECK_LAUNCH("binlinear_i8_fwd_kernel");
    return SVNET_OK;
}

extern "C" int svnet_binlinear_i8_fwd_f32(const float* x, int64_t ldx, const float* beta, const int8_t* w_i8, const float* scale,
                                          const float* bias, int64_t M, int64_t K, int64_t O, float* y, uint64_t* x_sign, uint64_t* x_nz,
                                          uint64_t* x_ste, double* col_sums, void* stream) {
    return binlinear_i8_launch(x, ldx, beta, w_i8, scale, bias, M, K, O, y, x_sign, x_nz, x_ste, col_sums, nullptr, 0, stream);
}

extern "C" int svnet_binlinear_i8_cloud_fwd_f32(const float* x, int64_t ldx, const float* beta, const int8_t* w_i8, const float* scale,
                                                const float* bias, int64_t M, int64_t K, int64_t O, float* y, uint64_t* x_sign,
                                                uint64_t* x_nz, uint64_t* x_ste, double* col_sums, const float* cloud_n,
                                                int64_t rows_per_cloud, void* stream) {
    SVNET_REQUIRE(cloud_n, SVNET_E_ARG, "svnet_binlinear_i8_cloud_fwd_f32: null cloud counts");
    return binlinear_i8_launch(x, ldx, beta, w_i8, scale, bias, M, K, O, y, x_sign, x_nz, x_ste, col_sums, cloud_n, rows_per_cloud, stream);
}
