// Binarized dense layers: ternary bit-planes + XNOR/popcount.
//
// Replaces  models/sv_layers.py:29-53 (Linear with bw/ba) and :55-78 (Conv1d binary):
//   x_b = sign(x + beta)   (sign(0) = 0  -> TERNARY)
//   w_b = sign(W)
//   y   = (x_b . w_b^T) * scale (+ bias)
// With a sign plane s and a non-zero plane z per operand,
//   dot = popc(zx & zw) - 2 * popc(zx & zw & (sx ^ sw)).
// This is a bitwise path (no MFMA): per 64-bit word 10 VALU ops, the row bits are produced by
// wave-wide ballots straight from the coalesced fp32 row loads, the weight words of "my" output
// channel live in registers for the whole kernel, and row bit-planes are staged in LDS so that all
// four waves of a workgroup share one packing pass.  HBM traffic = read x once + write y once
// (+ 3 bits per input element of saved planes in training, instead of keeping the fp32 input).
#include "common.h"

namespace {

constexpr int ROWS = 64;  // rows per workgroup tile

__global__ __launch_bounds__(256) void binweight_values_kernel(const float* __restrict__ W, const float* __restrict__ scale,
                                                               int64_t O, int64_t K, float* __restrict__ w_b,
                                                               float* __restrict__ w_eff) {
    const int64_t total = O * K;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const float w = W[e];
        const float s = (w > 0.f) ? 1.f : ((w < 0.f) ? -1.f : 0.f);
        if (w_b) w_b[e] = s;
        if (w_eff) w_eff[e] = s * scale[e / K];
    }
}

// one wave per (row, word): lane b loads the weight of bit b, two ballots make the plane words
__global__ __launch_bounds__(256) void binweight_pack_kernel(const float* __restrict__ W, int64_t O, int64_t K, int64_t KW,
                                                             uint64_t* __restrict__ w_sign, uint64_t* __restrict__ w_nz) {
    const int lane = threadIdx.x & 63;
    const int64_t total = O * KW;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t e = wave0; e < total; e += nwaves) {
        const int64_t o = e / KW, w = e - o * KW;
        const int64_t k = w * 64 + lane;
        const float v = k < K ? W[o * K + k] : 0.f;
        const uint64_t sg = __ballot(v > 0.f), nz = __ballot(v != 0.f);
        if (lane == 0) { w_sign[e] = sg; w_nz[e] = nz; }
    }
}

// KWM: compile-time bound on the number of 64-bit words per row (KW <= KWM).
template <int KWM, int PPM>
__global__ __launch_bounds__(256) void binlinear_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ beta,
                                                            const uint64_t* __restrict__ w_sign,
                                                            const uint64_t* __restrict__ w_nz,
                                                            const float* __restrict__ scale, const float* __restrict__ bias,
                                                            int64_t M, int K, int O, int o_base, int o_blocks /*workgroups per row tile (small M)*/,
                                                            int KW, int og_shift /*log2 threads per row group*/,
                                                            float* __restrict__ y, uint64_t* __restrict__ x_sign,
                                                            uint64_t* __restrict__ x_nz, uint64_t* __restrict__ x_ste,
                                                            int wld /*words per weight row*/, int64_t pld /*saved-plane row stride*/,
                                                            int ymode /*K chunks: bit 0 = add the int32 count kept in y, bit 1 = keep the count*/) {
    extern __shared__ uint64_t lds[];  // [3][ROWS][KW]
    uint64_t* ls = lds;
    uint64_t* lz = lds + (size_t)ROWS * KW;
    uint64_t* lt = lds + (size_t)2 * ROWS * KW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // y = count * scale + bias; with K split into chunks (K > 2176) the exact integer count of the earlier chunks travels in y
    // itself (as int32): the popcount sum stays an exact integer whatever the split, so equal sums stay exactly equal
    auto emit = [&](float* dst, int cnt, float scv, float bsv) {
        if (ymode & 1) cnt += *reinterpret_cast<const int*>(dst);
        if (ymode & 2) *reinterpret_cast<int*>(dst) = cnt;
        else *dst = (float)cnt * scv + bsv;
    };
    const int OG = 1 << og_shift;           // threads that share one row in phase 2 (32..256)
    const int nsub = 256 >> og_shift;        // row sub-groups in phase 2
    const int o_in = tid & (OG - 1);
    const int sub = tid >> og_shift;
    // few row tiles (the classifier head: M = batch): o_blocks workgroups share a tile, each owning OG output channels
    const int ob = (int)(blockIdx.x % o_blocks);
    o_base += ob * OG;

    // my output channels' weight words (registers for the whole kernel).  WIDE (KWM > 16, i.e. K > 1024): 2 x 34 words per
    // thread were 270 VGPRs = one wave per SIMD; there the loops are turned inside out instead (word outermost, one weight word
    // pair from L2 per step, 64 row counters in registers: ~110 VGPRs)
    constexpr bool WIDE = KWM > 16;
    constexpr int KWR = WIDE ? 1 : KWM;
    uint64_t wsg[PPM][KWR], wnz[PPM][KWR];
    float sc[PPM], bs[PPM];
#pragma unroll
    for (int p = 0; p < PPM; ++p) {
        const int o = o_base + o_in + 256 * p;
        const bool ok = (o < O);
        sc[p] = ok ? scale[o] : 0.f;
        bs[p] = (ok && bias) ? bias[o] : 0.f;
        if (!WIDE) {
#pragma unroll
            for (int w = 0; w < KWR; ++w) {
                const bool okw = ok && (w < KW);
                wsg[p][w] = okw ? w_sign[(size_t)o * wld + w] : 0ull;
                wnz[p][w] = okw ? w_nz[(size_t)o * wld + w] : 0ull;
            }
        }
    }

    const int64_t tiles = (M + ROWS - 1) / ROWS;
    for (int64_t tile = blockIdx.x / o_blocks; tile < tiles; tile += gridDim.x / o_blocks) {
        const int64_t row0 = tile * ROWS;
        const int rows = (int)min((int64_t)ROWS, M - row0);
        // ---- phase 1: binarize + pack. wave w handles rows w, w+4, ... in batches of PB words; the next batch's loads (possibly
        // of the next row) are issued before the current batch is balloted, so 2*PB loads per lane are always in flight
        {
            int lr = wave, lw = 0;                                   // load cursor (row, first word of the batch)
            constexpr int PB = KWM >= 16 ? 16 : (KWM >= 8 ? 8 : 4);  // wide rows: whole-row batches (these passes are pure latency)
            // RAW values are kept (x and beta apart) and the request is unconditional (past the tile's last row it repeats that
            // row): an add at request time waits for both loads right there, and a branch around the request makes the waitcnt
            // pass drain it - either way every batch paid a full memory latency
            float tn[PB], bn[PB];
#define SVNET_BL_LOAD()                                                                                   \
    do {                                                                                                  \
        const float* xr_ = x + (row0 + min(lr, rows - 1)) * ldx;                                          \
        _Pragma("unroll") for (int u = 0; u < PB; ++u) {                                                  \
            const int k_ = (lw + u) * 64 + lane;                                                          \
            const int kc_ = k_ < K ? k_ : K - 1; /* clamped: unconditional loads */                       \
            tn[u] = xr_[kc_];                                                                             \
            bn[u] = beta[kc_];                                                                            \
        }                                                                                                 \
    } while (0)
            SVNET_BL_LOAD();
            int r = wave, w0 = 0;                                    // use cursor
            while (r < rows) {
                float t[PB];
#pragma unroll
                for (int u = 0; u < PB; ++u) t[u] = tn[u] + bn[u];
                lw += PB;
                if (lw >= KW) { lw = 0; lr += 4; }
                SVNET_BL_LOAD();
#pragma unroll
                for (int u = 0; u < PB; ++u) {
                    const int w = w0 + u;
                    const bool in = (w * 64 + lane) < K;
                    const float tv = in ? t[u] : 0.f;
                    const uint64_t sg = __ballot(tv > 0.f);
                    const uint64_t nz = __ballot(tv != 0.f);
                    const uint64_t st = __ballot(in && (fabsf(tv) <= 1.2f));
                    if (lane == 0 && w < KW) {
                        ls[r * KW + w] = sg;
                        lz[r * KW + w] = nz;
                        lt[r * KW + w] = st;
                    }
                }
                w0 += PB;
                if (w0 >= KW) { w0 = 0; r += 4; }
            }
#undef SVNET_BL_LOAD
        }
        __syncthreads();
        // ---- saved planes (training), written ROW-SLICED: word [tile*K + k] = bit r of column k for the tile's
        // 64 rows (a 64x64 bit transpose per word by 64 ballots; rows beyond M contribute 0).  This is the layout
        // the backward MFMA kernels consume with one coalesced u64 per lane (gemm_mfma.hip).
        if (x_sign) {   // (the o_blocks workgroups of a small-M tile share the 3*KW transpositions)
            for (int item = ob * 4 + wave; item < 3 * KW; item += 4 * o_blocks) {
                const int pl = item / KW, w = item - pl * KW;
                const uint64_t* src = (pl == 0) ? ls : ((pl == 1) ? lz : lt);
                uint64_t* dst = (pl == 0) ? x_sign : ((pl == 1) ? x_nz : x_ste);
                const uint64_t mine_row = (lane < rows) ? src[lane * KW + w] : 0ull;
                uint64_t col_word = 0ull;
#pragma unroll 8
                for (int b = 0; b < 64; ++b) {
                    const uint64_t t = __ballot((mine_row >> b) & 1ull);
                    if (lane == b) col_word = t;
                }
                const int k = w * 64 + lane;
                if (k < K) dst[tile * pld + k] = col_word;
            }
        }
        // ---- phase 2: popcount dot products
        if (WIDE && og_shift < 8) {
            // small M (the classifier head: a few row tiles, 32 channels per workgroup, 8 threads per channel take different rows):
            // weight words straight from L2, one row at a time
            const int o = min(o_base + o_in, O - 1);
            for (int r = sub; r < rows; r += nsub) {
                int cnt = 0;
                for (int w = 0; w < KW; ++w) {
                    const uint64_t m = lz[r * KW + w] & w_nz[(size_t)o * wld + w];
                    cnt += __popcll(m) - 2 * __popcll(m & (ls[r * KW + w] ^ w_sign[(size_t)o * wld + w]));
                }
                if (o_base + o_in < O) emit(&y[(row0 + r) * O + o], cnt, sc[0], bs[0]);
            }
        } else if (WIDE) {
            // (PPM = 1, og_shift = 8: thread = output channel, every thread walks all ROWS rows)
            const int o = min(o_base + o_in, O - 1);
            int cnt[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) cnt[r] = 0;
            uint64_t ws_n = w_sign[(size_t)o * wld], wz_n = w_nz[(size_t)o * wld];
            for (int w = 0; w < KW; ++w) {
                const uint64_t ws = ws_n, wz = wz_n;
                const int wn = min(w + 1, KW - 1);                     // next word pair (unconditional, clamped)
                ws_n = w_sign[(size_t)o * wld + wn]; wz_n = w_nz[(size_t)o * wld + wn];
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const uint64_t xs = ls[r * KW + w], xz = lz[r * KW + w];   // wave-uniform addresses: broadcast reads
                    const uint64_t m = xz & wz;
                    cnt[r] += __popcll(m) - 2 * __popcll(m & (xs ^ ws));
                }
            }
            if (o_base + o_in < O) {
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
                    if (r < rows) emit(&y[(row0 + r) * O + o], cnt[r], sc[0], bs[0]);
            }
        } else
        if (o_base + o_in < O) {
            for (int r = sub; r < rows; r += nsub) {
                // two running popcounts (v_bcnt_u32_b32 adds into its third operand for free) combined once per row, instead of
                // popc(m) - 2*popc(dneg) added up word by word
                int pm[PPM], pd[PPM];
#pragma unroll
                for (int p = 0; p < PPM; ++p) pm[p] = pd[p] = 0;
#pragma unroll
                for (int w = 0; w < KWM; ++w) {
                    if (w < KW) {
                        const uint64_t xs = ls[r * KW + w], xz = lz[r * KW + w];
#pragma unroll
                        for (int p = 0; p < PPM; ++p) {
                            const uint64_t m = xz & wnz[p][w];
                            const uint64_t dneg = m & (xs ^ wsg[p][w]);
                            pm[p] += __popcll(m);
                            pd[p] += __popcll(dneg);
                        }
                    }
                }
#pragma unroll
                for (int p = 0; p < PPM; ++p) {
                    const int o = o_base + o_in + 256 * p;
                    if (o < O) emit(&y[(row0 + r) * O + o], pm[p] - 2 * pd[p], sc[p], bs[p]);
                }
            }
        }
        __syncthreads();
    }
}

// dW[o,k] (+)= scale[o]*GX[o,k]*[|W|<=1.2] ; dscale[o] (+)= sum_k sign(W[o,k])*GX[o,k].  One wave per output row.
// gx_sliced: GX is a sliced accumulator of O*K floats (SVNET_SLICED_LEN: Vector2Scalar's weight-gradient sums) whose slices are added up
// on the way in.  sum_buf / sum_len (optional): another sliced accumulator of this layer's backward (dL/dbeta: the column sums of the
// input-gradient product) whose totals workgroup 0 leaves in its first sum_len elements - the launch svnet_slices_sum_f32 would have been.
__global__ __launch_bounds__(256) void binweight_grad_kernel(const float* __restrict__ GX, const float* __restrict__ W,
                                                             const float* __restrict__ scale, int64_t O, int64_t K,
                                                             float* __restrict__ dW, float* __restrict__ dscale, int accumulate,
                                                             int gx_sliced, float* __restrict__ sum_buf, int sum_len) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    if (sum_buf && blockIdx.x == 0)
        for (int i = threadIdx.x; i < sum_len; i += blockDim.x) sum_buf[i] = svnet_slices_total(sum_buf, sum_len, i);
    const int L = (int)(O * K);
    for (int64_t o = wave; o < O; o += nwaves) {
        const float sc = scale[o];
        float part = 0.f;
        for (int64_t k = lane; k < K; k += 64) {
            const float w = W[o * K + k], g = gx_sliced ? svnet_slices_total(GX, L, (int)(o * K + k)) : GX[o * K + k];
            const float s = (w > 0.f) ? 1.f : ((w < 0.f) ? -1.f : 0.f);
            part += s * g;
            if (dW) {
                const float t = (fabsf(w) <= 1.2f) ? sc * g : 0.f;
                dW[o * K + k] = accumulate ? dW[o * K + k] + t : t;
            }
        }
        part = wave_sum(part);
        if (lane == 0 && dscale) dscale[o] = accumulate ? dscale[o] + part : part;
    }
}

template <int KWM, int PPM>
void launch_fwd(const float* x, int64_t ldx, const float* beta, const uint64_t* w_sign, const uint64_t* w_nz, const float* scale,
                const float* bias, int64_t M, int K, int O, int KW, float* y, uint64_t* xs, uint64_t* xz, uint64_t* xt,
                int wld, int64_t pld, int ymode, hipStream_t st) {
    int og_shift = 5;
    while ((1 << og_shift) < O && og_shift < 8) ++og_shift;
    const int64_t tiles = svnet_cdiv(M, ROWS);
    const size_t lds_bytes = (size_t)3 * ROWS * KW * sizeof(uint64_t);
    if (PPM == 1 && tiles <= 8 && O > 32) {
        // small M: one launch, ceil(O/32) workgroups per row tile (each repeats the cheap packing pass of its tile)
        const int o_blocks = (O + 31) / 32;
        hipLaunchKernelGGL((binlinear_fwd_kernel<KWM, PPM>), dim3((unsigned)(tiles * o_blocks)), dim3(256), lds_bytes, st, x, ldx, beta,
                           w_sign, w_nz, scale, bias, M, K, O, 0, o_blocks, KW, 5, y, xs, xz, xt, wld, pld, ymode);
        return;
    }
    if (PPM == 1 && O > 256 && tiles * ((O + 255) / 256) <= 256 * 16) {
        // a few hundred row tiles (conv5: 512 tiles x 512 channels): one workgroup per (tile, 256 channels) in ONE launch, so that the
        // grid has >= 4 workgroups per CU (2 per CU - 2 waves per SIMD - left every load latency of the packing pass exposed)
        const int o_blocks = (O + 255) / 256;
        hipLaunchKernelGGL((binlinear_fwd_kernel<KWM, PPM>), dim3((unsigned)(tiles * o_blocks)), dim3(256), lds_bytes, st, x, ldx, beta,
                           w_sign, w_nz, scale, bias, M, K, O, 0, o_blocks, KW, 8, y, xs, xz, xt, wld, pld, ymode);
        return;
    }
    const unsigned grid = (unsigned)(tiles < 256 * 8 ? tiles : 256 * 8);
    for (int o_base = 0; o_base < O; o_base += 256 * PPM)
        hipLaunchKernelGGL((binlinear_fwd_kernel<KWM, PPM>), dim3(grid), dim3(256), lds_bytes, st, x, ldx, beta, w_sign, w_nz, scale,
                           bias, M, K, O, o_base, 1, KW, og_shift, y, xs, xz, xt, wld, pld, ymode);
}

}  // namespace

extern "C" int svnet_binweight_prepare_f32(const float* W, const float* scale, int64_t O, int64_t K, uint64_t* w_sign,
                                           uint64_t* w_nz, float* w_b, float* w_eff, void* stream) {
    SVNET_REQUIRE(W && O > 0 && K > 0, SVNET_E_ARG, "svnet_binweight_prepare_f32: bad arguments");
    SVNET_REQUIRE(!w_eff || scale, SVNET_E_ARG, "svnet_binweight_prepare_f32: w_eff needs scale");
    SVNET_REQUIRE((w_sign == nullptr) == (w_nz == nullptr), SVNET_E_ARG, "svnet_binweight_prepare_f32: need both planes or none");
    hipStream_t st = (hipStream_t)stream;
    if (w_b || w_eff) {
        hipLaunchKernelGGL(binweight_values_kernel, dim3(svnet_grid(O * K, 256)), dim3(256), 0, st, W, scale, O, K, w_b, w_eff);
        SVNET_CHECK_LAUNCH("binweight_values_kernel");
    }
    if (w_sign) {
        const int64_t KW = svnet_cdiv(K, 64);
        hipLaunchKernelGGL(binweight_pack_kernel, dim3(svnet_grid(O * KW * 64, 256)), dim3(256), 0, st, W, O, K, KW, w_sign, w_nz);
        SVNET_CHECK_LAUNCH("binweight_pack_kernel");
    }
    return SVNET_OK;
}

extern "C" int svnet_binlinear_fwd_f32(const float* x, int64_t ldx, const float* beta, const uint64_t* w_sign,
                                       const uint64_t* w_nz, const float* scale, const float* bias, int64_t M, int64_t K,
                                       int64_t O, float* y, uint64_t* x_sign, uint64_t* x_nz, uint64_t* x_ste, void* stream) {
    SVNET_REQUIRE(x && beta && w_sign && w_nz && scale && y, SVNET_E_ARG, "svnet_binlinear_fwd_f32: null pointer");
    SVNET_REQUIRE(M >= 0 && K > 0 && O > 0 && ldx >= K, SVNET_E_ARG, "svnet_binlinear_fwd_f32: bad sizes");
    const bool any = x_sign || x_nz || x_ste, all = x_sign && x_nz && x_ste;
    SVNET_REQUIRE(!any || all, SVNET_E_ARG, "svnet_binlinear_fwd_f32: pass all three saved planes or none");
    SVNET_REQUIRE(K <= (1 << 20) && O <= (1 << 20), SVNET_E_UNSUPPORTED, "svnet_binlinear_fwd_f32: K=%lld, O=%lld too large", (long long)K,
                  (long long)O);
    const int64_t KWF = svnet_cdiv(K, 64);                    // words per full row (= weight plane row stride)
    if (M == 0) return SVNET_OK;
    hipStream_t st = (hipStream_t)stream;
    // K > 2176 (34 words: what the row tile's planes may take in LDS) is processed in chunks of whole words; the integer count of
    // the earlier chunks travels in y (int32), the last chunk applies scale and bias.  O > 256 (512 with two channels per thread)
    // takes several launches over the output channels, each of which repeats the (cheap) packing pass.
    const int64_t nchunk = svnet_cdiv(KWF, 34);
    const int64_t wpc = svnet_cdiv(KWF, nchunk);              // words per chunk (<= 34)
    for (int64_t c = 0; c < nchunk; ++c) {
        const int64_t w0 = c * wpc, col0 = w0 * 64;
        const int64_t Kc = (c + 1 == nchunk) ? K - col0 : wpc * 64;
        const int64_t KW = svnet_cdiv(Kc, 64);
        const int ymode = (c > 0 ? 1 : 0) | (c + 1 < nchunk ? 2 : 0);
#define SVNET_BL(KWM, PPM)                                                                                                          \
    launch_fwd<KWM, PPM>(x + col0, ldx, beta + col0, w_sign + w0, w_nz + w0, scale, bias, M, (int)Kc, (int)O, (int)KW, y,          \
                         x_sign ? x_sign + col0 : nullptr, x_nz ? x_nz + col0 : nullptr, x_ste ? x_ste + col0 : nullptr, (int)KWF, K, \
                         ymode, st)
        // two output channels per thread only when there are row tiles in plenty; fewer tiles take a one-launch split over the
        // output channels (PPM = 1) so that the grid still fills the chip
        const bool two = O > 256 && svnet_cdiv(M, ROWS) * ((O + 255) / 256) > 256 * 16;
        if (KW <= 2) { if (two) SVNET_BL(2, 2); else SVNET_BL(2, 1); }
        else if (KW <= 4) { if (two) SVNET_BL(4, 2); else SVNET_BL(4, 1); }
        else if (KW <= 8) { if (two) SVNET_BL(8, 2); else SVNET_BL(8, 1); }
        else if (KW <= 16) SVNET_BL(16, 1);
        else SVNET_BL(34, 1);
#undef SVNET_BL
        SVNET_CHECK_LAUNCH("binlinear_fwd_kernel");
    }
    return SVNET_OK;
}

extern "C" int svnet_binweight_grad_f32(const float* GX, const float* W, const float* scale, int64_t O, int64_t K,
                                        float* dW, float* dscale, int accumulate, int gx_sliced, float* sum_buf, int64_t sum_len, void* stream) {
    SVNET_REQUIRE(GX && W && scale && O > 0 && K > 0 && O * K < ((int64_t)1 << 30) && (!sum_buf || (sum_len > 0 && sum_len < ((int64_t)1 << 30))),
                  SVNET_E_ARG, "svnet_binweight_grad_f32: bad arguments");
    hipLaunchKernelGGL(binweight_grad_kernel, dim3(svnet_grid(O * 64, 256)), dim3(256), 0, (hipStream_t)stream, GX, W, scale, O, K,
                       dW, dscale, accumulate, gx_sliced, sum_buf, (int)sum_len);
    SVNET_CHECK_LAUNCH("binweight_grad_kernel");
    return SVNET_OK;
}
