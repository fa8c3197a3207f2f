// MFMA kernels for the two GEMM families that dominate the backward (and the sign-weight forward):
//
//   rows :  C[M,N] = A[M,K] . B[K,N]        M = edge/point rows (up to 2.6 M), K,N <= 512, B EXACT in bf16 (+-1/0)
//   tn   :  C[p,q] = sum_m A[m,p] . B[m,q]   reduction over the rows (weight gradients), B fp32 or ternary bit-planes
//
// Both run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation and an EXACT operand decomposition: an fp32 value is
// split into three bf16 pieces x = h + m + l (8+8+8 mantissa bits, by truncation, each residual exact), and the
// other operand is exactly representable in bf16 (sign weights, ternary activations), so every product is exact
// and the result differs from an fp32 GEMM only by summation order.  Where both operands are general fp32 (tn with
// fp32 B) the six leading cross terms are used (hh, hm, mh, hl, lh, mm: relative error < 2^-23).
// These shapes are HBM-bound (tall and skinny); three MFMAs per fragment at the bf16 rate cost 3/16 of an f32 MFMA.
//
// Fragment maps (cdna_hip_programming.md §3): lane l, r = l & 31, h = l >> 5
//   A: row r, k = 8h + j (j = 0..7)      B: k = 8h + j, col r      C/D reg i: row (i&3) + 8(i>>2) + 4h, col r.
//
// Bit-plane layout ("row-sliced", written by binlinear_fwd): word [(m >> 6) * K + k] holds bit (m & 63) of rows
// 64*(m>>6) .. +63 for column k — one coalesced u64 per lane gives a column's bits for 64 consecutive rows, which
// is exactly what the k(=m)-strided MFMA operand and the per-column epilogue mask need.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ __bf16 bf16_from_bits(uint32_t b) { return __builtin_bit_cast(__bf16, (unsigned short)b); }

struct Split3 {
    bf16x8 h, m, l;
};

__device__ __forceinline__ void split3(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
    const uint32_t hu = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(hu);  // exact
    const uint32_t mu = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(mu);  // exact, <= 8 significant bits
    h = hu >> 16;
    m = mu >> 16;
    l = __float_as_uint(r2) >> 16;
}

__device__ __forceinline__ Split3 split_frag(const float (&x)[8]) {
    Split3 s;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t h, m, l;
        split3(x[j], h, m, l);
        s.h[j] = bf16_from_bits(h);
        s.m[j] = bf16_from_bits(m);
        s.l[j] = bf16_from_bits(l);
    }
    return s;
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
// Diagnostic builds only (-DSVNET_ROWS_ABLATE=n, results WRONG): 1 no C stores, 2 no A loads, 3 no B staging loads, 4 no MFMAs
#ifndef SVNET_ROWS_ABLATE
#define SVNET_ROWS_ABLATE 0
#endif

// ------------------------------------------------------------------------------------------------ rows kernel
struct RowsArgs {
    const float* A; int64_t lda; const float* a_scale;
    const float* B; int64_t b_rs, b_cs;
    float* C; int64_t ldc;
    int64_t M; int N, K;
    float alpha; const float* col_scale; const float* bias;
    const uint64_t* mask;   // row-sliced [ceil(M/64)][N]
    float* col_sum;
    int accumulate;
    int a_vec;              // A rows are 16-byte aligned and lda % 4 == 0: fragment loads as 2 x dwordx4
    const uint16_t* B16;    // optional: B packed as bf16 [round32(N)][Kp] (k contiguous, zero padded), Kp = round16(K)
    int Kp;
    int64_t b_piece;        // split B (svnet_mfma_rows_split): elements between the three packed pieces in B16; 0 = one piece
};

// B (any strides, values exact in bf16) -> bf16 [Np][Kp], zero padded
// piece: 0 = the value's upper 16 bits (exact for +-1 / 0 / any bf16-representable B), 1 / 2 = the second / third bf16 piece of the
// exact three-way split x = h + m + l (general fp32 B: svnet_mfma_rows_split)
// piece 3 = all three, `piece_stride` elements apart (one launch per product instead of three)
__global__ void pack_b_bf16_kernel(const float* __restrict__ B, int64_t b_rs, int64_t b_cs, int K, int N, int Kp, int Np,
                                   uint16_t* __restrict__ out, int piece = 0, int64_t piece_stride = 0) {
    const int total = Np * Kp;
    // consecutive threads walk n for a fixed k when B is n-contiguous, k otherwise: coalesced reads either way
    const bool n_fast = b_cs == 1;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        int n, k;
        if (n_fast) { n = e % Np; k = e / Np; } else { k = e % Kp; n = e / Kp; }
        float v = 0.f;
        if (n < N && k < K) v = B[(int64_t)k * b_rs + (int64_t)n * b_cs];
        uint32_t h, m, l;
        split3(v, h, m, l);
        if (piece == 3) {
            out[(int64_t)n * Kp + k] = (uint16_t)h;
            out[piece_stride + (int64_t)n * Kp + k] = (uint16_t)m;
            out[2 * piece_stride + (int64_t)n * Kp + k] = (uint16_t)l;
        } else {
            out[(int64_t)n * Kp + k] = (uint16_t)(piece == 0 ? h : (piece == 1 ? m : l));
        }
    }
}

constexpr int KC = 128;         // K chunk staged in LDS
constexpr int LDS_STRIDE = KC + 8;  // bf16 elements per LDS row: (KC/8 + 1) 16-byte slots, odd -> conflict-free b128 reads

// NT = number of 32-column tiles handled by a workgroup (N_tile = 32*NT <= 256); 4 waves x 32 rows per iteration.
// AVEC: A rows are 16-byte aligned, lda % 4 == 0 and K % 8 == 0 -> every fragment is two unconditional dwordx4 loads (a
// k-step past K re-reads the row's last 8 values against zero-padded B); otherwise per-element guarded loads.
// DEEP: K > 64 (more than four k-steps): A fragments are requested several k-steps ahead; short reductions (the K = 10 .. 42 products
// of the edge layers' vector paths) keep one fragment in flight - a deep ring only added clamped, unused requests there.
template <int NT, bool AVEC, bool DEEP>
__global__ __launch_bounds__(256, 2) void mfma_rows_kernel(RowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) __bf16 Bt[];  // [NT*32][LDS_STRIDE]
    // per-k scale of the A operand for the current K chunk (ones without a_scale): read through LDS, i.e. counted by lgkmcnt -
    // a global read per k-step shared vmcnt with the A prefetch and was drained together with it before every MFMA group
    __shared__ __attribute__((aligned(16))) float ascale_l[KC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * (NT * 32);
    const int nchunks = (a.K + KC - 1) / KC;
    const int64_t row_blocks = (a.M + 127) / 128;

    float colpart[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) colpart[t] = 0.f;

    bool b_loaded = false;
    for (int64_t rb = blockIdx.x; rb < row_blocks; rb += gridDim.x) {
        const int64_t m0 = rb * 128 + wave * 32;
        const int64_t arow = m0 + r;
        const bool row_ok = arow < a.M;
        const float* arp = a.A + (row_ok ? arow : 0) * a.lda;     // rows past the end read row 0 and are never stored
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

        // A fragments (8 consecutive k of my row) are requested NPF k-steps ahead of the MFMAs that consume them, in a ring with
        // compile-time slots: A streams from HBM (a new 128-byte line of the row every second k-step), a k-step lasts 0.1 - 0.35 us
        // and with one fragment in flight the kernel sat in s_waitcnt for 60 % of its wave cycles (two waves per SIMD)
        constexpr int NPF = !DEEP ? 1 : (AVEC ? (NT >= 8 ? 4 : 8) : (NT >= 8 ? 2 : 4));   // (register budget) KC / 16 = 8 k-steps per chunk: a multiple of any
        float xn[NPF][8];
#define SVNET_LOAD_A(S, KK)                                                                                   \
    do {                                                                                                      \
        const int kk_ = AVEC ? min((KK), a.K - 8) : (KK);                                                     \
        if (AVEC) {                                                                                           \
            const float4 v0_ = *reinterpret_cast<const float4*>(arp + kk_);                                   \
            const float4 v1_ = *reinterpret_cast<const float4*>(arp + kk_ + 4);                               \
            xn[S][0] = v0_.x; xn[S][1] = v0_.y; xn[S][2] = v0_.z; xn[S][3] = v0_.w;                           \
            xn[S][4] = v1_.x; xn[S][5] = v1_.y; xn[S][6] = v1_.z; xn[S][7] = v1_.w;                           \
        } else {                                                                                              \
            /* clamped, unconditional: columns past K meet zero-padded B rows (and a zero scale).  Wave-uniform tile base in SGPRs + \
               32-bit lane byte offsets: one VGPR and two VALU instructions per element instead of a 64-bit address each */ \
            _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                     \
                xn[S][j] = ld_f32_sbase(abase, row_off + 4u * (uint32_t)min(kk_ + j, a.K - 1));               \
        }                                                                                                     \
    } while (0)
        const int64_t mbase = min(m0, a.M - 1);                                    // (a tile entirely past M reads the last row)
        const float* abase = a.A + mbase * a.lda;
        const uint32_t row_off = (row_ok ? (uint32_t)(arow - mbase) : 0u) * (uint32_t)a.lda * 4u;   // 32 rows * lda * 4 < 2^32: checked on the host
#if SVNET_ROWS_ABLATE == 2
#pragma unroll
        for (int u = 0; u < NPF; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) xn[u][j] = 1.f + (float)lane;
#else
#pragma unroll
        for (int u = 0; u < NPF; ++u) SVNET_LOAD_A(u, 16 * u + 8 * h);
#endif

        for (int ch = 0; ch < nchunks; ++ch) {
            const int k0 = ch * KC;
            const int kc = min(KC, a.K - k0);
            const int kc16 = (kc + 15) & ~15;
            if (nchunks > 1 || !b_loaded) {
                __syncthreads();  // previous readers of Bt are done
                // stage B[k0 : k0+kc, n0 : n0+NT*32] as bf16 [n][k]: one 16-byte LDS store per 8 consecutive k
                const int pieces = NT * 32 * (kc16 >> 3);
                if (tid < KC) ascale_l[tid] = a.a_scale ? ((k0 + tid < a.K) ? a.a_scale[k0 + tid] : 0.f) : 1.f;
                if (SVNET_ROWS_ABLATE == 3) {
                } else if (a.B16) {                  // pre-packed bf16 [n][k]: plain 16-byte copies, eight in flight per thread
                    // (written as load-all-then-store-all batches: the plain loop ran load -> wait -> LDS store sixteen times in a row,
                    //  a third of the kernel's wave cycles on conv5's dx product)
                    const int kp = kc16 >> 3;
                    for (int e0 = tid; e0 < pieces; e0 += 8 * 256) {
                        uint4 tmp[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int e = min(e0 + 256 * u, pieces - 1);
                            const int n = e / kp, k8 = (e - n * kp) << 3;
                            tmp[u] = *reinterpret_cast<const uint4*>(a.B16 + (int64_t)(n0 + n) * a.Kp + k0 + k8);
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int e = e0 + 256 * u;
                            const int n = e / kp, k8 = (e - n * kp) << 3;
                            if (e < pieces) *reinterpret_cast<uint4*>(&Bt[n * LDS_STRIDE + k8]) = tmp[u];
                        }
                    }
                } else if (a.b_rs == 1) {     // k contiguous in memory: consecutive threads walk k
                    const int kp = kc16 >> 3;
                    for (int e = tid; e < pieces; e += 256) {
                        const int n = e / kp, k8 = (e - n * kp) << 3;
                        bf16x8 pk;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float v = 0.f;
                            if (n0 + n < a.N && k8 + j < kc) v = a.B[(int64_t)(k0 + k8 + j) + (int64_t)(n0 + n) * a.b_cs];
                            pk[j] = bf16_from_bits(__float_as_uint(v) >> 16);
                        }
                        *reinterpret_cast<bf16x8*>(&Bt[n * LDS_STRIDE + k8]) = pk;
                    }
                } else {                       // n contiguous (or general): consecutive threads walk n
                    for (int e = tid; e < pieces; e += 256) {
                        const int n = e % (NT * 32), k8 = (e / (NT * 32)) << 3;
                        bf16x8 pk;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float v = 0.f;
                            if (n0 + n < a.N && k8 + j < kc) v = a.B[(int64_t)(k0 + k8 + j) * a.b_rs + (int64_t)(n0 + n) * a.b_cs];
                            pk[j] = bf16_from_bits(__float_as_uint(v) >> 16);
                        }
                        *reinterpret_cast<bf16x8*>(&Bt[n * LDS_STRIDE + k8]) = pk;
                    }
                }
                __syncthreads();
                b_loaded = true;
            }
            for (int ks0 = 0; ks0 < kc16; ks0 += 16 * NPF) {
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                const int ks = ks0 + 16 * u;
                if (ks >= kc16) break;                 // (wave-uniform; only the last chunk of a ragged K)
                float x[8];
                const int kk = k0 + ks + 8 * h;
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = xn[u][j];
                // the fragment NPF k-steps on (the k-steps of consecutive chunks are consecutive: only the last chunk may be short),
                // into the slot just consumed; unconditional (clamped): a branch around the request makes the waitcnt pass drain it
                // (requested in PAIRS of k-steps: a 128-byte line of a row holds two k-steps, four load instructions touch it, and
                //  issued a k-step apart the line has to survive ~800 cycles in a 32 KB L1 that eight waves stream through)
#if SVNET_ROWS_ABLATE != 2
                if (NPF == 1) SVNET_LOAD_A(0, kk + 16);
                else if (u & 1) { SVNET_LOAD_A(u > 0 ? u - 1 : 0, kk - 16 + 16 * NPF); SVNET_LOAD_A(u, kk + 16 * NPF); }
#endif
                {
                    const float4 s0 = *reinterpret_cast<const float4*>(&ascale_l[ks + 8 * h]);
                    const float4 s1 = *reinterpret_cast<const float4*>(&ascale_l[ks + 8 * h + 4]);
                    x[0] *= s0.x; x[1] *= s0.y; x[2] *= s0.z; x[3] *= s0.w; x[4] *= s1.x; x[5] *= s1.y; x[6] *= s1.z; x[7] *= s1.w;
                }
                const Split3 s = split_frag(x);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Bt[(t * 32 + r) * LDS_STRIDE + ks + 8 * h]);
#if SVNET_ROWS_ABLATE == 4
                    acc[t][0] += (float)s.h[0] + (float)s.m[1] + (float)s.l[2] + (float)b[0];
#else
                    acc[t] = MFMA(s.h, b, acc[t]);
                    acc[t] = MFMA(s.m, b, acc[t]);
                    acc[t] = MFMA(s.l, b, acc[t]);
#endif
                }
            }
            }
        }
#undef SVNET_LOAD_A
        // ---- epilogue
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = n0 + t * 32 + r;
            if (col < a.N) {
                const float cs = a.alpha * (a.col_scale ? a.col_scale[col] : 1.f);
                const float bs = a.bias ? a.bias[col] : 0.f;
                uint64_t mw = ~0ull;
                if (a.mask && m0 < a.M) mw = a.mask[(m0 >> 6) * a.N + col];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rr = (i & 3) + 8 * (i >> 2) + 4 * h;
                    const int64_t row = m0 + rr;
                    if (row < a.M) {
                        float v = acc[t][i] * cs + bs;
                        if (!((mw >> (row & 63)) & 1ull)) v = 0.f;
                        colpart[t] += v;
                        float* dst = a.C + row * a.ldc + col;
#if SVNET_ROWS_ABLATE == 1
                        if (v == 12345.678f) *dst = v;
#else
                        if (a.accumulate) *dst += v;
                        else __builtin_nontemporal_store(v, dst);     // written once, read by a later kernel: all workgroups reach this point together
#endif
                    }
                }
            }
        }
    }
    if (a.col_sum) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float s = colpart[t] + __shfl_xor(colpart[t], 32, 64);
            const int col = n0 + t * 32 + r;
            if (h == 0 && col < a.N) atomicAdd(&svnet_slice_ptr(a.col_sum, a.N)[col], s);      // (sliced accumulator: svnet_hip.h)
        }
    }
}

// ---- rows kernel for the big shapes (M >= 4096, 16-byte aligned A rows, pre-packed B): the canonical LDS-tiled GEMM main loop.
// Workgroup tile 128 rows x 256 columns x 32 k, 4 waves as 2 (rows) x 2 (columns), 2 x 4 accumulator tiles of 32 x 32 per wave.
// BOTH operands go through double-buffered LDS: A arrives with full-line coalesced float4 loads (8 lanes per 128-byte row piece;
// the kernel above reads 32-byte pieces of 32 different rows per instruction), is scaled (a_scale) and parked as fp32 (three bf16
// planes of it would not leave room for two workgroups per CU; the waves split their fragments as before); B is copied from the packed bf16 table.  The loads of
// k tile t+1 are issued before the MFMAs of tile t and written to the other buffer after them: one barrier per k tile.
// Same exact arithmetic (3 bf16 x bf16 products per fp32 product, fp32 accumulation) and the same epilogue as mfma_rows_kernel.
constexpr int R2_BM = 128, R2_BN = 256, R2_BK = 32;
constexpr int R2_LDA = R2_BK + 4;                    // floats per LDS row of the A tile (144 bytes: 9 slots of 16 bytes)
constexpr int R2_LDB = R2_BK + 8;                    // bf16 per LDS row of the B tile (80 bytes: 5 slots)
constexpr int R2_A_BYTES = R2_BM * R2_LDA * 4, R2_B_BYTES = R2_BN * R2_LDB * 2;
constexpr int R2_BUF_BYTES = R2_A_BYTES + R2_B_BYTES;      // 38 912 bytes per buffer: two buffers, two workgroups per CU

// NP = bf16 pieces of B (1: B exact in bf16; 3: general fp32 B split exactly as Bh + Bm + Bl, packed one after the other in B16 with
// a.b_piece elements between them).  The pipeline stage is (k tile, piece): the B tile changes every stage, the A tile every NP stages -
// A is read from HBM once for all three pieces, C is written once.
// AV: A rows 16-byte aligned with K % 4 == 0 (float4 loads); else four clamped scalar loads per float4 slot (e.g. K = 127 feature rows).
template <int NP, bool AV = true>
__global__ __launch_bounds__(256, 2) void mfma_rows2_kernel(RowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char L2s[];       // [A fp32 tile x 2 | B bf16 tile x 2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                            // this wave: rows 64 wm .. +63, columns 128 wn .. +127 of the tile
    const int64_t m0 = (int64_t)blockIdx.x * R2_BM;
    const int n0 = blockIdx.y * R2_BN;
    const int nkt = (a.K + R2_BK - 1) / R2_BK;
    const int nst = nkt * NP;

    // staging assignments: A - float4 number q of the tile (4 per thread): row = q / 8, k = 4 (q % 8); B - uint4 number q: column = q / 4, k = 8 (q % 4)
    float4 areg[4];
    uint4 breg[4];
    float4 sreg = make_float4(1.f, 1.f, 1.f, 1.f);
#define SVNET_R2_LOAD_A(KT)                                                                                        \
    do {                                                                                                           \
        const int k0_ = (KT) * R2_BK;                                                                              \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                            \
            const int q = tid + 256 * u, row = q >> 3, k4 = (q & 7) << 2;                                          \
            const int64_t gr = min(m0 + row, a.M - 1);                                                             \
            if (AV) {                                                                                              \
                const int kk = min(k0_ + k4, a.K - 4);             /* (K % 4 == 0: a whole float4 is in range, or clamped and zeroed) */ \
                areg[u] = *reinterpret_cast<const float4*>(a.A + gr * a.lda + kk);                                 \
                if (k0_ + k4 >= a.K) areg[u] = make_float4(0.f, 0.f, 0.f, 0.f);                                    \
            } else {                                                                                               \
                const float* ar_ = a.A + gr * a.lda;                                                               \
                const int kb = k0_ + k4, kl = a.K - 1;                                                             \
                const float e0 = ar_[min(kb, kl)], e1 = ar_[min(kb + 1, kl)], e2 = ar_[min(kb + 2, kl)], e3 = ar_[min(kb + 3, kl)]; \
                areg[u] = make_float4(kb < a.K ? e0 : 0.f, kb + 1 < a.K ? e1 : 0.f, kb + 2 < a.K ? e2 : 0.f, kb + 3 < a.K ? e3 : 0.f); \
            }                                                                                                      \
        }                                                                                                          \
        if (a.a_scale) {                                                                                           \
            const int kb = k0_ + ((tid & 7) << 2), kl = a.K - 1;                                                   \
            if (AV) sreg = *reinterpret_cast<const float4*>(a.a_scale + min(kb, a.K - 4));                         \
            else sreg = make_float4(a.a_scale[min(kb, kl)], a.a_scale[min(kb + 1, kl)], a.a_scale[min(kb + 2, kl)], a.a_scale[min(kb + 3, kl)]); \
        }                                                                                                          \
    } while (0)
#define SVNET_R2_LOAD_B(KT, PC)                                                                                    \
    do {                                                                                                           \
        const int k0_ = (KT) * R2_BK;                                                                              \
        const uint16_t* bp_ = a.B16 + (int64_t)(PC) * a.b_piece;                                                   \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                            \
            const int q = tid + 256 * u, col = q >> 2, k8 = (q & 3) << 3;                                          \
            breg[u] = *reinterpret_cast<const uint4*>(bp_ + (int64_t)(n0 + col) * a.Kp + min(k0_ + k8, a.Kp - 8)); \
            if (k0_ + k8 >= a.Kp) breg[u] = make_uint4(0u, 0u, 0u, 0u);                                            \
        }                                                                                                          \
    } while (0)
#define SVNET_R2_STORE_A(BUFI)                                                                                     \
    do {                                                                                                           \
        float* abuf_ = reinterpret_cast<float*>(L2s + (BUFI) * R2_A_BYTES);                                        \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                            \
            const int q = tid + 256 * u, row = q >> 3, k4 = (q & 7) << 2;                                          \
            float4 v = areg[u];                                                                                    \
            v.x *= sreg.x; v.y *= sreg.y; v.z *= sreg.z; v.w *= sreg.w;       /* the per-k scale of the A operand (ones without one) */ \
            *reinterpret_cast<float4*>(abuf_ + row * R2_LDA + k4) = v;                                             \
        }                                                                                                          \
    } while (0)
#define SVNET_R2_STORE_B(BUFI)                                                                                     \
    do {                                                                                                           \
        __bf16* bbuf_ = reinterpret_cast<__bf16*>(L2s + 2 * R2_A_BYTES + (BUFI) * R2_B_BYTES);                     \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                            \
            const int q = tid + 256 * u, col = q >> 2, k8 = (q & 3) << 3;                                          \
            *reinterpret_cast<uint4*>(bbuf_ + col * R2_LDB + k8) = breg[u];                                        \
        }                                                                                                          \
    } while (0)

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    SVNET_R2_LOAD_A(0);
    SVNET_R2_LOAD_B(0, 0);
    SVNET_R2_STORE_A(0);
    SVNET_R2_STORE_B(0);
    __syncthreads();
    int kt = 0, pc = 0;                                                 // k tile and piece of the stage being multiplied
    for (int st = 0; st < nst; ++st) {
        const float* abuf = reinterpret_cast<const float*>(L2s + (kt & 1) * R2_A_BYTES);
        const __bf16* bbuf = reinterpret_cast<const __bf16*>(L2s + 2 * R2_A_BYTES + (st & 1) * R2_B_BYTES);
        const int npc = (pc + 1 == NP) ? 0 : pc + 1, nkt_ = (pc + 1 == NP) ? kt + 1 : kt;      // the next stage
        const bool more = st + 1 < nst, new_a = more && npc == 0;      // (uniform)
        if (more) SVNET_R2_LOAD_B(nkt_, npc);                           // in flight across this stage's MFMAs
        if (new_a) SVNET_R2_LOAD_A(nkt_);
#pragma unroll
        for (int ks = 0; ks < R2_BK; ks += 16) {
            Split3 sa[2];
            bf16x8 bb[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* ap = abuf + (64 * wm + 32 * i + r) * R2_LDA + ks + 8 * h;
                const float4 v0 = *reinterpret_cast<const float4*>(ap), v1 = *reinterpret_cast<const float4*>(ap + 4);
                const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                sa[i] = split_frag(x);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) bb[j] = *reinterpret_cast<const bf16x8*>(bbuf + (128 * wn + 32 * j + r) * R2_LDB + ks + 8 * h);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = MFMA(sa[i].h, bb[j], acc[i][j]);
                    acc[i][j] = MFMA(sa[i].m, bb[j], acc[i][j]);
                    acc[i][j] = MFMA(sa[i].l, bb[j], acc[i][j]);
                }
        }
        if (more) SVNET_R2_STORE_B((st + 1) & 1);
        if (new_a) SVNET_R2_STORE_A(nkt_ & 1);
        __syncthreads();
        kt = nkt_; pc = npc;
    }
#undef SVNET_R2_LOAD_A
#undef SVNET_R2_LOAD_B
#undef SVNET_R2_STORE_A
#undef SVNET_R2_STORE_B
    // ---- epilogue (as in mfma_rows_kernel): D reg e of a tile: row (e & 3) + 8 (e >> 2) + 4 h, column r
    float colpart[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + 128 * wn + 32 * j + r;
        if (col < a.N) {
            const float cs = a.alpha * (a.col_scale ? a.col_scale[col] : 1.f);
            const float bs = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t mt = m0 + 64 * wm + 32 * i;                 // first row of this 32-row tile
                uint64_t mw = ~0ull;
                if (a.mask && mt < a.M) mw = a.mask[(mt >> 6) * a.N + col];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rr = (e & 3) + 8 * (e >> 2) + 4 * h;
                    const int64_t row = mt + rr;
                    if (row < a.M) {
                        float v = acc[i][j][e] * cs + bs;
                        if (!((mw >> (row & 63)) & 1ull)) v = 0.f;
                        colpart[j] += v;
                        float* dst = a.C + row * a.ldc + col;
                        if (a.accumulate) *dst += v;
                        else __builtin_nontemporal_store(v, dst);
                    }
                }
            }
        }
    }
    if (a.col_sum) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sum = colpart[j] + __shfl_xor(colpart[j], 32, 64);
            const int col = n0 + 128 * wn + 32 * j + r;
            if (h == 0 && col < a.N) atomicAdd(&svnet_slice_ptr(a.col_sum, a.N)[col], sum);    // (sliced accumulator: svnet_hip.h)
        }
    }
}

// ---- many rows x GENERAL fp32 weights in one launch: C[M x N] = A[M x K] . (Bh + Bm + Bl), A split ONCE per element.
// mfma_rows2_kernel<3> reads the fp32 A tile from LDS and splits it into its three bf16 pieces in the wave that multiplies - every A
// element was split six times (two column waves x three B pieces), ~130 vector instructions per 16-wide k-step beside its 24 MFMAs at two
// waves per SIMD.  Here the threads that STAGE the A tile split it (16 values per thread and k tile) and leave three bf16 tiles in LDS;
// the multiplying waves only read fragments.  Six bf16 products per fp32 product (the leading terms of the 3 x 3 expansion, see the stage
// macro): the result is within one fp32 rounding per product of the exact product sum, accumulated in fp32 - fp32-GEMM accuracy.
//   tile 128 rows x (64 NJ) columns x 32 k; 2 x 2 waves, a wave = 64 rows x 32 NJ columns; stage = (k tile, B piece)
//   LDS: A pieces [3][128][40] bf16 (single buffer: it changes every third stage, behind one extra barrier) + B [2][64 NJ][40] bf16
//   W = floats per A load (4: rows 16-byte aligned, K % 4 == 0; 2: 8-byte aligned, K even; 1: anything - consecutive lanes then walk a row)
constexpr int R3_LDH = R2_BK + 8;                          // bf16 per LDS row of an A piece tile (80 bytes)
constexpr int R3_A_BYTES = 3 * R2_BM * R3_LDH * 2;         // 30 720
template <int NJ, int W>
__global__ __launch_bounds__(256, 2) void mfma_rows3_kernel(RowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char L3s[];
    constexpr int BN = 64 * NJ, B_BYTES = BN * R2_LDB * 2;
    constexpr int NL = 16 / W, GPR = R2_BK / W;                      // A loads per thread and k tile; load groups per row
    uint16_t* apc = reinterpret_cast<uint16_t*>(L3s);                // [3][128][R3_LDH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.x * R2_BM;
    const int n0 = blockIdx.y * BN;
    const int nkt = (a.K + R2_BK - 1) / R2_BK;
    const int nst = nkt * 3;
    const int npad = (int)(a.b_piece / a.Kp);                         // packed columns of a piece (whole column tiles of the packing)

    float areg[NL][W];
    uint4 breg[NJ];
#define SVNET_R3_LOAD_A(KT)                                                                                        \
    do {                                                                                                           \
        const int k0_ = (KT) * R2_BK;                                                                              \
        _Pragma("unroll") for (int u = 0; u < NL; ++u) {                                                           \
            const int q = tid + 256 * u, row = q / GPR, kk = k0_ + (q % GPR) * W;                                  \
            const float* ar_ = a.A + min(m0 + row, a.M - 1) * a.lda;                                               \
            if (W == 4) {                                                                                          \
                const float4 v = *reinterpret_cast<const float4*>(ar_ + min(kk, a.K - 4));                        \
                areg[u][0] = v.x; areg[u][1 % W] = v.y; areg[u][2 % W] = v.z; areg[u][3 % W] = v.w;                \
            } else if (W == 2) {                                                                                   \
                const float2 v = *reinterpret_cast<const float2*>(ar_ + min(kk, a.K - 2));                        \
                areg[u][0] = v.x; areg[u][1 % W] = v.y;                                                            \
            } else {                                                                                               \
                areg[u][0] = ar_[min(kk, a.K - 1)];                                                                \
            }                                                                                                      \
            if (kk >= a.K) { _Pragma("unroll") for (int e = 0; e < W; ++e) areg[u][e] = 0.f; }                     \
        }                                                                                                          \
    } while (0)
#define SVNET_R3_LOAD_B(KT, PC)                                                                                    \
    do {                                                                                                           \
        const int k0_ = (KT) * R2_BK;                                                                              \
        const uint16_t* bp_ = a.B16 + (int64_t)(PC) * a.b_piece;                                                   \
        _Pragma("unroll") for (int u = 0; u < NJ; ++u) {                                                           \
            const int q = tid + 256 * u, col = q >> 2, k8 = (q & 3) << 3;                                          \
            breg[u] = *reinterpret_cast<const uint4*>(bp_ + (int64_t)min(n0 + col, npad - 1) * a.Kp + min(k0_ + k8, a.Kp - 8)); \
            if (k0_ + k8 >= a.Kp) breg[u] = make_uint4(0u, 0u, 0u, 0u);                                            \
        }                                                                                                          \
    } while (0)
#define SVNET_R3_STORE_A()                                                                                         \
    do {                                                                                                           \
        _Pragma("unroll") for (int u = 0; u < NL; ++u) {                                                           \
            const int q = tid + 256 * u, row = q / GPR, kc = (q % GPR) * W;                                        \
            uint32_t hh[W], mm[W], ll[W];                                                                          \
            _Pragma("unroll") for (int e = 0; e < W; ++e) split3(areg[u][e], hh[e], mm[e], ll[e]);                 \
            uint16_t* o_ = apc + row * R3_LDH + kc;                                                                \
            if (W == 4) {                                                                                          \
                *reinterpret_cast<uint2*>(o_) = make_uint2(hh[0] | (hh[1 % W] << 16), hh[2 % W] | (hh[3 % W] << 16)); \
                *reinterpret_cast<uint2*>(o_ + R2_BM * R3_LDH) = make_uint2(mm[0] | (mm[1 % W] << 16), mm[2 % W] | (mm[3 % W] << 16)); \
                *reinterpret_cast<uint2*>(o_ + 2 * R2_BM * R3_LDH) = make_uint2(ll[0] | (ll[1 % W] << 16), ll[2 % W] | (ll[3 % W] << 16)); \
            } else if (W == 2) {                                                                                   \
                *reinterpret_cast<uint32_t*>(o_) = hh[0] | (hh[1 % W] << 16);                                      \
                *reinterpret_cast<uint32_t*>(o_ + R2_BM * R3_LDH) = mm[0] | (mm[1 % W] << 16);                     \
                *reinterpret_cast<uint32_t*>(o_ + 2 * R2_BM * R3_LDH) = ll[0] | (ll[1 % W] << 16);                 \
            } else {                                                                                               \
                o_[0] = (uint16_t)hh[0]; o_[R2_BM * R3_LDH] = (uint16_t)mm[0]; o_[2 * R2_BM * R3_LDH] = (uint16_t)ll[0]; \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)
#define SVNET_R3_STORE_B(BUFI)                                                                                     \
    do {                                                                                                           \
        __bf16* bbuf_ = reinterpret_cast<__bf16*>(L3s + R3_A_BYTES + (BUFI) * B_BYTES);                            \
        _Pragma("unroll") for (int u = 0; u < NJ; ++u) {                                                           \
            const int q = tid + 256 * u, col = q >> 2, k8 = (q & 3) << 3;                                          \
            *reinterpret_cast<uint4*>(bbuf_ + col * R2_LDB + k8) = breg[u];                                        \
        }                                                                                                          \
    } while (0)

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    SVNET_R3_LOAD_A(0);
    SVNET_R3_LOAD_B(0, 0);
    SVNET_R3_STORE_A();
    SVNET_R3_STORE_B(0);
    __syncthreads();
    // One stage = (k tile KT, B piece PC), the piece a compile-time constant: against Bh all three A pieces are multiplied, against Bm the
    // upper two, against Bl the first - the six leading terms of (Ah + Am + Al)(Bh + Bm + Bl); the three dropped ones (Am Bl, Al Bm, Al Bl) are
    // below 2^-24 of the product, i.e. below the rounding of ONE fp32 multiply (the fp32 x fp32 weight-gradient kernels make the same cut).
#define SVNET_R3_STAGE(KT, PC)                                                                                     \
    do {                                                                                                           \
        const int st_ = 3 * (KT) + (PC);                                                                           \
        const __bf16* bbuf = reinterpret_cast<const __bf16*>(L3s + R3_A_BYTES + (st_ & 1) * B_BYTES);              \
        const int nkt_ = ((PC) == 2) ? (KT) + 1 : (KT);                                                            \
        const bool more = st_ + 1 < nst, new_a = more && (PC) == 2;      /* (uniform) */                           \
        if (more) SVNET_R3_LOAD_B(nkt_, ((PC) + 1) % 3);                  /* in flight across this stage's MFMAs */ \
        if ((PC) == 0 && (KT) + 1 < nkt) SVNET_R3_LOAD_A((KT) + 1);       /* the next k tile of A: three stages ahead (HBM latency) */ \
        _Pragma("unroll") for (int ks = 0; ks < R2_BK; ks += 16) {                                                 \
            bf16x8 af[2][3 - (PC)], bb[NJ];                                                                        \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                          \
                _Pragma("unroll") for (int pa = 0; pa < 3 - (PC); ++pa)                                            \
                    af[i][pa] = *reinterpret_cast<const bf16x8*>(apc + (pa * R2_BM + 64 * wm + 32 * i + r) * R3_LDH + ks + 8 * h); \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                         \
                bb[j] = *reinterpret_cast<const bf16x8*>(bbuf + (32 * NJ * wn + 32 * j + r) * R2_LDB + ks + 8 * h); \
            _Pragma("unroll") for (int pa = 0; pa < 3 - (PC); ++pa)                                                \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                      \
                    _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = MFMA(af[i][pa], bb[j], acc[i][j]);  \
        }                                                                                                          \
        if (more) SVNET_R3_STORE_B((st_ + 1) & 1);                                                                 \
        if (new_a) {                                                                                               \
            __syncthreads();                                            /* every wave has read the A pieces of this k tile */ \
            SVNET_R3_STORE_A();                                                                                    \
        }                                                                                                          \
        __syncthreads();                                                                                           \
    } while (0)
    for (int kt = 0; kt < nkt; ++kt) {
        SVNET_R3_STAGE(kt, 0);
        SVNET_R3_STAGE(kt, 1);
        SVNET_R3_STAGE(kt, 2);
    }
#undef SVNET_R3_STAGE
#undef SVNET_R3_LOAD_A
#undef SVNET_R3_LOAD_B
#undef SVNET_R3_STORE_A
#undef SVNET_R3_STORE_B
    // ---- epilogue: D reg e of a tile: row (e & 3) + 8 (e >> 2) + 4 h, column r; alpha, bias, accumulate
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = n0 + 32 * NJ * wn + 32 * j + r;
        if (col < a.N) {
            const float bs = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t mt = m0 + 64 * wm + 32 * i;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = mt + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (row < a.M) {
                        const float v = acc[i][j][e] * a.alpha + bs;
                        float* dst = a.C + row * a.ldc + col;
                        if (a.accumulate) *dst += v;
                        else __builtin_nontemporal_store(v, dst);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ tn kernel
struct TnArgs {
    const float* A; int64_t lda;        // [M, P] fp32 rows (p contiguous)
    const float* B; int64_t ldb;        // BMODE 0: [M, Q] fp32 rows
    const uint64_t* b_sign; const uint64_t* b_nz;   // BMODE 1: row-sliced planes [ceil(M/64)][Q]
    float* C; int64_t c_ps, c_qs;       // out(p,q) at C[p*c_ps + q*c_qs], accumulated with atomics (pre-zeroed)
    int64_t M; int P, Q;
    int64_t rows_per_block;             // multiple of 256
    int ptiles_per_block;               // p tiles (32 wide) handled by the 4 waves of a workgroup: 1, 2 or 4
    int lds_reduce;                     // combine the row-splitting waves in LDS before the global atomics
    float alpha;
    uint32_t qmask;                     // ternary B: bit t = q tile t (32 columns) may be non-zero; cleared tiles are skipped
    uint32_t qlist;                     // != 0: the workgroup's NQ tiles are the tile ids packed here, 4 bits each (+1; 0 = none)
    // AFFINE A (fused edge layers, n16 != nullptr): A(m,p) = dL/dy_pre of edge row m, channel p, is not read but recomputed from what
    // the forward kept:  cs[p]*g - (alpha[p] + beta[p]*n16[m,p]),  g = gy[point(m), p] on the point's pooled edge (slot == m % k), else 0
    const int16_t* n16; const float* gy; const uint8_t* smax; const uint8_t* smin;   // [M,P], [M/k,P] x 3
    const float* chc;                   // [cs | alpha | beta | scale | pooled-is-max], P each (edgeblock_bwd_coeffs)
    int kk; uint32_t kmagic; int64_t npts;
};
// bit t of the tile mask, without shifting a 32-bit value by >= 32 (Q > 1024 has more than 32 column tiles): an all-ones mask
// means "every tile", a partial mask covers tiles 0..31 only (enforced on the host)
__device__ __forceinline__ bool qtile_live(uint32_t qmask, int qt) {
    return qmask == 0xFFFFFFFFu || (qt < 32 && ((qmask >> qt) & 1u));
}


// NQ 32-wide q tiles per workgroup (blockIdx.z picks the group).  The 4 waves cover `ptw` p tiles (1, 2 or 4 per
// workgroup); when P is narrow (ptw < 4) the spare waves split the workgroup's row range instead of idling.
template <int NQ, int BMODE>
__global__ __launch_bounds__(256, 2) void mfma_tn_kernel(TnArgs a) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // row ranges stay in SGPRs
    const int r = lane & 31, h = lane >> 5;
    const int ptw = a.ptiles_per_block;                      // 1, 2 or 4
    const int p0 = (blockIdx.y * ptw + (wave % ptw)) * 32;
    const int q0 = blockIdx.z * (NQ * 32);
    if (p0 >= a.P) return;  // wave-uniform; no barriers in this kernel
    const int nsub = 4 / ptw, sub = wave / ptw;
    const int64_t rows_sub = ((a.rows_per_block / nsub + 63) >> 6) << 6;   // multiple of 64
    const int64_t mb = (int64_t)blockIdx.x * a.rows_per_block + (int64_t)sub * rows_sub;
    const int64_t me = min(min(a.M, (int64_t)(blockIdx.x + 1) * a.rows_per_block), mb + rows_sub);

    f32x16 acc[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // fp32 x fp32 (BMODE 0 is the only mode left here; ternary B runs in mfma_tn_tern_kernel).  The A fragment and the NQ B
    // fragments of the next 16-row step are requested before the MFMAs of the current one.  Columns past P / Q are computed
    // from clamped (valid) addresses and never stored, so only rows past the range's end need zeroing, and the k-step has no
    // per-tile branches (one basic block: the splits of one tile overlap the MFMAs of another).
    {
        const int64_t mlast = a.M - 1;
        const int pc = min(p0 + r, a.P - 1);
        int qc[NQ];
#pragma unroll
        for (int t = 0; t < NQ; ++t) qc[t] = min(q0 + t * 32 + r, a.Q - 1);
        float xn[8], yn[NQ][8];
#define SVNET_TN0_LOAD(MROW)                                                                                  \
    do {                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                        \
            const int64_t row_ = min((MROW) + j, mlast);                                                      \
            xn[j] = a.A[row_ * a.lda + pc];                                                                   \
            _Pragma("unroll") for (int t = 0; t < NQ; ++t) yn[t][j] = a.B[row_ * a.ldb + qc[t]];              \
        }                                                                                                     \
    } while (0)
        if (mb < me) SVNET_TN0_LOAD(mb + 8 * h);
        for (int64_t m16 = mb; m16 < me; m16 += 16) {
            float x[8], y[NQ][8];
            const int lim = (int)min((int64_t)16, me - m16) - 8 * h;   // rows of this lane's slice inside the range (32-bit compares)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = j < lim;
                x[j] = ok ? xn[j] : 0.f;
#pragma unroll
                for (int t = 0; t < NQ; ++t) y[t][j] = yn[t][j];   // (x = 0 is enough to drop the row)
            }
            if (m16 + 16 < me) SVNET_TN0_LOAD(m16 + 16 + 8 * h);
            const Split3 sa = split_frag(x);
#pragma unroll
            for (int t = 0; t < NQ; ++t) {
                const Split3 sb = split_frag(y[t]);
                acc[t] = MFMA(sa.h, sb.h, acc[t]);
                acc[t] = MFMA(sa.h, sb.m, acc[t]);
                acc[t] = MFMA(sa.m, sb.h, acc[t]);
                acc[t] = MFMA(sa.h, sb.l, acc[t]);
                acc[t] = MFMA(sa.l, sb.h, acc[t]);
                acc[t] = MFMA(sa.m, sb.m, acc[t]);
            }
        }
#undef SVNET_TN0_LOAD
    }
    // Waves that split the row range of one p tile (narrow P) first combine their partial tiles in LDS: the output
    // matrix is tiny and shared by the whole grid, and same-address float atomics serialise at the memory side.
    if (a.lds_reduce) {   // uniform; every wave of the workgroup is live in this configuration
        extern __shared__ float tnred[];                     // [nsub-1][NQ][16][64]
        if (sub > 0) {
#pragma unroll
            for (int t = 0; t < NQ; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) tnred[(((sub - 1) * ptw + (wave % ptw)) * NQ + t) * 1024 + i * 64 + lane] = acc[t][i];
        }
        __syncthreads();
        if (sub > 0) return;
        for (int s2 = 1; s2 < nsub; ++s2)
#pragma unroll
            for (int t = 0; t < NQ; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] += tnred[(((s2 - 1) * ptw + (wave % ptw)) * NQ + t) * 1024 + i * 64 + lane];
    }
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        const int q = q0 + t * 32 + r;  // D col = lane & 31  <-> B operand column (q)
        if (q < a.Q) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int pp = p0 + (i & 3) + 8 * (i >> 2) + 4 * h;  // D row <-> A operand row (p)
                if (pp < a.P) atomicAdd(&a.C[(int64_t)pp * a.c_ps + (int64_t)q * a.c_qs], acc[t][i] * a.alpha);
            }
        }
    }
}

// ---- tn, fp32 x fp32, LDS-tiled: out[P x Q] += A[rows, P]^T . B[rows, Q] over a workgroup's row range.
// mfma_tn_kernel above splits every operand value in the wave that multiplies it: at four p tiles per workgroup each B value was split
// four times per workgroup (and again by every workgroup of another p group) - ~320 vector instructions per 16-row step beside 24 MFMAs.
// Here the 256 threads stage a 32-row slab of both operands ONCE: thread (column c, row group g) loads 8 rows of its column (coalesced
// across the columns), splits them and leaves three bf16x8 fragments per operand in LDS, [piece][column][row] - exactly the layout the
// MFMA operands are read in.  2 x 2 waves, a wave = (32 IP) x (32 JQ) outputs; six bf16 products per fp32 product (h.h, h.m, m.h, h.l,
// l.h, m.m: the dropped terms are below 2^-24 of the product, as in the kernel above).  The next slab travels global -> registers while
// the current one is multiplied (requests TWO slabs ahead, a second register set, measured no faster: 536 against 516 us on
// [512 x 2044], 185 against 166 on [170 x 340]).  Output: float atomics into the pre-zeroed / accumulated C, one per element and row split.
#ifndef SVNET_TN2_ABLATE
#define SVNET_TN2_ABLATE 0      /* diagnostic builds only (results WRONG): 1 no atomics, 2 no global loads after the first slab, 3 no MFMAs */
#endif
constexpr int T2_MK = 32;                                   // rows per stage
constexpr int T2_LDM = T2_MK + 8;                           // bf16 per LDS row (80 bytes)
template <int IP, int JQ>
__global__ __launch_bounds__(256, 2) void mfma_tn2_kernel(TnArgs a) {
    constexpr int TP = 64 * IP, TQ = 64 * JQ;               // output tile of the workgroup
    constexpr int NA = 4 * TP / 256, NB = 4 * TQ / 256;     // (column, 8-row group) items per thread: 1 or 2
    __shared__ __attribute__((aligned(16))) uint16_t ap[3 * TP * T2_LDM];
    __shared__ __attribute__((aligned(16))) uint16_t bp[3 * TQ * T2_LDM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wp = wave >> 1, wq = wave & 1;
    const int p0 = blockIdx.y * TP, q0 = blockIdx.z * TQ;
    const int64_t mb = (int64_t)blockIdx.x * a.rows_per_block;
    const int64_t me = min(a.M, mb + a.rows_per_block);
    if (mb >= me) return;                                    // (uniform)
    const int64_t mlast = a.M - 1;

    const int ca = tid % TP, cb = tid % TQ;
    const int ga0 = __builtin_amdgcn_readfirstlane(tid / TP), gb0 = __builtin_amdgcn_readfirstlane(tid / TQ);   // 8-row group: wave-uniform
    const int pcl = min(p0 + ca, a.P - 1), qcl = min(q0 + cb, a.Q - 1);      // clamped: columns past P / Q are computed and never stored
    const uint32_t offa = 4u * (uint32_t)pcl, offb = 4u * (uint32_t)qcl;     // lane byte offsets: the row bases stay in SGPRs
    float xa[NA][8], xb[NB][8];
// whole slab inside the range: wave-uniform row base + lane offset, no clamps; the last slab of a range: clamped rows, zeros past the end
#define SVNET_T2_LOAD(M0)                                                                                      \
    do {                                                                                                      \
        if ((M0) + T2_MK <= me) {                                                                             \
            _Pragma("unroll") for (int u = 0; u < NA; ++u) {                                                   \
                const float* rp_ = a.A + ((M0) + 8 * (ga0 + u * (256 / TP))) * a.lda;                          \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) xa[u][j] = ld_f32_sbase(rp_ + j * a.lda, offa);  \
            }                                                                                                 \
            _Pragma("unroll") for (int u = 0; u < NB; ++u) {                                                   \
                const float* rp_ = a.B + ((M0) + 8 * (gb0 + u * (256 / TQ))) * a.ldb;                          \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) xb[u][j] = ld_f32_sbase(rp_ + j * a.ldb, offb);  \
            }                                                                                                 \
        } else {                                                                                              \
            _Pragma("unroll") for (int u = 0; u < NA; ++u)                                                     \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                \
                    const int64_t row_ = (M0) + 8 * (ga0 + u * (256 / TP)) + j;                                \
                    const float v_ = ld_f32_sbase(a.A + min(row_, mlast) * a.lda, offa);                       \
                    xa[u][j] = row_ < me ? v_ : 0.f;                                                           \
                }                                                                                             \
            _Pragma("unroll") for (int u = 0; u < NB; ++u)                                                     \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                \
                    const int64_t row_ = (M0) + 8 * (gb0 + u * (256 / TQ)) + j;                                \
                    const float v_ = ld_f32_sbase(a.B + min(row_, mlast) * a.ldb, offb);                       \
                    xb[u][j] = row_ < me ? v_ : 0.f;                                                           \
                }                                                                                             \
        }                                                                                                     \
    } while (0)
#define SVNET_T2_STORE1(X, DST, COL, G, TW)                                                                    \
    do {                                                                                                      \
        const Split3 s_ = split_frag(X);                                                                      \
        uint16_t* o_ = (DST) + (COL) * T2_LDM + 8 * (G);                                                       \
        *reinterpret_cast<bf16x8*>(o_) = s_.h;                                                                \
        *reinterpret_cast<bf16x8*>(o_ + (TW) * T2_LDM) = s_.m;                                                 \
        *reinterpret_cast<bf16x8*>(o_ + 2 * (TW) * T2_LDM) = s_.l;                                             \
    } while (0)

    f32x16 acc[IP][JQ];
#pragma unroll
    for (int i = 0; i < IP; ++i)
#pragma unroll
        for (int j = 0; j < JQ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    SVNET_T2_LOAD(mb);
    for (int64_t m0 = mb; m0 < me; m0 += T2_MK) {
#pragma unroll
        for (int u = 0; u < NA; ++u) SVNET_T2_STORE1(xa[u], ap, ca, ga0 + u * (256 / TP), TP);
#pragma unroll
        for (int u = 0; u < NB; ++u) SVNET_T2_STORE1(xb[u], bp, cb, gb0 + u * (256 / TQ), TQ);
        __syncthreads();
#if SVNET_TN2_ABLATE != 2
        if (m0 + T2_MK < me) SVNET_T2_LOAD(m0 + T2_MK);               // in flight across the MFMAs
#endif
#pragma unroll
        for (int ks = 0; ks < T2_MK; ks += 16) {
            bf16x8 fa[IP][3], fb[JQ][3];
#pragma unroll
            for (int i = 0; i < IP; ++i)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    fa[i][pc] = *reinterpret_cast<const bf16x8*>(ap + (pc * TP + 32 * (IP * wp + i) + r) * T2_LDM + ks + 8 * h);
#pragma unroll
            for (int j = 0; j < JQ; ++j)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    fb[j][pc] = *reinterpret_cast<const bf16x8*>(bp + (pc * TQ + 32 * (JQ * wq + j) + r) * T2_LDM + ks + 8 * h);
#pragma unroll
            for (int i = 0; i < IP; ++i)
#pragma unroll
                for (int j = 0; j < JQ; ++j) {
#if SVNET_TN2_ABLATE == 3
                    acc[i][j][0] += (float)fa[i][0][0] + (float)fb[j][0][0] + (float)fa[i][1][1] + (float)fb[j][1][1] + (float)fa[i][2][2] + (float)fb[j][2][2];
#else
                    acc[i][j] = MFMA(fa[i][0], fb[j][0], acc[i][j]);
                    acc[i][j] = MFMA(fa[i][0], fb[j][1], acc[i][j]);
                    acc[i][j] = MFMA(fa[i][1], fb[j][0], acc[i][j]);
                    acc[i][j] = MFMA(fa[i][0], fb[j][2], acc[i][j]);
                    acc[i][j] = MFMA(fa[i][2], fb[j][0], acc[i][j]);
                    acc[i][j] = MFMA(fa[i][1], fb[j][1], acc[i][j]);
#endif
                }
        }
        __syncthreads();
    }
#undef SVNET_T2_LOAD
#undef SVNET_T2_STORE1
#pragma unroll
    for (int j = 0; j < JQ; ++j) {
        const int q = q0 + 32 * (JQ * wq + j) + r;           // D column = lane & 31 <-> B operand column (q)
        if (q < a.Q) {
#pragma unroll
            for (int i = 0; i < IP; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int pp = p0 + 32 * (IP * wp + i) + (e & 3) + 8 * (e >> 2) + 4 * h;   // D row <-> A operand column (p)
#if SVNET_TN2_ABLATE == 1
                    if (pp < a.P && acc[i][j][e] == 123.f) a.C[(int64_t)pp * a.c_ps + (int64_t)q * a.c_qs] = 1.f;
#else
                    if (pp < a.P) atomicAdd(&a.C[(int64_t)pp * a.c_ps + (int64_t)q * a.c_qs], acc[i][j][e] * a.alpha);
#endif
                }
        }
    }
}

// ---- tn with a TERNARY B operand (row-sliced planes): the weight-gradient product GX = x_b^T . dy of every binarized layer.
// Same tiling as mfma_tn_kernel<NQ, 1>, but (a) the 8-row slices of the planes are expanded to bf16 fragments through two
// 256-entry LDS tables (magnitude from the non-zero byte, sign bit from the negative byte) instead of 48 VALU operations
// per fragment, (b) the A fragment of the next k-step and the plane words of the next 64-row block are loaded before the
// MFMAs of the current one, (c) <= 256 VGPRs so that two waves per SIMD overlap each other's loads.
template <int NQ, bool AFF = false>
__global__ __launch_bounds__(256, 2) void mfma_tn_tern_kernel(TnArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t lut_mag[256 * 4], lut_neg[256 * 4];   // bf16x8 per byte value
    {
        const int b = threadIdx.x;   // 256 threads: one table row each
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t lo = (b >> (2 * w)) & 1u, hi = (b >> (2 * w + 1)) & 1u;
            lut_mag[b * 4 + w] = lo * 0x3F80u | hi * 0x3F800000u;
            lut_neg[b * 4 + w] = lo * 0x8000u | hi * 0x80000000u;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // row ranges stay in SGPRs
    const int r = lane & 31, h = lane >> 5;
    const int ptw = a.ptiles_per_block;                      // 1, 2 or 4
    const int p0 = (blockIdx.y * ptw + (wave % ptw)) * 32;
    const int q0 = blockIdx.z * (NQ * 32);
    // column tile handled as slot t: consecutive tiles of this z group, or (compacted launch) the t-th tile in use
#define SVNET_QT(t) (a.qlist ? (int)((a.qlist >> (4 * (t))) & 15u) - 1 : (q0 >> 5) + (t))
    const bool live = p0 < a.P;                              // wave-uniform
    const int nsub = 4 / ptw, sub = wave / ptw;
    const int64_t rows_sub = ((a.rows_per_block / nsub + 63) >> 6) << 6;   // multiple of 64
    const int64_t mb = (int64_t)blockIdx.x * a.rows_per_block + (int64_t)sub * rows_sub;
    const int64_t me = min(min(a.M, (int64_t)(blockIdx.x + 1) * a.rows_per_block), mb + rows_sub);
    const int p = min(p0 + r, a.P - 1);                      // clamped: columns past P are computed and never stored

    f32x16 acc[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    if (live && mb < me) {
        const int64_t mlast = a.M - 1;
        uint64_t wsg[NQ], wnz[NQ], nsg[NQ], nnz[NQ];
        uint64_t slot_live[NQ];          // all ones when slot t holds a tile that exists and is in use (wave-uniform)
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            const int qt = SVNET_QT(t);
            slot_live[t] = (qt >= 0 && qt * 32 < a.Q && qtile_live(a.qmask, qt)) ? ~0ull : 0ull;
        }
        int qcol[NQ];
#pragma unroll
        for (int t = 0; t < NQ; ++t) qcol[t] = min(max(SVNET_QT(t), 0) * 32 + r, a.Q - 1);
        const int lane_off = 8 * h * (int)a.lda + p;
        constexpr int NPF = NQ <= 5 ? 4 : 1;   // k-steps of A in flight (8 loads each): a whole 64-row block ahead when the registers allow
        float xn[NPF * 8];
        // AFFINE A: the ring holds the raw int16 sums (bit patterns) plus, per k-step, what decides g for this lane's 8 rows: they lie in
        // at most two points (k >= 8, checked on the host) - pooled slot and upstream gradient of both, and the slot of the first row
        int rsa[NPF], rsb[NPF], rt[NPF];
        float rga[NPF], rgb[NPF];
        float a_cs = 0.f, a_al = 0.f, a_be = 0.f;
        const uint8_t* slot_tab = nullptr;
        uint32_t cur_gp = 0, cur_t = 0;         // position of the next k-step to be requested (its row 0): point, slot
        const uint32_t aff_off = 2u * (uint32_t)(8 * h * (int)a.lda + p);   // byte offset of this lane's first row inside a k-step of n16
        const uint32_t npts1 = (uint32_t)(a.npts - 1);
        if (AFF) {
            a_cs = a.chc[p]; a_al = a.chc[a.P + p]; a_be = a.chc[2 * a.P + p];
            slot_tab = (a.chc[4 * a.P + p] != 0.f) ? a.smax : a.smin;
            cur_gp = (uint32_t)(mb / a.kk);
            cur_t = (uint32_t)(mb - (int64_t)cur_gp * a.kk);
        }
#define SVNET_TN_WORDS(M64, SG, NZ)                                                      \
    do {                                                                                 \
        const uint64_t* sg_ = a.b_sign + ((M64) >> 6) * a.Q;                             \
        const uint64_t* nz_ = a.b_nz + ((M64) >> 6) * a.Q;                               \
        _Pragma("unroll") for (int t = 0; t < NQ; ++t) {                                 \
            SG[t] = sg_[qcol[t]];                                                        \
            NZ[t] = nz_[qcol[t]] & slot_live[t];                                         \
        }                                                                                \
    } while (0)
// k-step of 16 rows from uniform row M16: lane (r, h) takes rows M16 + 8h .. + 7 of column p.  Rows past M (last k-step only)
// read the last row; their plane bits are 0 (written so by the producers), so they contribute nothing.
#define SVNET_TN_LOAD_A(S, M16)                                                          \
    do {                                                                                 \
        if (AFF) {   /* requests go out in row order, 16 rows apart: the (point, slot) cursor just advances */ \
            if ((M16) + 16 <= a.M) {   /* whole k-step in range (all but a ragged last one): wave-uniform row base in SGPRs + one 32-bit lane \
                                          offset; the general form below costs nine 64-bit VALU instructions per load */ \
                const int16_t* rb_ = a.n16 + (M16) * a.lda;                              \
                _Pragma("unroll") for (int j = 0; j < 8; ++j)                            \
                    xn[(S) * 8 + j] = __int_as_float((int)ld_i16_sbase(rb_ + j * a.lda, aff_off)); \
            } else {                                                                     \
                _Pragma("unroll") for (int j = 0; j < 8; ++j)                            \
                    xn[(S) * 8 + j] = __int_as_float((int)a.n16[min((M16) + 8 * h + j, mlast) * a.lda + p]); \
            }                                                                            \
            uint32_t tl_ = cur_t + 8u * (uint32_t)h;                                     \
            const uint32_t dq_ = (tl_ * a.kmagic) >> 16;                                 \
            tl_ -= dq_ * (uint32_t)a.kk;                                                 \
            const uint32_t pa_ = min(cur_gp + dq_, npts1), pb_ = min(cur_gp + dq_ + 1u, npts1);   /* (npts * P < 2^31: checked on the host) */ \
            const uint32_t oa_ = pa_ * (uint32_t)a.P + (uint32_t)p, ob_ = pb_ * (uint32_t)a.P + (uint32_t)p; \
            rsa[S] = (int)slot_tab[oa_]; rsb[S] = (int)slot_tab[ob_];                    \
            rga[S] = ld_f32_sbase(a.gy, 4u * oa_); rgb[S] = ld_f32_sbase(a.gy, 4u * ob_); \
            rt[S] = (int)tl_;                                                            \
            cur_t += 16u;                                                                \
            const uint32_t dw_ = (cur_t * a.kmagic) >> 16;                               \
            cur_gp += dw_; cur_t -= dw_ * (uint32_t)a.kk;                                \
        } else                                                                           \
        if ((M16) + 16 <= a.M) {                                                         \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) xn[(S) * 8 + j] = (a.A + ((M16) + j) * a.lda)[lane_off]; \
        } else {                                                                         \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) xn[(S) * 8 + j] = a.A[min((M16) + 8 * h + j, mlast) * a.lda + p]; \
        }                                                                                \
    } while (0)
        constexpr bool PFW = NQ <= 5 && !(AFF && NQ == 5);   // plane words of the next block in flight too (register budget permitting)
        if (PFW) SVNET_TN_WORDS(mb, nsg, nnz);
#pragma unroll
        for (int s = 0; s < NPF; ++s) SVNET_TN_LOAD_A(s, mb + 16 * s);
        for (int64_t m64 = mb; m64 < me; m64 += 64) {
            if (PFW) {
#pragma unroll
                for (int t = 0; t < NQ; ++t) { wsg[t] = nsg[t]; wnz[t] = nnz[t]; }
                if (m64 + 64 < me) SVNET_TN_WORDS(m64 + 64, nsg, nnz);
            } else {
                SVNET_TN_WORDS(m64, wsg, wnz);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int s16 = 16 * s;
                if (m64 + s16 >= me) break;  // wave-uniform
                float x[8];      // (every row range ends on a multiple of 64 or at M, and rows past M have empty planes: no masking)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = xn[(s % NPF) * 8 + j];
                if (AFF) {
                    // row j of this lane is slot rt + j of point A while rt + j < k, slot rt + j - k of point B after that
                    const int ja = rsa[s % NPF] - rt[s % NPF], jb = rsb[s % NPF] + a.kk - rt[s % NPF];
                    const float ga = rga[s % NPF], gb = rgb[s % NPF];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float g = (j == ja) ? ga : ((j == jb) ? gb : 0.f);
                        x[j] = a_cs * g - (a_al + a_be * (float)__float_as_int(x[j]));
                    }
                }
                const Split3 sa = split_frag(x);
                if (m64 + s16 + 16 * NPF < me) SVNET_TN_LOAD_A(s % NPF, m64 + s16 + 16 * NPF);   // NPF k-steps ahead, into the registers just consumed
                const int sh = s16 + 8 * h;
                // No per-tile branches in here: a slot without a live tile has its non-zero plane forced to 0 (SVNET_TN_WORDS),
                // so the whole k-step is one basic block and the LDS lookups of one tile overlap the MFMAs of another.
                constexpr int TG = NQ <= 5 ? NQ : 2;   // tiles expanded together (4 VGPRs each)
#pragma unroll
                for (int t0 = 0; t0 < NQ; t0 += TG) {
                    bf16x8 bfr[TG];
#pragma unroll
                    for (int u = 0; u < TG; ++u) {
                        const int t = t0 + u;
                        const uint32_t nzb = (uint32_t)(wnz[t] >> sh) & 0xFFu;
                        const uint32_t ngb = nzb & ~(uint32_t)(wsg[t] >> sh);
                        const uint4 mg = *reinterpret_cast<const uint4*>(&lut_mag[nzb * 4]);
                        const uint4 ng = *reinterpret_cast<const uint4*>(&lut_neg[ngb * 4]);
                        const uint4 bw = make_uint4(mg.x | ng.x, mg.y | ng.y, mg.z | ng.z, mg.w | ng.w);
                        bfr[u] = __builtin_bit_cast(bf16x8, bw);
                    }
#pragma unroll
                    for (int u = 0; u < TG; ++u) acc[t0 + u] = MFMA(sa.h, bfr[u], acc[t0 + u]);
#pragma unroll
                    for (int u = 0; u < TG; ++u) acc[t0 + u] = MFMA(sa.m, bfr[u], acc[t0 + u]);
#pragma unroll
                    for (int u = 0; u < TG; ++u) acc[t0 + u] = MFMA(sa.l, bfr[u], acc[t0 + u]);
                }
            }
        }
#undef SVNET_TN_WORDS
#undef SVNET_TN_LOAD_A
    }
    // Waves that split the row range of one p tile (narrow P) first combine their partial tiles in LDS: the output
    // matrix is tiny and shared by the whole grid, and same-address float atomics serialise at the memory side.
    if (a.lds_reduce) {   // uniform; every wave of the workgroup is live in this configuration
        extern __shared__ float tnred[];                     // [nsub-1][NQ][16][64]
        if (sub > 0) {
#pragma unroll
            for (int t = 0; t < NQ; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) tnred[(((sub - 1) * ptw + (wave % ptw)) * NQ + t) * 1024 + i * 64 + lane] = acc[t][i];
        }
        __syncthreads();
        if (sub > 0) return;
        for (int s2 = 1; s2 < nsub; ++s2)
#pragma unroll
            for (int t = 0; t < NQ; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] += tnred[(((s2 - 1) * ptw + (wave % ptw)) * NQ + t) * 1024 + i * 64 + lane];
    }
    if (!live) return;
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        const int qt = SVNET_QT(t);
        const int q = qt * 32 + r;      // D col = lane & 31  <-> B operand column (q)
        if (qt >= 0 && q < a.Q && qtile_live(a.qmask, qt)) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int pp = p0 + (i & 3) + 8 * (i >> 2) + 4 * h;  // D row <-> A operand row (p)
#if defined(SVNET_TN_ABLATE) && SVNET_TN_ABLATE == 1       /* diagnostic build (results WRONG): no output atomics */
                if (pp < a.P && acc[t][i] == 123.456f) a.C[(int64_t)pp * a.c_ps + (int64_t)q * a.c_qs] = 1.f;
#else
                if (pp < a.P) atomicAdd(&a.C[(int64_t)pp * a.c_ps + (int64_t)q * a.c_qs], acc[t][i] * a.alpha);
#endif
            }
        }
    }
}

#undef SVNET_QT

// ---- the weight-gradient product of a WIDE fused edge layer (Os = 128, all ten column tiles in use: conv4 of the bench model),
// GX[128 x 320] += dy[rows, 128]^T . x_b[rows, 320], as ONE output tile per workgroup.
// mfma_tn_tern_kernel<5, true> gives every wave one p tile x five q tiles: the dy values (recomputed from n16 and split three ways, ~100
// vector instructions per 16-row step and wave) are prepared twice (two q groups = two workgroups) and the plane bytes are expanded four
// times (four p-tile waves) - ~165 vector instructions per 15 MFMAs, and ablation builds showed its parts ADD (nothing overlaps at two
// waves per SIMD).  Here 8 waves share 32-row slabs through DOUBLE-BUFFERED LDS: thread (channel p, 8-row group g) recomputes and
// splits ITS eight dy values once and leaves three bf16x8 fragments [piece][p][row]; thread q < 320 expands the slab's four plane bytes
// of column q once, [q][row]; wave (wp, wq) multiplies p tile wp with q tiles 5 wq .. 5 wq + 4 (3 + 5 fragment reads for 15 MFMAs per
// 16-row step).  One barrier per slab: the MFMAs of slab s, the preparation of slab s+1 into the other buffer and the requests for slab
// s+3 are independent of each other (a first, single-buffered form with two barriers per slab measured 271 us against the old kernel's
// 265: loads 108 + MFMAs 114 + preparation 137 simply added up).  Measured alone at conv4 (E = 655 360): 222 us against 257 on the same
// box; ablation builds (-DSVNET_AFF2_ABLATE, on the two-set form at 249): no requests 187, no MFMAs 154, no preparation 184 - the parts
// still mostly add: LDS traffic (~216 KB per slab
// and CU: fragments 131, pieces 44, table look-ups 41 + their bank conflicts) is what the phases share.
#ifndef SVNET_AFF2_DEPTH
#define SVNET_AFF2_DEPTH 1                                 /* slabs between a request and its use (2: two register sets - 256 VGPRs, 8 spilled, 231 us against 222) */
#endif
constexpr int A2_LD = 40;                                  // bf16 per LDS row (32 rows + 8: 80 bytes)
constexpr int A2_AP = 3 * 128 * A2_LD, A2_BP = 320 * A2_LD; // elements of one A / B buffer
constexpr size_t A2_LDS_BYTES = 2 * 1024 * 4 + 2 * (size_t)(A2_AP + A2_BP) * 2;
__global__ __launch_bounds__(512, 2) void mfma_tn_aff2_kernel(TnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char A2s[];
    uint32_t* lut_mag = reinterpret_cast<uint32_t*>(A2s);               // [256 * 4] bf16x8 per byte value
    uint32_t* lut_neg = lut_mag + 1024;
    uint16_t* apb = reinterpret_cast<uint16_t*>(A2s + 8192);            // [2][3][128][A2_LD]
    uint16_t* bpb = apb + 2 * A2_AP;                                    // [2][320][A2_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 256) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t lo = (tid >> (2 * w)) & 1u, hi = (tid >> (2 * w + 1)) & 1u;
            lut_mag[tid * 4 + w] = lo * 0x3F80u | hi * 0x3F800000u;
            lut_neg[tid * 4 + w] = lo * 0x8000u | hi * 0x80000000u;
        }
    }
    const int r = lane & 31, h = lane >> 5;
    const int wp = wave & 3, wq = wave >> 2;
    const int64_t mb = (int64_t)blockIdx.x * a.rows_per_block;          // multiple of 64
    const int64_t me = min(a.M, mb + a.rows_per_block);
    if (mb >= me) return;                                               // (uniform)

    // A role: channel p, 8-row group g of the slab (wave-uniform)
    const int p = tid & 127, g = __builtin_amdgcn_readfirstlane(tid >> 7);
    const float a_cs = a.chc[p], a_al = a.chc[128 + p], a_be = a.chc[256 + p];
    const uint8_t* slot_tab = (a.chc[512 + p] != 0.f) ? a.smax : a.smin;
    const uint32_t npts1 = (uint32_t)(a.npts - 1);
    uint32_t cur_gp = (uint32_t)((mb + 8 * g) / a.kk);                  // (point, slot) of the first row of this thread's group in the next slab
    uint32_t cur_t = (uint32_t)((mb + 8 * g) - (int64_t)cur_gp * a.kk);
    // request register sets (R = 0, 1; the second one only with SVNET_AFF2_DEPTH == 2)
    int nraw0[8], nraw1[8];
    int sa_0 = 0, sb_0 = 0, t0_0 = 0, sa_1 = 0, sb_1 = 0, t0_1 = 0;
    float ga_0 = 0.f, gb_0 = 0.f, ga_1 = 0.f, gb_1 = 0.f;
    uint64_t wsg0 = 0, wnz0 = 0, wsg1 = 0, wnz1 = 0;
    const int bq = min(tid, 319);                                       // B role: column (threads past 319 redo column 319: no branch in the loop)
    const int64_t wlast = (a.M - 1) >> 6;
// requests of slab M0 into register set R (rows past M: clamped, the values are zeroed / the plane bits are empty).  Issued in slab order:
// the (point, slot) cursor just advances.
#define SVNET_A2_LOAD(M0, R)                                                                                   \
    do {                                                                                                      \
        /* M % 32 == 0 (checked on the host): a slab lies inside the matrix or entirely past it - then the last slab is re-read and ignored */ \
        const int16_t* rb_ = a.n16 + (min((int64_t)(M0), a.M - 32) + 8 * g) * 128;                            \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) nraw##R[j] = (int)ld_i16_sbase(rb_ + j * 128, 2u * (uint32_t)p); \
        const uint32_t pa_ = min(cur_gp, npts1), pb_ = min(cur_gp + 1u, npts1);                                \
        const uint32_t oa_ = pa_ * 128u + (uint32_t)p, ob_ = pb_ * 128u + (uint32_t)p;                         \
        sa_##R = (int)slot_tab[oa_]; sb_##R = (int)slot_tab[ob_];                                              \
        ga_##R = ld_f32_sbase(a.gy, 4u * oa_); gb_##R = ld_f32_sbase(a.gy, 4u * ob_);                          \
        t0_##R = (int)cur_t;                                                                                  \
        cur_t += 32u;                                                                                         \
        const uint32_t dw_ = (cur_t * a.kmagic) >> 16;                                                        \
        cur_gp += dw_; cur_t -= dw_ * (uint32_t)a.kk;                                                         \
        const int64_t w_ = min((int64_t)((M0) >> 6), wlast);                                                  \
        wsg##R = a.b_sign[w_ * 320 + bq];                                                                     \
        wnz##R = a.b_nz[w_ * 320 + bq];                                                                       \
    } while (0)
// slab M0 from register set R into buffer BUF
#define SVNET_A2_PREP(M0, BUF, R)                                                                              \
    do {                                                                                                      \
        /* this thread's eight dy values: row j of the group is slot t0 + j of point A while t0 + j < k, slot t0 + j - k of point B after */ \
        const int ja_ = sa_##R - t0_##R, jb_ = sb_##R + a.kk - t0_##R;                                         \
        const int lim_ = (int)min((int64_t)8, me - ((M0) + 8 * g));     /* rows of the group inside the range (<= 0: none) */ \
        float x_[8];                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                        \
            const float gsel_ = (j == ja_) ? ga_##R : ((j == jb_) ? gb_##R : 0.f);                             \
            const float v_ = a_cs * gsel_ - (a_al + a_be * (float)nraw##R[j]);                                 \
            x_[j] = j < lim_ ? v_ : 0.f;                                                                       \
        }                                                                                                     \
        const Split3 s3_ = split_frag(x_);                                                                    \
        uint16_t* o_ = apb + (BUF) * A2_AP + p * A2_LD + 8 * g;                                                \
        *reinterpret_cast<bf16x8*>(o_) = s3_.h;                                                               \
        *reinterpret_cast<bf16x8*>(o_ + 128 * A2_LD) = s3_.m;                                                  \
        *reinterpret_cast<bf16x8*>(o_ + 256 * A2_LD) = s3_.l;                                                  \
        /* column bq of x_b: the slab's 32 rows are one half of the 64-row plane words */                     \
        const int sh0_ = (int)((M0) & 32);                                                                    \
        const uint32_t nz32_ = (M0) < me ? (uint32_t)(wnz##R >> sh0_) : 0u, sg32_ = (uint32_t)(wsg##R >> sh0_); \
        _Pragma("unroll") for (int gg = 0; gg < 4; ++gg) {                                                     \
            const uint32_t nzb_ = (nz32_ >> (8 * gg)) & 0xFFu;                                                 \
            const uint32_t ngb_ = nzb_ & ~(sg32_ >> (8 * gg));                                                 \
            const uint4 mg_ = *reinterpret_cast<const uint4*>(&lut_mag[nzb_ * 4]);                             \
            const uint4 ng_ = *reinterpret_cast<const uint4*>(&lut_neg[ngb_ * 4]);                             \
            *reinterpret_cast<uint4*>(bpb + (BUF) * A2_BP + bq * A2_LD + 8 * gg) = make_uint4(mg_.x | ng_.x, mg_.y | ng_.y, mg_.z | ng_.z, mg_.w | ng_.w); \
        }                                                                                                     \
    } while (0)
#if defined(SVNET_AFF2_ABLATE) && SVNET_AFF2_ABLATE == 2                /* diagnostic build (results WRONG): no MFMAs */
#define SVNET_A2_MFMAS(BUF) do { acc[0][0] += (float)apb[(BUF) * A2_AP + tid] + (float)bpb[(BUF) * A2_BP + tid]; } while (0)
#else
#define SVNET_A2_MFMAS(BUF)                                                                                    \
    do {                                                                                                      \
        const uint16_t* ap_ = apb + (BUF) * A2_AP;                                                            \
        const uint16_t* bp_ = bpb + (BUF) * A2_BP;                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 32; ks += 16) {                                                \
            bf16x8 fa[3], fb[5];                                                                              \
            _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                                   \
                fa[pc] = *reinterpret_cast<const bf16x8*>(ap_ + (pc * 128 + 32 * wp + r) * A2_LD + ks + 8 * h); \
            _Pragma("unroll") for (int t = 0; t < 5; ++t)                                                      \
                fb[t] = *reinterpret_cast<const bf16x8*>(bp_ + (32 * (5 * wq + t) + r) * A2_LD + ks + 8 * h);  \
            _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                                   \
                _Pragma("unroll") for (int t = 0; t < 5; ++t) acc[t] = MFMA(fa[pc], fb[t], acc[t]);            \
        }                                                                                                     \
    } while (0)
#endif
// One slab: slab M0 + 32 goes from register set R into the other buffer (its last readers passed the barrier one iteration ago) and the
// requests of slab M0 + 96 go out into the same set; the MFMAs of slab M0 read buffer BUF.  The two are independent: half of the waves
// prepare first, the other half multiply first (first_prep below).
#if defined(SVNET_AFF2_ABLATE) && SVNET_AFF2_ABLATE == 1        /* diagnostic builds (results WRONG): 1 no requests in the loop, 3 no preparation */
#define SVNET_A2_STEP_PL(M0, BUF, R) do { SVNET_A2_PREP((M0) + 32, (BUF) ^ 1, R); } while (0)
#elif defined(SVNET_AFF2_ABLATE) && SVNET_AFF2_ABLATE == 3
#define SVNET_A2_STEP_PL(M0, BUF, R) do { SVNET_A2_LOAD((M0) + 32 * (SVNET_AFF2_DEPTH + 1), R); acc[1][1] += (float)nraw##R[0] + (float)nraw##R[7] + ga_##R + gb_##R + (float)(sa_##R + sb_##R) + (float)(wsg##R ^ wnz##R); } while (0)
#else
#define SVNET_A2_STEP_PL(M0, BUF, R) do { SVNET_A2_PREP((M0) + 32, (BUF) ^ 1, R); SVNET_A2_LOAD((M0) + 32 * (SVNET_AFF2_DEPTH + 1), R); } while (0)
#endif
#define SVNET_A2_STEP(M0, BUF, R)                                                                              \
    do {                                                                                                      \
        if (first_prep) {                                                                                     \
            SVNET_A2_STEP_PL(M0, BUF, R);                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            SVNET_A2_MFMAS(BUF);                                                                              \
        } else {                                                                                              \
            SVNET_A2_MFMAS(BUF);                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            SVNET_A2_STEP_PL(M0, BUF, R);                                                                     \
        }                                                                                                     \
        __syncthreads();                                                                                      \
    } while (0)

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // Which waves prepare first: waves w and w + 4 share a SIMD (tools/hwid/hwid.hip), and giving THEM opposite orders measured slowest
    // (249 us; by wave & 1: 225; by SIMD pair, as here: 216, one box, two-set form) - two waves of a SIMD fill each other's dependency gaps inside the same phase,
    // while the halves of the CU alternate between the LDS-write-heavy preparation and the LDS-read-heavy MFMA operand fetch.
    const bool first_prep = ((wave >> 1) & 1) == 0;

#if SVNET_AFF2_DEPTH == 2
    SVNET_A2_LOAD(mb, 0);
    SVNET_A2_LOAD(mb + 32, 1);
    __syncthreads();                                                    // the tables
    SVNET_A2_PREP(mb, 0, 0);
    SVNET_A2_LOAD(mb + 64, 0);
    __syncthreads();
    // slab mb + 32 waits in set 1, slab mb + 64 in set 0: two slabs per trip, the sets alternate
    for (int64_t m0 = mb; m0 < me; m0 += 64) {
        SVNET_A2_STEP(m0, 0, 1);
        if (m0 + 32 >= me) break;                                       // (uniform)
        SVNET_A2_STEP(m0 + 32, 1, 0);
    }
#else
    SVNET_A2_LOAD(mb, 0);
    __syncthreads();                                                    // the tables
    SVNET_A2_PREP(mb, 0, 0);
    SVNET_A2_LOAD(mb + 32, 0);
    __syncthreads();
    for (int64_t m0 = mb; m0 < me; m0 += 64) {
        SVNET_A2_STEP(m0, 0, 0);
        if (m0 + 32 >= me) break;                                       // (uniform)
        SVNET_A2_STEP(m0 + 32, 1, 0);
    }
#endif
#undef SVNET_A2_STEP
#undef SVNET_A2_STEP_PL
#undef SVNET_A2_MFMAS
#undef SVNET_A2_LOAD
#undef SVNET_A2_PREP
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int q = 32 * (5 * wq + t) + r;                             // D column = lane & 31 <-> x_b column
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int pp = 32 * wp + (i & 3) + 8 * (i >> 2) + 4 * h;     // D row <-> dy channel
            atomicAdd(&a.C[(int64_t)pp * a.c_ps + (int64_t)q * a.c_qs], acc[t][i] * a.alpha);
        }
    }
}

__global__ void zero2d_kernel(float* C, int64_t P, int64_t Q, int64_t ps, int64_t qs) {
    const int64_t total = P * Q;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x)
        C[(o / Q) * ps + (o % Q) * qs] = 0.f;
}

template <int NT, bool AVEC, bool DEEP>
void launch_rows_v(const RowsArgs& a, hipStream_t st) {
    const int64_t row_blocks = svnet_cdiv(a.M, 128);
    const int ny = (int)svnet_cdiv(a.N, NT * 32);
    int64_t gx = row_blocks;
    const int64_t cap = (a.K > KC) ? 4096 : 512 * 4;  // single-chunk B is loaded once per workgroup: keep workgroups persistent
    if (gx > cap) gx = cap;
    const size_t lds = (size_t)NT * 32 * LDS_STRIDE * sizeof(__bf16);
    bool ok = true;                // > 64 KiB of dynamic LDS needs an explicit opt-in (NT = 8: 68 KiB), per device
    SVNET_LDS_OPTIN(ok, lds, "mfma_rows_kernel", reinterpret_cast<const void*>(&mfma_rows_kernel<NT, AVEC, DEEP>));
    (void)ok;                      // (a failed opt-in is in svnet_last_error; the launch below then fails and is reported by the caller's check)
    hipLaunchKernelGGL((mfma_rows_kernel<NT, AVEC, DEEP>), dim3((unsigned)gx, (unsigned)ny), dim3(256), lds, st, a);
}
// the LDS-tiled kernel: many rows, aligned A rows with K % 4 == 0, pre-packed B whose padded column count covers whole 256-column groups
bool rows2_eligible(const RowsArgs& a) {
    static const bool off = getenv("SVNET_ROWS2_OFF") != nullptr;       // (diagnostic switch)
    return !(off || !a.B16 || a.K < 64 || a.M < 4096 || a.N <= 128);
}
bool rows2_aligned(const RowsArgs& a) {
    return a.a_vec && (a.K & 3) == 0 && !(a.a_scale && (reinterpret_cast<uintptr_t>(a.a_scale) & 15) != 0);
}
bool launch_rows2(const RowsArgs& a, hipStream_t st) {
    if (!rows2_eligible(a)) return false;
    const size_t lds = (size_t)2 * R2_BUF_BYTES;
    bool ok = true;
    SVNET_LDS_OPTIN(ok, lds, "mfma_rows2_kernel", reinterpret_cast<const void*>(&mfma_rows2_kernel<1, true>), reinterpret_cast<const void*>(&mfma_rows2_kernel<3, true>),
                    reinterpret_cast<const void*>(&mfma_rows2_kernel<1, false>), reinterpret_cast<const void*>(&mfma_rows2_kernel<3, false>));
    (void)ok;
    const dim3 grid((unsigned)svnet_cdiv(a.M, R2_BM), (unsigned)svnet_cdiv(a.N, R2_BN));
    const bool av = rows2_aligned(a);
    if (a.b_piece) { if (av) hipLaunchKernelGGL((mfma_rows2_kernel<3, true>), grid, dim3(256), lds, st, a); else hipLaunchKernelGGL((mfma_rows2_kernel<3, false>), grid, dim3(256), lds, st, a); }
    else { if (av) hipLaunchKernelGGL((mfma_rows2_kernel<1, true>), grid, dim3(256), lds, st, a); else hipLaunchKernelGGL((mfma_rows2_kernel<1, false>), grid, dim3(256), lds, st, a); }
    return true;
}
template <int NJ>
void launch_rows3_nj(const RowsArgs& a, hipStream_t st) {
    const size_t lds = (size_t)R3_A_BYTES + 2 * (size_t)(64 * NJ) * R2_LDB * 2;
    bool ok = true;
    SVNET_LDS_OPTIN(ok, lds, "mfma_rows3_kernel", reinterpret_cast<const void*>(&mfma_rows3_kernel<NJ, 4>), reinterpret_cast<const void*>(&mfma_rows3_kernel<NJ, 2>),
                    reinterpret_cast<const void*>(&mfma_rows3_kernel<NJ, 1>));
    (void)ok;
    const dim3 grid((unsigned)svnet_cdiv(a.M, R2_BM), (unsigned)svnet_cdiv(a.N, 64 * NJ));
    const uintptr_t ap = reinterpret_cast<uintptr_t>(a.A);
    if ((ap & 15) == 0 && (a.lda & 3) == 0 && (a.K & 3) == 0) hipLaunchKernelGGL((mfma_rows3_kernel<NJ, 4>), grid, dim3(256), lds, st, a);
    else if ((ap & 7) == 0 && (a.lda & 1) == 0 && (a.K & 1) == 0) hipLaunchKernelGGL((mfma_rows3_kernel<NJ, 2>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((mfma_rows3_kernel<NJ, 1>), grid, dim3(256), lds, st, a);
}
// general fp32 B (three packed pieces, a.b_piece apart), any K >= 1 / N >= 1, plain / alpha / bias / accumulate epilogue
bool launch_rows3(const RowsArgs& a, hipStream_t st) {
    static const bool off = getenv("SVNET_ROWS3_OFF") != nullptr;       // (diagnostic switch: the older one- / three-launch forms)
    if (off || !a.B16 || !a.b_piece || a.col_scale || a.mask || a.col_sum || a.a_scale || a.M < 1024) return false;
    if (a.N <= 64) launch_rows3_nj<1>(a, st);
    else if (a.N <= 128) launch_rows3_nj<2>(a, st);
    else launch_rows3_nj<4>(a, st);
    return true;
}
template <int NT>
void launch_rows(const RowsArgs& a, hipStream_t st) {
    if (NT == 8 && launch_rows2(a, st)) return;
    const bool vec = a.a_vec && a.K >= 8 && (a.K & 7) == 0;
    if (a.K > 64) { if (vec) launch_rows_v<NT, true, true>(a, st); else launch_rows_v<NT, false, true>(a, st); }
    else { if (vec) launch_rows_v<NT, true, false>(a, st); else launch_rows_v<NT, false, false>(a, st); }
}

template <int IP, int JQ>
void launch_tn2(TnArgs a, hipStream_t st) {
    const int64_t tiles = svnet_cdiv(a.P, 64 * IP) * svnet_cdiv(a.Q, 64 * JQ);
    const int64_t want = svnet_cdiv(512, tiles);              // one resident round (2 workgroups per CU)
    int64_t rpb = svnet_cdiv(svnet_cdiv(a.M, want), T2_MK) * T2_MK;
    if (rpb < 8 * T2_MK) rpb = 8 * T2_MK;
    a.rows_per_block = rpb;
    hipLaunchKernelGGL((mfma_tn2_kernel<IP, JQ>), dim3((unsigned)svnet_cdiv(a.M, rpb), (unsigned)svnet_cdiv(a.P, 64 * IP), (unsigned)svnet_cdiv(a.Q, 64 * JQ)),
                       dim3(256), 0, st, a);
}
template <int NQ, int BMODE>
void launch_tn(TnArgs a, hipStream_t st) {
    // (a partial tile mask names tiles 0..31 only: svnet_mfma_tn refuses one for Q > 1024)
    const int ptiles = (int)svnet_cdiv(a.P, 32);
    a.ptiles_per_block = ptiles >= 3 ? 4 : ptiles;
    int gz = (int)svnet_cdiv(a.Q, NQ * 32);
    a.qlist = 0;
    if (BMODE == 1 && a.qmask != 0xFFFFFFFFu) {   // few tiles in use: one z group over exactly those (the A split is then shared)
        int used = 0;
        uint32_t list = 0;
        for (int t = 0; t < 15 && t * 32 < a.Q; ++t)
            if ((a.qmask >> t) & 1u) { if (used < 8) list |= (uint32_t)(t + 1) << (4 * used); ++used; }
        if (used > 0 && used <= NQ) { a.qlist = list; gz = 1; }
    }
    const int gy = (int)svnet_cdiv(ptiles, a.ptiles_per_block);
    // ~4 workgroups per CU in total; 2 when the output is large (every row split ends in P*Q float atomics, which the memory
    // side executes at ~1 TB/s: conv5's 512 x 505 gradient spent half its time there with 128 splits)
    int64_t target = (BMODE == 1 || (int64_t)a.P * a.Q >= 128 * 1024) ? 512 : 1024;   // ternary: exactly one resident round (2 per CU)
    if (const char* e = getenv("SVNET_TN_TARGET")) target = atoi(e);
    int64_t want = svnet_cdiv(target, (int64_t)gy * gz);
    int64_t rpb = svnet_cdiv(svnet_cdiv(a.M, want), 256) * 256;
    if (rpb < 256) rpb = 256;
    a.rows_per_block = rpb;
    const int64_t gx = svnet_cdiv(a.M, rpb);
    const int nsub = 4 / a.ptiles_per_block;
    const size_t lds = (size_t)(nsub - 1) * a.ptiles_per_block * NQ * 1024 * sizeof(float);
    a.lds_reduce = (nsub > 1 && ptiles == a.ptiles_per_block && lds <= 64 * 1024) ? 1 : 0;
    if (BMODE == 1) {
        bool ok = true;                 // 8 KiB of static tables + up to 64 KiB of dynamic LDS: above the 64 KiB default
        SVNET_LDS_OPTIN(ok, 64 * 1024, "mfma_tn_tern_kernel", reinterpret_cast<const void*>(&mfma_tn_tern_kernel<NQ, false>),
                        reinterpret_cast<const void*>(&mfma_tn_tern_kernel<NQ, true>));
        (void)ok;
    }
    if (BMODE == 1 && a.n16)
        hipLaunchKernelGGL((mfma_tn_tern_kernel<NQ, true>), dim3((unsigned)gx, (unsigned)gy, (unsigned)gz), dim3(256), a.lds_reduce ? lds : 0, st, a);
    else if (BMODE == 1)
        hipLaunchKernelGGL((mfma_tn_tern_kernel<NQ, false>), dim3((unsigned)gx, (unsigned)gy, (unsigned)gz), dim3(256), a.lds_reduce ? lds : 0, st, a);
    else
        hipLaunchKernelGGL((mfma_tn_kernel<NQ, 0>), dim3((unsigned)gx, (unsigned)gy, (unsigned)gz), dim3(256), a.lds_reduce ? lds : 0, st, a);
}

}  // namespace

extern "C" size_t svnet_gemm_workspace_bytes(int64_t N, int64_t K) {
    if (N <= 0 || K <= 0) return 0;
    const int64_t tile = N <= 32 ? 32 : (N <= 64 ? 64 : (N <= 128 ? 128 : 256));     // widest column tile a workgroup may use
    return (size_t)((N + tile - 1) / tile * tile) * (size_t)((K + 15) / 16 * 16) * 2;
}

// Internal entry points used by svnet_gemm_f32 (gemm.hip).
int svnet_mfma_rows(const svnet_gemm_desc& d, hipStream_t st) {
    SVNET_REQUIRE(d.a_rs < ((int64_t)1 << 24), SVNET_E_UNSUPPORTED, "svnet_mfma_rows: A row stride %lld >= 2^24 (32-bit tile offsets)", (long long)d.a_rs);
    RowsArgs a;
    a.A = d.A; a.lda = d.a_rs; a.a_scale = d.a_scale;
    a.B = d.B; a.b_rs = d.b_rs; a.b_cs = d.b_cs;
    a.C = d.C; a.ldc = d.ldc;
    a.M = d.M; a.N = (int)d.N; a.K = (int)d.K;
    a.alpha = d.alpha; a.col_scale = d.col_scale; a.bias = d.bias;
    a.mask = d.mask; a.col_sum = d.col_sum; a.accumulate = d.accumulate;
    a.a_vec = (d.a_rs % 4 == 0) && (reinterpret_cast<uintptr_t>(d.A) % 16 == 0);
    a.B16 = nullptr; a.Kp = 0; a.b_piece = 0;
    const int cpb = (d.N <= 32 || d.M <= 512) ? 32 : (d.N <= 64 ? 64 : (d.N <= 128 ? 128 : 256));   // columns per workgroup (NT * 32)
    if (d.workspace && d.workspace_bytes >= svnet_gemm_workspace_bytes(d.N, d.K) && reinterpret_cast<uintptr_t>(d.workspace) % 16 == 0) {
        const int Kp = (int)((d.K + 15) / 16 * 16), Np = (int)((d.N + cpb - 1) / cpb * cpb);
        uint16_t* w = reinterpret_cast<uint16_t*>(d.workspace);
        hipLaunchKernelGGL(pack_b_bf16_kernel, dim3(svnet_grid((int64_t)Kp * Np, 256)), dim3(256), 0, st, d.B, d.b_rs, d.b_cs, (int)d.K,
                           (int)d.N, Kp, Np, w);
        SVNET_CHECK_LAUNCH("pack_b_bf16_kernel");
        a.B16 = w; a.Kp = Kp;
    }
    // few row blocks (the classifier head: M = batch): narrow column tiles so that the grid still has tens of workgroups
    if (d.N <= 32 || d.M <= 512) launch_rows<1>(a, st);
    else if (d.N <= 64) launch_rows<2>(a, st);
    else if (d.N <= 128) launch_rows<4>(a, st);
    else launch_rows<8>(a, st);
    SVNET_CHECK_LAUNCH("mfma_rows_kernel");
    return SVNET_OK;
}

// rows x GENERAL fp32 weights (the fp layers of the PointNet-style callers: conv_fuse 2044 -> 512 on 32 768 rows ran on the vector-ALU
// tile GEMM at 18 % of the f32 rate): B is split exactly into three bf16 pieces B = Bh + Bm + Bl, packed once (one launch), and
// mfma_rows3_kernel multiplies them with the three pieces of A - the six leading bf16 x bf16 products per fp32 product (the dropped
// ones are below 2^-24 of the product), fp32 accumulation: fp32-GEMM accuracy (tests: 2e-6 of the largest output against float64).
// On the bf16 matrix cores six passes cost 6/16 of ONE f32-input MFMA pass.  (SVNET_ROWS3_OFF: the older nine-product forms below.)
// Epilogue terms that are not linear in B (column scale, mask, column sums, per-k scale) are not taken here (checked by the caller).
int svnet_mfma_rows_split(const svnet_gemm_desc& d, hipStream_t st) {
    const size_t one = svnet_gemm_workspace_bytes(d.N, d.K);
    SVNET_REQUIRE(d.workspace && d.workspace_bytes >= 3 * one && reinterpret_cast<uintptr_t>(d.workspace) % 16 == 0 && one % 16 == 0, SVNET_E_WORKSPACE,
                  "svnet_mfma_rows_split: needs 3 x svnet_gemm_workspace_bytes of 16-byte aligned workspace");
    SVNET_REQUIRE(!d.col_scale && !d.mask && !d.col_sum && !d.a_scale, SVNET_E_UNSUPPORTED, "svnet_mfma_rows_split: plain or bias epilogue only");
    SVNET_REQUIRE(d.a_rs < ((int64_t)1 << 24), SVNET_E_UNSUPPORTED, "svnet_mfma_rows_split: A row stride %lld >= 2^24", (long long)d.a_rs);
    RowsArgs a;
    a.A = d.A; a.lda = d.a_rs; a.a_scale = nullptr;
    a.B = d.B; a.b_rs = d.b_rs; a.b_cs = d.b_cs;
    a.C = d.C; a.ldc = d.ldc;
    a.M = d.M; a.N = (int)d.N; a.K = (int)d.K;
    a.alpha = d.alpha; a.col_scale = nullptr; a.bias = d.bias;
    a.mask = nullptr; a.col_sum = nullptr;
    a.a_vec = (d.a_rs % 4 == 0) && (reinterpret_cast<uintptr_t>(d.A) % 16 == 0);
    const int cpb = (d.N <= 32 || d.M <= 512) ? 32 : (d.N <= 64 ? 64 : (d.N <= 128 ? 128 : 256));
    const int Kp = (int)((d.K + 15) / 16 * 16), Np = (int)((d.N + cpb - 1) / cpb * cpb);
    a.Kp = Kp;
    hipLaunchKernelGGL(pack_b_bf16_kernel, dim3(svnet_grid((int64_t)Kp * Np, 256)), dim3(256), 0, st, d.B, d.b_rs, d.b_cs, (int)d.K, (int)d.N, Kp, Np,
                       reinterpret_cast<uint16_t*>(d.workspace), 3, (int64_t)(one / 2));
    SVNET_CHECK_LAUNCH("pack_b_bf16_kernel");
    a.B16 = reinterpret_cast<uint16_t*>(d.workspace);
    a.accumulate = d.accumulate;
    a.b_piece = (int64_t)(one / 2);
    if (launch_rows3(a, st)) {                                            // A split once per element, one launch
        SVNET_CHECK_LAUNCH("mfma_rows3_kernel");
        return SVNET_OK;
    }
    if (launch_rows2(a, st)) {                                            // one launch: A read once for the three pieces, C written once
        SVNET_CHECK_LAUNCH("mfma_rows2_kernel (split B)");
        return SVNET_OK;
    }
    a.b_piece = 0;
    for (int piece = 0; piece < 3; ++piece) {                             // other shapes: the exact-B kernel once per piece, accumulating
        a.B16 = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(d.workspace) + piece * one);
        a.accumulate = (piece > 0 || d.accumulate) ? 1 : 0;
        if (piece > 0) a.bias = nullptr;                                  // (the bias goes in once)
        if (d.N <= 32 || d.M <= 512) launch_rows<1>(a, st);
        else if (d.N <= 64) launch_rows<2>(a, st);
        else if (d.N <= 128) launch_rows<4>(a, st);
        else launch_rows<8>(a, st);
        SVNET_CHECK_LAUNCH("mfma_rows_kernel (split B)");
    }
    return SVNET_OK;
}

// out(p,q) = alpha * sum_m A[m*lda+p] * B(m,q); B fp32 rows (ldb) or row-sliced ternary planes.
int svnet_mfma_tn(const float* A, int64_t lda, const float* B, int64_t ldb, const uint64_t* b_sign, const uint64_t* b_nz,
                  int64_t M, int64_t P, int64_t Q, float* C, int64_t c_ps, int64_t c_qs, float alpha, int accumulate,
                  hipStream_t st, uint32_t q_tile_mask) {
    SVNET_REQUIRE(q_tile_mask == 0 || q_tile_mask == 0xFFFFFFFFu || Q <= 1024, SVNET_E_UNSUPPORTED,
                  "svnet_mfma_tn: a partial column-tile mask needs Q <= 1024 (got %lld)", (long long)Q);
    if (!accumulate) {
        hipLaunchKernelGGL(zero2d_kernel, dim3(svnet_grid(P * Q, 256)), dim3(256), 0, st, C, P, Q, c_ps, c_qs);
        SVNET_CHECK_LAUNCH("zero2d_kernel");
    }
    if (M == 0) return SVNET_OK;
    TnArgs a;
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.b_sign = b_sign; a.b_nz = b_nz;
    a.C = C; a.c_ps = c_ps; a.c_qs = c_qs; a.M = M; a.P = (int)P; a.Q = (int)Q; a.alpha = alpha; a.rows_per_block = 0; a.lds_reduce = 0;
    a.qmask = q_tile_mask ? q_tile_mask : 0xFFFFFFFFu;
    a.n16 = nullptr; a.gy = nullptr; a.smax = a.smin = nullptr; a.chc = nullptr; a.kk = 1; a.kmagic = 0; a.npts = 0;
    const bool tern = b_sign != nullptr;
    static const bool tn2_off = getenv("SVNET_TN2_OFF") != nullptr;     // (diagnostic switch: the per-wave-split kernel)
    if (!tern && !tn2_off) {                                            // fp32 x fp32: the LDS-tiled kernel
        // the largest tile that still leaves >= 32 tiles: every row split adds one float atomic per output element, and with few tiles
        // the grid is filled by row splits ([512 x 127] as four 128 x 128 tiles: 128 splits, 50 of its 82 us in the atomics)
        auto tiles = [&](int ip, int jq) { return svnet_cdiv(P, 64 * ip) * svnet_cdiv(Q, 64 * jq); };
        if (P > 64 && Q > 64 && tiles(2, 2) >= 32) launch_tn2<2, 2>(a, st);
        else if (P > 64 && (Q <= 64 || P >= Q) && tiles(2, 1) >= 32) launch_tn2<2, 1>(a, st);
        else if (Q > 64 && tiles(1, 2) >= 32) launch_tn2<1, 2>(a, st);
        else if (P > 64 && tiles(2, 1) >= 32) launch_tn2<2, 1>(a, st);
        else launch_tn2<1, 1>(a, st);
        SVNET_CHECK_LAUNCH("mfma_tn2_kernel");
        return SVNET_OK;
    }
    if (Q <= 32) { if (tern) launch_tn<1, 1>(a, st); else launch_tn<1, 0>(a, st); }
    else if (Q <= 64) { if (tern) launch_tn<2, 1>(a, st); else launch_tn<2, 0>(a, st); }
    else if (Q <= 128) { if (tern) launch_tn<4, 1>(a, st); else launch_tn<4, 0>(a, st); }
    else if (Q <= 160 || Q == 320) { if (tern) launch_tn<5, 1>(a, st); else launch_tn<5, 0>(a, st); }   // 320 = fused edge block
    else { if (tern) launch_tn<5, 1>(a, st); else launch_tn<4, 0>(a, st); }   // ternary: 160-column groups too - the 5-tile kernel keeps a whole block of A and the next plane words in flight (8 tiles: 255 VGPRs, neither), worth more than the extra L2 reads of A (conv5: -60 us per step)   // fp32 B: 128-column groups (register budget of the prefetch)
    SVNET_CHECK_LAUNCH("mfma_tn_kernel");
    return SVNET_OK;
}

// Weight-gradient product of a FUSED edge layer:  GX[p, q] += sum over the E edge rows of dy[e, p] * x_b[e, q]  (p < Os output
// channels, q < 320 fused feature columns), with dy = dL/dy_pre recomputed inside the GEMM from the int16 sums the forward kept
// (TnArgs "AFFINE A") instead of read from an fp32 [E, Os] tensor that the tile kernel would have to write first.
extern "C" int svnet_edgeblock_wgrad_f32(const int16_t* n16, const uint8_t* slot_max, const uint8_t* slot_min, const float* gy,
                                         const float* chc, const uint64_t* x_sign, const uint64_t* x_nz, int64_t E, int64_t k, int64_t Os,
                                         float* GX, uint32_t q_tile_mask, void* stream) {
    SVNET_REQUIRE(n16 && slot_max && slot_min && gy && chc && x_sign && x_nz && GX, SVNET_E_ARG, "svnet_edgeblock_wgrad_f32: null pointer");
    SVNET_REQUIRE(E > 0 && k >= 8 && k <= 64 && E % k == 0 && Os > 0 && Os <= 128, SVNET_E_UNSUPPORTED,
                  "svnet_edgeblock_wgrad_f32: needs 8 <= k <= 64, Os <= 128 (got k=%lld, Os=%lld)", (long long)k, (long long)Os);
    SVNET_REQUIRE((E / k) * Os < ((int64_t)1 << 29), SVNET_E_UNSUPPORTED, "svnet_edgeblock_wgrad_f32: more than 2^29 point-channels (32-bit offsets)");
    hipStream_t st = (hipStream_t)stream;
    TnArgs a;
    a.A = nullptr; a.lda = Os; a.B = nullptr; a.ldb = 0; a.b_sign = x_sign; a.b_nz = x_nz;
    a.C = GX; a.c_ps = 320; a.c_qs = 1; a.M = E; a.P = (int)Os; a.Q = 320; a.alpha = 1.f; a.rows_per_block = 0; a.lds_reduce = 0;
    a.qmask = q_tile_mask ? q_tile_mask : 0xFFFFFFFFu;
    a.n16 = n16; a.gy = gy; a.smax = slot_max; a.smin = slot_min; a.chc = chc;
    a.kk = (int)k; a.kmagic = (uint32_t)((65536 + k - 1) / k); a.npts = E / k;
    static const bool aff2_off = getenv("SVNET_AFF2_OFF") != nullptr;   // (diagnostic switch: the one-p-tile-per-wave kernel for every width)
    if (Os == 128 && (a.qmask & 0x3FFu) == 0x3FFu && (E & 31) == 0 && !aff2_off) {       // wide layer, all ten column tiles in use: one output tile per workgroup
        // (one 8-wave workgroup per CU: 126 KB of LDS.  256 workgroups - one round, every CU held for the whole launch - measured 4.356 /
        //  4.370 / 4.366 ms per step against 4.339 / 4.338 / 4.349 with 384: the shorter workgroups hand their CUs back to the gather on the main
        //  stream half-way; 320: 4.351, 448: 4.361, 512: 4.406, 128: 4.417; profiles/r04_ab_aff2_target.log)
        int64_t target = 384;
        if (const char* e = getenv("SVNET_AFF2_TARGET")) target = atoi(e);
        int64_t rpb = svnet_cdiv(svnet_cdiv(E, target), 64) * 64;
        if (rpb < 256) rpb = 256;
        a.rows_per_block = rpb;
        bool ok = true;
        SVNET_LDS_OPTIN(ok, A2_LDS_BYTES, "mfma_tn_aff2_kernel", reinterpret_cast<const void*>(&mfma_tn_aff2_kernel));
        if (!ok) return SVNET_E_LAUNCH;
        hipLaunchKernelGGL(mfma_tn_aff2_kernel, dim3((unsigned)svnet_cdiv(E, rpb)), dim3(512), A2_LDS_BYTES, st, a);
        SVNET_CHECK_LAUNCH("mfma_tn_aff2_kernel");
        return SVNET_OK;
    }
    launch_tn<5, 1>(a, st);
    SVNET_CHECK_LAUNCH("mfma_tn_tern_kernel (affine)");
    return SVNET_OK;
}
