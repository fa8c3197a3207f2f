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
};

constexpr int KC = 128;         // K chunk staged in LDS
constexpr int LDS_STRIDE = KC + 8;  // bf16 elements per LDS row: (KC/8 + 1) 16-byte slots, odd -> conflict-free b128 reads

// NT = number of 32-column tiles handled by a workgroup (N_tile = 32*NT <= 256); 4 waves x 32 rows per iteration.
template <int NT>
__global__ __launch_bounds__(256) void mfma_rows_kernel(RowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) __bf16 Bt[];  // [NT*32][LDS_STRIDE]
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
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

        for (int ch = 0; ch < nchunks; ++ch) {
            const int k0 = ch * KC;
            const int kc = min(KC, a.K - k0);
            const int kc16 = (kc + 15) & ~15;
            if (nchunks > 1 || !b_loaded) {
                __syncthreads();  // previous readers of Bt are done
                const int total = NT * 32 * kc16;
                const bool k_fast = (a.b_rs == 1);
                for (int e = tid; e < total; e += 256) {
                    int n, k;
                    if (k_fast) { k = e % kc16; n = e / kc16; } else { n = e % (NT * 32); k = e / (NT * 32); }
                    float v = 0.f;
                    if (n0 + n < a.N && k < kc) v = a.B[(int64_t)(k0 + k) * a.b_rs + (int64_t)(n0 + n) * a.b_cs];
                    Bt[n * LDS_STRIDE + k] = bf16_from_bits(__float_as_uint(v) >> 16);
                }
                __syncthreads();
                b_loaded = true;
            }
            for (int ks = 0; ks < kc16; ks += 16) {
                float x[8];
                const int kk = k0 + ks + 8 * h;
                if (a.a_vec && row_ok && kk + 8 <= a.K) {
                    const float4 v0 = *reinterpret_cast<const float4*>(a.A + arow * a.lda + kk);
                    const float4 v1 = *reinterpret_cast<const float4*>(a.A + arow * a.lda + kk + 4);
                    x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w;
                    x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = (row_ok && kk + j < a.K) ? a.A[arow * a.lda + kk + j] : 0.f;
                }
                if (a.a_scale) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] *= (kk + j < a.K) ? a.a_scale[kk + j] : 0.f;
                }
                const Split3 s = split_frag(x);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Bt[(t * 32 + r) * LDS_STRIDE + ks + 8 * h]);
                    acc[t] = MFMA(s.h, b, acc[t]);
                    acc[t] = MFMA(s.m, b, acc[t]);
                    acc[t] = MFMA(s.l, b, acc[t]);
                }
            }
        }
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
                        *dst = a.accumulate ? (*dst + v) : v;
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
            if (h == 0 && col < a.N) atomicAdd(&a.col_sum[col], s);
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
};

// NQ 32-wide q tiles per workgroup (blockIdx.z picks the group).  The 4 waves cover `ptw` p tiles (1, 2 or 4 per
// workgroup); when P is narrow (ptw < 4) the spare waves split the workgroup's row range instead of idling.
template <int NQ, int BMODE>
__global__ __launch_bounds__(256) void mfma_tn_kernel(TnArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int ptw = a.ptiles_per_block;                      // 1, 2 or 4
    const int p0 = (blockIdx.y * ptw + (wave % ptw)) * 32;
    const int q0 = blockIdx.z * (NQ * 32);
    if (p0 >= a.P) return;  // wave-uniform; no barriers in this kernel
    const int nsub = 4 / ptw, sub = wave / ptw;
    const int64_t rows_sub = ((a.rows_per_block / nsub + 63) >> 6) << 6;   // multiple of 64
    const int64_t mb = (int64_t)blockIdx.x * a.rows_per_block + (int64_t)sub * rows_sub;
    const int64_t me = min(min(a.M, (int64_t)(blockIdx.x + 1) * a.rows_per_block), mb + rows_sub);
    const int p = p0 + r;
    const bool p_ok = p < a.P;

    f32x16 acc[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    for (int64_t m64 = mb; m64 < me; m64 += 64) {
        uint64_t wsg[NQ], wnz[NQ];
        if (BMODE == 1) {
#pragma unroll
            for (int t = 0; t < NQ; ++t) {
                const int q = q0 + t * 32 + r;
                const bool ok = q < a.Q;
                wsg[t] = ok ? a.b_sign[(m64 >> 6) * a.Q + q] : 0ull;
                wnz[t] = ok ? a.b_nz[(m64 >> 6) * a.Q + q] : 0ull;
            }
        }
#pragma unroll
        for (int s16 = 0; s16 < 64; s16 += 16) {
            const int64_t mrow = m64 + s16 + 8 * h;
            if (m64 + s16 >= me) break;  // wave-uniform
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (p_ok && mrow + j < me) ? a.A[(mrow + j) * a.lda + p] : 0.f;
            const Split3 sa = split_frag(x);
#pragma unroll
            for (int t = 0; t < NQ; ++t) {
                if (q0 + t * 32 >= a.Q) break;  // uniform
                if (BMODE == 1) {
                    bf16x8 b;
                    const int sh = s16 + 8 * h;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t nzb = (uint32_t)(wnz[t] >> (sh + j)) & 1u;
                        const uint32_t sgb = (uint32_t)(wsg[t] >> (sh + j)) & 1u;
                        b[j] = bf16_from_bits(nzb ? (sgb ? 0x3F80u : 0xBF80u) : 0u);  // rows beyond M carry nz = 0
                    }
                    acc[t] = MFMA(sa.h, b, acc[t]);
                    acc[t] = MFMA(sa.m, b, acc[t]);
                    acc[t] = MFMA(sa.l, b, acc[t]);
                } else {
                    const int q = q0 + t * 32 + r;
                    float y[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) y[j] = (q < a.Q && mrow + j < me) ? a.B[(mrow + j) * a.ldb + q] : 0.f;
                    const Split3 sb = split_frag(y);
                    acc[t] = MFMA(sa.h, sb.h, acc[t]);
                    acc[t] = MFMA(sa.h, sb.m, acc[t]);
                    acc[t] = MFMA(sa.m, sb.h, acc[t]);
                    acc[t] = MFMA(sa.h, sb.l, acc[t]);
                    acc[t] = MFMA(sa.l, sb.h, acc[t]);
                    acc[t] = MFMA(sa.m, sb.m, acc[t]);
                }
            }
        }
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

__global__ void zero2d_kernel(float* C, int64_t P, int64_t Q, int64_t ps, int64_t qs) {
    const int64_t total = P * Q;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x)
        C[(o / Q) * ps + (o % Q) * qs] = 0.f;
}

template <int NT>
void launch_rows(const RowsArgs& a, hipStream_t st) {
    const int64_t row_blocks = svnet_cdiv(a.M, 128);
    const int ny = (int)svnet_cdiv(a.N, NT * 32);
    int64_t gx = row_blocks;
    const int64_t cap = (a.K > KC) ? 4096 : 512 * 4;  // single-chunk B is loaded once per workgroup: keep workgroups persistent
    if (gx > cap) gx = cap;
    const size_t lds = (size_t)NT * 32 * LDS_STRIDE * sizeof(__bf16);
    static bool attr_set = false;  // > 64 KiB of dynamic LDS needs an explicit opt-in (NT = 8: 68 KiB)
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_rows_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((mfma_rows_kernel<NT>), dim3((unsigned)gx, (unsigned)ny), dim3(256), lds, st, a);
}

template <int NQ, int BMODE>
void launch_tn(TnArgs a, hipStream_t st) {
    const int ptiles = (int)svnet_cdiv(a.P, 32);
    a.ptiles_per_block = ptiles >= 3 ? 4 : ptiles;
    const int gy = (int)svnet_cdiv(ptiles, a.ptiles_per_block), gz = (int)svnet_cdiv(a.Q, NQ * 32);
    int64_t want = svnet_cdiv(1024, (int64_t)gy * gz);               // ~4 workgroups per CU in total
    int64_t rpb = svnet_cdiv(svnet_cdiv(a.M, want), 256) * 256;
    if (rpb < 256) rpb = 256;
    a.rows_per_block = rpb;
    const int64_t gx = svnet_cdiv(a.M, rpb);
    const int nsub = 4 / a.ptiles_per_block;
    const size_t lds = (size_t)(nsub - 1) * a.ptiles_per_block * NQ * 1024 * sizeof(float);
    a.lds_reduce = (nsub > 1 && ptiles == a.ptiles_per_block && lds <= 64 * 1024) ? 1 : 0;
    hipLaunchKernelGGL((mfma_tn_kernel<NQ, BMODE>), dim3((unsigned)gx, (unsigned)gy, (unsigned)gz), dim3(256), a.lds_reduce ? lds : 0, st, a);
}

}  // namespace

// Internal entry points used by svnet_gemm_f32 (gemm.hip).
int svnet_mfma_rows(const svnet_gemm_desc& d, hipStream_t st) {
    RowsArgs a;
    a.A = d.A; a.lda = d.a_rs; a.a_scale = d.a_scale;
    a.B = d.B; a.b_rs = d.b_rs; a.b_cs = d.b_cs;
    a.C = d.C; a.ldc = d.ldc;
    a.M = d.M; a.N = (int)d.N; a.K = (int)d.K;
    a.alpha = d.alpha; a.col_scale = d.col_scale; a.bias = d.bias;
    a.mask = d.mask; a.col_sum = d.col_sum; a.accumulate = d.accumulate;
    a.a_vec = (d.a_rs % 4 == 0) && (reinterpret_cast<uintptr_t>(d.A) % 16 == 0);
    if (d.N <= 32) launch_rows<1>(a, st);
    else if (d.N <= 64) launch_rows<2>(a, st);
    else if (d.N <= 128) launch_rows<4>(a, st);
    else launch_rows<8>(a, st);
    SVNET_CHECK_LAUNCH("mfma_rows_kernel");
    return SVNET_OK;
}

// out(p,q) = alpha * sum_m A[m*lda+p] * B(m,q); B fp32 rows (ldb) or row-sliced ternary planes.
int svnet_mfma_tn(const float* A, int64_t lda, const float* B, int64_t ldb, const uint64_t* b_sign, const uint64_t* b_nz,
                  int64_t M, int64_t P, int64_t Q, float* C, int64_t c_ps, int64_t c_qs, float alpha, int accumulate,
                  hipStream_t st) {
    if (!accumulate) {
        hipLaunchKernelGGL(zero2d_kernel, dim3(svnet_grid(P * Q, 256)), dim3(256), 0, st, C, P, Q, c_ps, c_qs);
        SVNET_CHECK_LAUNCH("zero2d_kernel");
    }
    if (M == 0) return SVNET_OK;
    TnArgs a;
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.b_sign = b_sign; a.b_nz = b_nz;
    a.C = C; a.c_ps = c_ps; a.c_qs = c_qs; a.M = M; a.P = (int)P; a.Q = (int)Q; a.alpha = alpha; a.rows_per_block = 0; a.lds_reduce = 0;
    const bool tern = b_sign != nullptr;
    if (Q <= 32) { if (tern) launch_tn<1, 1>(a, st); else launch_tn<1, 0>(a, st); }
    else if (Q <= 64) { if (tern) launch_tn<2, 1>(a, st); else launch_tn<2, 0>(a, st); }
    else if (Q <= 128) { if (tern) launch_tn<4, 1>(a, st); else launch_tn<4, 0>(a, st); }
    else if (Q <= 160 || Q == 320) { if (tern) launch_tn<5, 1>(a, st); else launch_tn<5, 0>(a, st); }   // 320 = fused edge block
    else { if (tern) launch_tn<8, 1>(a, st); else launch_tn<8, 0>(a, st); }
    SVNET_CHECK_LAUNCH("mfma_tn_kernel");
    return SVNET_OK;
}
