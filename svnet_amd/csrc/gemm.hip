// svnet_gemm_f32: dispatcher + the generic fp32 vector-ALU GEMM.
//
// Serves every dense contraction on the path: F.linear of sv_layers.py:31,49 (fp layers, the sign-weight
// vector linear `linear2`, the gate, the classifier) and the autograd products of their backward.
// Dispatch (all shapes are tall-and-skinny, i.e. HBM-bound):
//   * B exact in bf16 (sign weights) and A row-major        -> mfma_rows_kernel   (gemm_mfma.hip)
//   * reduction over the rows (weight gradients), K >= 1024 -> mfma_tn_kernel     (fp32 or ternary-plane operand)
//   * everything else (small fp layers, K <= 12 ...)         -> gemm_kernel below: 64x64x16 LDS tiles, 4x4 register
//     micro-tiles on the vector ALUs, strided operands, split-K with float atomics.
#include "common.h"

int svnet_mfma_rows(const svnet_gemm_desc& d, hipStream_t st);
int svnet_mfma_rows_split(const svnet_gemm_desc& d, hipStream_t st);
extern "C" size_t svnet_gemm_workspace_bytes(int64_t N, int64_t K);
int svnet_mfma_tn(const float* A, int64_t lda, const float* B, int64_t ldb, const uint64_t* b_sign, const uint64_t* b_nz,
                  int64_t M, int64_t P, int64_t Q, float* C, int64_t c_ps, int64_t c_qs, float alpha, int accumulate,
                  hipStream_t st, uint32_t q_tile_mask = 0);

namespace {

constexpr int BM = 64, BN = 64, BK = 16, PAD = 4;

struct GemmArgs {
    svnet_gemm_desc d;
    int64_t k_chunk;
    int atomic_out;
};

__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs ga) {
    const svnet_gemm_desc& d = ga.d;
    __shared__ float As[BK][BM + PAD];
    __shared__ float Bs[BK][BN + PAD];

    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t i0 = (int64_t)blockIdx.x * BM, j0 = (int64_t)blockIdx.y * BN;
    const int64_t k_begin = (int64_t)blockIdx.z * ga.k_chunk;
    const int64_t k_end = min(d.K, k_begin + ga.k_chunk);

    float acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;

    const bool a_k_fast = (d.a_cs == 1 || d.a_rs != 1);
    const bool b_k_fast = (d.b_rs == 1 && d.b_cs != 1);

    for (int64_t kb = k_begin; kb < k_end; kb += BK) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + 256 * r;
            int ii, kk;
            if (a_k_fast) { kk = e & 15; ii = e >> 4; } else { ii = e & 63; kk = e >> 6; }
            const int64_t gi = i0 + ii, gk = kb + kk;
            float v = 0.f;
            if (gi < d.M && gk < k_end) {
                v = d.A[gi * d.a_rs + gk * d.a_cs];
                if (d.a_scale) v *= d.a_scale[gk];
            }
            As[kk][ii] = v;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + 256 * r;
            int jj, kk;
            if (b_k_fast) { kk = e & 15; jj = e >> 4; } else { jj = e & 63; kk = e >> 6; }
            const int64_t gj = j0 + jj, gk = kb + kk;
            float v = 0.f;
            if (gj < d.N && gk < k_end) v = d.B[gk * d.b_rs + gj * d.b_cs];
            Bs[kk][jj] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            const float4 a4 = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]);
            const float a[4] = {a4.x, a4.y, a4.z, a4.w};
            const float b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(a[r], b[c], acc[r][c]);
        }
        __syncthreads();
    }

    // ---- epilogue
    float colpart[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t gi = i0 + ty * 4 + r;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t gj = j0 + tx * 4 + c;
            if (gi < d.M && gj < d.N) {
                float v = acc[r][c] * d.alpha;
                if (d.col_scale) v *= d.col_scale[gj];
                if (d.bias && blockIdx.z == 0) v += d.bias[gj];
                if (d.mask) {
                    const uint64_t w = d.mask[(gi >> 6) * d.N + gj];
                    if (!((w >> (gi & 63)) & 1ull)) v = 0.f;
                }
                colpart[c] += v;
                float* dst = d.C + gi * d.ldc + gj * d.c_cs;
                if (ga.atomic_out) atomicAdd(dst, v);
                else if (d.accumulate) *dst += v;
                else *dst = v;
            }
        }
    }
    if (d.col_sum) {
        float* red = &As[0][0];  // reuse: [16][64]
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) red[ty * 64 + tx * 4 + c] = colpart[c];
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[r * 64 + tid];
            const int64_t gj = j0 + tid;
            if (gj < d.N) atomicAdd(&svnet_slice_ptr(d.col_sum, (int)d.N)[gj], s);      // (sliced accumulator: svnet_hip.h)
        }
    }
}

// Few outputs, long reduction (the classifier's last layer: [32 x 256] . [256 x 40]): the tile kernel above would run as ONE
// workgroup walking K in 16-wide steps with two barriers each (37 us).  Here a wave owns one output and its lanes split K
// (coalesced when k is the contiguous index of both operands), combined with a DPP wave sum.
__global__ __launch_bounds__(256) void gemm_dot_kernel(svnet_gemm_desc d) {
    const int lane = threadIdx.x & 63;
    const int64_t out = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (out >= d.M * d.N) return;                       // wave-uniform
    const int64_t i = out / d.N, j = out - i * d.N;
    const float* a = d.A + i * d.a_rs;
    const float* b = d.B + j * d.b_cs;
    float s0 = 0.f, s1 = 0.f;
    int64_t k = lane;
    for (; k + 64 < d.K; k += 128) {
        float x0 = a[k * d.a_cs], x1 = a[(k + 64) * d.a_cs];
        if (d.a_scale) { x0 *= d.a_scale[k]; x1 *= d.a_scale[k + 64]; }
        s0 = fmaf(x0, b[k * d.b_rs], s0);
        s1 = fmaf(x1, b[(k + 64) * d.b_rs], s1);
    }
    if (k < d.K) {
        float x0 = a[k * d.a_cs];
        if (d.a_scale) x0 *= d.a_scale[k];
        s0 = fmaf(x0, b[k * d.b_rs], s0);
    }
    float v = wave_sum(s0 + s1) * d.alpha;
    if (lane == 0) {
        if (d.col_scale) v *= d.col_scale[j];
        if (d.bias) v += d.bias[j];
        float* dst = d.C + i * d.ldc + j * d.c_cs;
        *dst = d.accumulate ? *dst + v : v;
    }
}

__global__ void zero_strided_kernel(float* C, int64_t M, int64_t N, int64_t rs, int64_t cs) {
    const int64_t total = M * N;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x)
        C[(o / N) * rs + (o % N) * cs] = 0.f;
}

}  // namespace

extern "C" int svnet_gemm_f32(const svnet_gemm_desc* desc, void* stream) {
    SVNET_REQUIRE(desc, SVNET_E_ARG, "svnet_gemm_f32: null descriptor");
    const svnet_gemm_desc& d = *desc;
    SVNET_REQUIRE(d.M >= 0 && d.N >= 0 && d.K >= 0, SVNET_E_ARG, "svnet_gemm_f32: negative size");
    SVNET_REQUIRE(d.C && d.B && (d.A || d.a_sign), SVNET_E_ARG, "svnet_gemm_f32: null operand");
    SVNET_REQUIRE(!d.a_sign || d.a_nz, SVNET_E_ARG, "svnet_gemm_f32: ternary A needs both planes");
    SVNET_REQUIRE(d.c_cs != 0 && d.ldc != 0, SVNET_E_ARG, "svnet_gemm_f32: zero C stride");
    if (d.M == 0 || d.N == 0) return SVNET_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool plain_epi = !d.col_scale && !d.bias && !d.mask && !d.col_sum && !d.a_scale;

    // ---- ternary A (row-sliced planes over the reduction index): C(i,j) = sum_k tern(i,k) * B[k*b_rs + j]
    if (d.a_sign) {
        SVNET_REQUIRE(d.b_cs == 1 && plain_epi, SVNET_E_UNSUPPORTED, "svnet_gemm_f32: ternary A needs row-major B and a plain epilogue");
        return svnet_mfma_tn(d.B, d.b_rs, nullptr, 0, d.a_sign, d.a_nz, d.K, /*P=*/d.N, /*Q=*/d.M, d.C, /*c_ps=*/d.c_cs,
                             /*c_qs=*/d.ldc, d.alpha, d.accumulate, st, d.tern_tile_mask);
    }
    // ---- reduction over rows with both operands fp32 rows: A(i,k) = A[k*a_cs + i], B(k,j) = B[k*b_rs + j]
    if (d.a_rs == 1 && d.b_cs == 1 && d.K >= 1024 && plain_epi && d.M <= 4096 && d.N <= 4096) {
        return svnet_mfma_tn(d.A, d.a_cs, d.B, d.b_rs, nullptr, nullptr, d.K, /*P=*/d.M, /*Q=*/d.N, d.C, d.ldc, d.c_cs, d.alpha,
                             d.accumulate, st);
    }
    // ---- rows x exact-bf16 weights
    if (d.b_exact && d.a_cs == 1 && d.c_cs == 1 && d.M >= 16 && d.K >= 8) return svnet_mfma_rows(d, st);
    // ---- many rows x general fp32 weights: A and B split into three bf16 pieces each, the six leading products on the matrix cores (mfma_rows3_kernel)
    if (!d.b_exact && d.a_cs == 1 && d.c_cs == 1 && d.M >= 1024 && d.K >= 8 && d.N >= 8 && !d.col_scale && !d.mask && !d.col_sum && !d.a_scale &&
        d.workspace && d.workspace_bytes >= 3 * svnet_gemm_workspace_bytes(d.N, d.K))
        return svnet_mfma_rows_split(d, st);

    if (d.M * d.N <= 8192 && d.K >= 128 && !d.mask && !d.col_sum && d.split_k <= 1) {
        hipLaunchKernelGGL(gemm_dot_kernel, dim3((unsigned)svnet_cdiv(d.M * d.N, 4)), dim3(256), 0, st, d);
        SVNET_CHECK_LAUNCH("gemm_dot_kernel");
        return SVNET_OK;
    }
    GemmArgs ga;
    ga.d = d;
    const int64_t tiles = svnet_cdiv(d.M, BM) * svnet_cdiv(d.N, BN);
    int split = d.split_k;
    if (split <= 0) {
        split = 1;
        // few output tiles and a long reduction (the fp classifier head: [32 x 1022] . [1022 x 512] is 8 tiles - as 8 workgroups walking
        // K in 16-wide steps it took 210 us): K is cut into chunks of >= 64 until ~2048 / tiles workgroups exist
        if (tiles < 512 && d.K >= 256) {
            int64_t want = svnet_cdiv(2048, tiles);
            int64_t maxs = svnet_cdiv(d.K, 64);
            split = (int)(want < maxs ? want : maxs);
            if (split < 1) split = 1;
        }
    }
    int64_t chunk = svnet_cdiv(svnet_cdiv(d.K > 0 ? d.K : 1, split), BK) * BK;
    split = (int)svnet_cdiv(d.K > 0 ? d.K : 1, chunk);
    ga.k_chunk = chunk;
    ga.atomic_out = split > 1;
    if (ga.atomic_out && !d.accumulate) {
        hipLaunchKernelGGL(zero_strided_kernel, dim3(svnet_grid(d.M * d.N, 256)), dim3(256), 0, st, d.C, d.M, d.N, d.ldc, d.c_cs);
        SVNET_CHECK_LAUNCH("zero_strided_kernel");
    }
    SVNET_REQUIRE(svnet_cdiv(d.M, BM) <= 2147483647ll && svnet_cdiv(d.N, BN) <= 65535 && split <= 65535, SVNET_E_UNSUPPORTED,
                  "svnet_gemm_f32: grid too large");
    dim3 grid((unsigned)svnet_cdiv(d.M, BM), (unsigned)svnet_cdiv(d.N, BN), (unsigned)split);
    hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, st, ga);
    SVNET_CHECK_LAUNCH("gemm_kernel");
    return SVNET_OK;
}
