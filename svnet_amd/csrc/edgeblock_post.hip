// Small point-level / parameter-level stages around the fused edge pass of a binarized edge layer (edgeblock.hip,
// edgeblock_bwd.hip).  Each replaces a handful of launch-bound framework ops by ONE launch:
//   svnet_edgeblock_prepare_vec_f32 : sign(W2), sign(Wz) rearranged for the per-point products U|T and Zp|Zq
//   svnet_edgeblock_bwd_mid_f32     : accumulators of the edge pass -> A = [dU | dT | dZp | dZq] rows, dbeta in reference order
//   svnet_edgeblock_bwd_params_f32  : STE chain rule to (W1, scale1), (W2, scale2), (Wz, scalez) from the three Gram products
// Reference: models/sv_layers.py:35-51 (Linear bw/ba), :172-196 (SVBlock), models/utils/sv_util.py:90-116 (edge features).
#include "common.h"

namespace {

__device__ __forceinline__ float sgn(float w) { return (w > 0.f) ? 1.f : ((w < 0.f) ? -1.f : 0.f); }

// wv [2Ov+6, Cv]: rows 0..Ov-1 = sign(W2[:, :Cv]) (acts on v_j - v_i -> U), Ov..2Ov-1 = sign(W2[:, Cv:]) (acts on v_i -> T),
// then 3 rows sign(Wz[:, :Cv]) (Zp) and 3 rows sign(Wz[:, Cv:]) (Zq).  scv [2Ov+6] = [sc2, sc2, scz, scz].
__global__ void prepare_vec_kernel(const float* __restrict__ W2, const float* __restrict__ sc2, const float* __restrict__ Wz,
                                   const float* __restrict__ scz, int Ov, int Cv, float* __restrict__ wv, float* __restrict__ scv) {
    const int R = 2 * Ov + 6;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < R * Cv + R; e += gridDim.x * blockDim.x) {
        if (e < R * Cv) {
            const int r = e / Cv, c = e - r * Cv;
            float w;
            if (r < 2 * Ov) w = W2[(r % Ov) * 2 * Cv + (r / Ov) * Cv + c];
            else { const int q = r - 2 * Ov; w = Wz[(q % 3) * 2 * Cv + (q / 3) * Cv + c]; }
            wv[e] = sgn(w);
        } else {
            const int r = e - R * Cv;
            scv[r] = r < 2 * Ov ? sc2[r % Ov] : scz[(r - 2 * Ov) % 3];
        }
    }
}

// acat[(p,a), :] = [du - dvc | dvc | dzp - dzc | dzc];  dbeta1[f] = dbeta_perm[fused column of f]
__global__ __launch_bounds__(256) void bwd_mid_kernel(const float* __restrict__ du, const float* __restrict__ dvc,
                                                      const float* __restrict__ dzp, const float* __restrict__ dzc, int64_t rows,
                                                      int Ov, float* __restrict__ acat, const float* __restrict__ dbeta_perm, int Cs,
                                                      int Cv, float* __restrict__ dbeta1) {
    const int W = 2 * Ov + 6;
    const int64_t total = rows * W;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t e = t0; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / W;
        const int c = (int)(e - r * W);
        float val;
        if (c < Ov) val = du[r * Ov + c] - dvc[r * Ov + c];
        else if (c < 2 * Ov) val = dvc[r * Ov + c - Ov];
        else if (c < 2 * Ov + 3) val = dzp[r * 3 + c - 2 * Ov] - dzc[r * 3 + c - 2 * Ov];
        else val = dzc[r * 3 + c - 2 * Ov - 3];
        acat[e] = val;
    }
    const int K1 = 2 * Cs + 6 * Cv;
    if (t0 < K1) {
        const int f = (int)t0, g = f - 2 * Cs;
        const int col = f < Cs ? f : (f < 2 * Cs ? 64 + f - Cs : 128 + 64 * (g % 3) + g / 3);
        dbeta1[f] = dbeta_perm[col];
    }
}

// One wave per weight row.  Rows [0,Os): linear1 from GXp [Os,320] (fused column order); [Os,Os+Ov): linear2 from
// GXc[0:2Ov] ([o] = U part, [Ov+o] = T part); last 3: v2s frame from GXc[2Ov:2Ov+6].  Gradients are ASSIGNED.
__global__ __launch_bounds__(256) void bwd_params_kernel(const float* __restrict__ GXp, const float* __restrict__ GXc,
                                                         const float* __restrict__ W1, const float* __restrict__ sc1,
                                                         const float* __restrict__ W2, const float* __restrict__ sc2,
                                                         const float* __restrict__ Wz, const float* __restrict__ scz, int Os, int Ov,
                                                         int Cs, int Cv, float* __restrict__ dW1, float* __restrict__ dsc1,
                                                         float* __restrict__ dW2, float* __restrict__ dsc2, float* __restrict__ dWz,
                                                         float* __restrict__ dscz) {
    const int lane = threadIdx.x & 63;
    const int row = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (row >= Os + Ov + 3) return;
    float part = 0.f;
    if (row < Os) {
        const int K1 = 2 * Cs + 6 * Cv;
        const float sc = sc1[row];
        for (int f = lane; f < K1; f += 64) {
            const int g = f - 2 * Cs;
            const int col = f < Cs ? f : (f < 2 * Cs ? 64 + f - Cs : 128 + 64 * (g % 3) + g / 3);
            const float w = W1[row * K1 + f], gx = GXp[row * 320 + col];
            part += sgn(w) * gx;
            dW1[row * K1 + f] = (fabsf(w) <= 1.2f) ? sc * gx : 0.f;
        }
        part = wave_sum(part);
        if (lane == 0) dsc1[row] = part;
        return;
    }
    const bool is2 = row < Os + Ov;
    const int o = is2 ? row - Os : row - Os - Ov;
    const int O = is2 ? Ov : 3;
    const float* W = is2 ? W2 : Wz;
    const float* G = is2 ? GXc : GXc + (int64_t)2 * Ov * Cv;
    float* dW = is2 ? dW2 : dWz;
    const float sc = is2 ? sc2[o] : scz[o];
    for (int f = lane; f < 2 * Cv; f += 64) {
        const int half = f >= Cv, c = f - half * Cv;
        const float w = W[o * 2 * Cv + f], gx = G[(o + half * O) * Cv + c];
        part += sgn(w) * gx;
        dW[o * 2 * Cv + f] = (fabsf(w) <= 1.2f) ? sc * gx : 0.f;
    }
    part = wave_sum(part);
    if (lane == 0) (is2 ? dsc2 : dscz)[o] = part;
}

}  // namespace

extern "C" int svnet_edgeblock_prepare_vec_f32(const float* W2, const float* scale2, const float* Wz, const float* scalez, int64_t Ov,
                                               int64_t Cv, float* wv, float* scv, void* stream) {
    SVNET_REQUIRE(W2 && scale2 && Wz && scalez && wv && scv && Ov > 0 && Cv > 0, SVNET_E_ARG, "svnet_edgeblock_prepare_vec_f32: bad arguments");
    const int64_t n = (2 * Ov + 6) * (Cv + 1);
    hipLaunchKernelGGL(prepare_vec_kernel, dim3(svnet_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, W2, scale2, Wz, scalez, (int)Ov,
                       (int)Cv, wv, scv);
    SVNET_CHECK_LAUNCH("prepare_vec_kernel");
    return SVNET_OK;
}

extern "C" int svnet_edgeblock_bwd_mid_f32(const float* du_acc, const float* dvc, const float* dzp_acc, const float* dzc, int64_t P,
                                           int64_t Ov, float* acat, const float* dbeta_perm, int64_t Cs, int64_t Cv, float* dbeta1,
                                           void* stream) {
    SVNET_REQUIRE(du_acc && dvc && dzp_acc && dzc && acat && dbeta_perm && dbeta1 && P > 0 && Ov > 0, SVNET_E_ARG,
                  "svnet_edgeblock_bwd_mid_f32: bad arguments");
    SVNET_REQUIRE(Cs > 0 && Cs <= 64 && Cv > 0 && 2 * Cv <= 64, SVNET_E_UNSUPPORTED, "svnet_edgeblock_bwd_mid_f32: needs Cs <= 64, 2*Cv <= 64");
    hipLaunchKernelGGL(bwd_mid_kernel, dim3(svnet_grid(3 * P * (2 * Ov + 6), 256)), dim3(256), 0, (hipStream_t)stream, du_acc, dvc,
                       dzp_acc, dzc, 3 * P, (int)Ov, acat, dbeta_perm, (int)Cs, (int)Cv, dbeta1);
    SVNET_CHECK_LAUNCH("bwd_mid_kernel");
    return SVNET_OK;
}

extern "C" int svnet_edgeblock_bwd_params_f32(const float* GXp, const float* GXc, const float* W1, const float* scale1, const float* W2,
                                              const float* scale2, const float* Wz, const float* scalez, int64_t Os, int64_t Ov,
                                              int64_t Cs, int64_t Cv, float* dW1, float* dscale1, float* dW2, float* dscale2, float* dWz,
                                              float* dscalez, void* stream) {
    SVNET_REQUIRE(GXp && GXc && W1 && scale1 && W2 && scale2 && Wz && scalez && dW1 && dscale1 && dW2 && dscale2 && dWz && dscalez,
                  SVNET_E_ARG, "svnet_edgeblock_bwd_params_f32: null pointer");
    SVNET_REQUIRE(Cs > 0 && Cs <= 64 && Cv > 0 && 2 * Cv <= 64 && Os > 0 && Ov > 0, SVNET_E_UNSUPPORTED,
                  "svnet_edgeblock_bwd_params_f32: needs Cs <= 64, 2*Cv <= 64");
    hipLaunchKernelGGL(bwd_params_kernel, dim3((unsigned)svnet_cdiv((Os + Ov + 3) * 64, 256)), dim3(256), 0, (hipStream_t)stream, GXp, GXc,
                       W1, scale1, W2, scale2, Wz, scalez, (int)Os, (int)Ov, (int)Cs, (int)Cv, dW1, dscale1, dW2, dscale2, dWz, dscalez);
    SVNET_CHECK_LAUNCH("bwd_params_kernel");
    return SVNET_OK;
}
