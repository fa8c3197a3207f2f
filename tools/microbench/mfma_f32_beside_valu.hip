// microbench: f32 MFMA 16x16x4 and plain VALU work, alone and side by side on the same SIMDs (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// role 0: MFMA loop (16 accumulators), role 1: VALU loop (v_fma + v_max + v_cndmask mix), role 2: by workgroup parity (w>>8)&1
template <int ROLE>
__global__ __launch_bounds__(512) void k(float* out, int it_m, int it_v) {
    int role = ROLE;
    if (ROLE == 2) role = (blockIdx.x >> 8) & 1;
    if (ROLE == 3) role = (threadIdx.x >> 8) & 1;   // waves 0-3 MFMA, waves 4-7 VALU: one of each per SIMD
    float s = 0;
    if (role == 0) {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        float a = threadIdx.x * 0.001f, b = 0.5f;
        for (int it = 0; it < it_m; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    } else {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
        for (int it = 0; it < it_v; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(0.999f), "v"(0.5f));
                asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int ROLE>
float run(float* out, int grid, int it_m, int it_v) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<ROLE><<<grid, 512>>>(out, 10, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<ROLE><<<grid, 512>>>(out, it_m, it_v); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* out; (void)hipMalloc(&out, 512 * 8192 * 4);
    const int it_m = 4000, it_v = 4000;   // per wave: 64000 MFMAs (x32 cyc = 2.05M cyc) | 128000 VALU (x4 cyc = 0.51M cyc)
    for (int wg : {256, 512}) {
        float m = run<0>(out, wg, it_m, it_v), v = run<1>(out, wg, it_m, it_v), mv = run<2>(out, wg, it_m, it_v), mv3 = run<3>(out, wg, it_m, it_v);
        double tf = (double)wg * 4 * it_m * 16 * 2048.0 / (m * 1e-3) * 1e-12;
        printf("grid %4d: MFMA only %.3f ms (%.1f TFLOP/s), VALU only %.3f ms, half/half by (w>>8)&1 %.3f ms, in-WG waves 0-3 | 4-7 %.3f ms\n", wg, m, tf, v, mv, mv3);
    }
    // equal-duration roles: VALU loop 4x longer so both take ~ the same time alone
    for (int wg : {256, 512}) {
        float m = run<0>(out, wg, it_m, 4 * it_v), v = run<1>(out, wg, it_m, 4 * it_v), mv = run<2>(out, wg, it_m, 4 * it_v), mv3 = run<3>(out, wg, it_m, 4 * it_v);
        printf("grid %4d (VALU x4): MFMA only %.3f ms, VALU only %.3f ms, half/half by (w>>8)&1 %.3f ms, in-WG waves 0-3 | 4-7 %.3f ms\n", wg, m, v, mv, mv3);
    }
    return 0;
}
