// microbench: issue rate of v_pk_fma_f32 (with a broadcast operand) vs v_fma_f32 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float2v acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = float2v{(float)threadIdx.x + i, (float)i};
    float2v q = {a, b}, c = {b, a};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (MODE == 0) {        // pk, plain pairs
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(q), "v"(c));
            } else if (MODE == 1) { // pk, src0 low half broadcast
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(q), "v"(c));
            } else if (MODE == 2) { // two scalar fmas
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(q.x), "v"(c.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(q.x), "v"(c.y));
            } else if (MODE == 3) { // v_fmac (VOP2)
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].x) : "v"(q.x), "v"(c.x));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].y) : "v"(q.x), "v"(c.y));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int wg_per_cu) {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    int grid = 256 * wg_per_cu, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(out, 10, 1.f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<grid, 256>>>(out, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fma = (double)grid * 256 * iters * 64;   // 32 pairs = 64 FMAs per thread per iter
    printf("%-28s wg/cu %d: %.3f ms  %.1f TFLOP/s  (%.2f ns per wave per 32 pair-FMAs)\n", name, wg_per_cu, ms, 2 * fma / ms * 1e-9, ms * 1e6 / iters / wg_per_cu);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_pk_fma_f32", w);
        run<1>("v_pk_fma_f32 bcast", w);
        run<2>("2 x v_fma_f32", w);
        run<3>("2 x v_fmac_f32", w);
    }
    return 0;
}
