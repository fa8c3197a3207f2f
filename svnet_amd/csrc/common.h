// Shared helpers for the gfx950 kernels of libsvnet_hip.so.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/svnet_hip.h"

#define SVNET_WAVE 64

void svnet_set_error(const char* fmt, ...);

#define SVNET_REQUIRE(cond, code, ...)        \
    do {                                      \
        if (!(cond)) {                        \
            svnet_set_error(__VA_ARGS__);     \
            return (code);                    \
        }                                     \
    } while (0)

#define SVNET_CHECK_LAUNCH(name)                                                   \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            svnet_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return SVNET_E_LAUNCH;                                                 \
        }                                                                          \
    } while (0)

static inline int64_t svnet_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// More than 64 KiB of dynamic LDS needs an explicit opt-in, and hipFuncSetAttribute applies to the CURRENT device only: one bit per
// device ordinal and call site (a process that touches a second GPU opts in there too), atomic (two host threads may race: the call is
// idempotent), and the hipError_t is reported instead of dropped - the message of the launch failure that follows would not name it.
static inline bool svnet_lds_optin(std::atomic<uint64_t>& done, const void* const* fns, int nf, int bytes, const char* name) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) {
        const uint64_t bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return true;
        for (int i = 0; i < nf && e == hipSuccess; ++i) e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) { done.fetch_or(bit, std::memory_order_release); return true; }
    }
    svnet_set_error("%s: cannot opt in to %d bytes of dynamic LDS on device %d: %s", name, bytes, dev, hipGetErrorString(e));
    return false;
}
// SVNET_LDS_OPTIN(ok, bytes, name, kernel, ...): ok = false (and svnet_last_error set) when the opt-in failed on this device
#define SVNET_LDS_OPTIN(ok, bytes, name, ...)                                                                 \
    do {                                                                                                      \
        static std::atomic<uint64_t> done_{0};                                                                \
        const void* const fns_[] = {__VA_ARGS__};                                                             \
        (ok) = svnet_lds_optin(done_, fns_, (int)(sizeof(fns_) / sizeof(fns_[0])), (int)(bytes), name);       \
    } while (0)

// Grid for memory-bound grid-stride kernels: enough blocks to fill 256 CUs x 8, capped.
static inline unsigned svnet_grid(int64_t work_items, int block, int64_t cap = 256 * 16) {
    int64_t g = svnet_cdiv(work_items, block);
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

#ifdef __HIPCC__
// Sums over groups of G consecutive lanes (G = 4 .. 64), every lane of a group ending up with its group's sum, WITHOUT the LDS
// crossbar: __shfl_xor is a ds_bpermute (an LDS round trip per step, six dependent ones per sum); here the steps inside a 16-lane
// row are DPP operand modifiers of the adds themselves and the two steps across rows are the gfx950 row / half swaps.
template <int CTRL>
__device__ __forceinline__ float svnet_dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int G>
__device__ __forceinline__ float group_sum_dpp(float v) {
    static_assert(G == 1 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "group size");
    if (G >= 4) {
        v += svnet_dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
        v += svnet_dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
    }
    if (G >= 8) v += svnet_dpp_f32<0x141>(v);    // row_half_mirror: the other quad of the 8
    if (G >= 16) v += svnet_dpp_f32<0x140>(v);   // row_mirror: the other half of the row
    if (G >= 32) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);      // rows 0+1 | 0+1 | 2+3 | 2+3
    }
    if (G >= 64) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum_dpp<64>(v); }
// ---- grid-wide column sums without a thousand adders per address.  A reduction kernel used to end in one atomic per output per
// workgroup onto the SAME L addresses; same-address atomics are served one after the other at the memory side (~2.5 ns each per cache
// line, measured: 512 workgroups x 340 doubles = 12 us of a 39 us kernel, the float reductions twice that), and removing them from
// the BatchNorm / VectorBN reductions of one step was worth 0.065 ms.  Now the caller's (zero-filled) buffer holds
//   [L result | SVNET_RED_SLICES x L slices | 2 spare]               (SVNET_SLICED_LEN(L) elements, svnet_hip.h)
// workgroup w adds to slice w % SVNET_RED_SLICES and NOTHING else happens in the reducing kernel: the slices are added up by the
// kernel that consumes the sums (svnet_slices_total: bn_finalize / vbn_fwd / the *_bwd_apply kernels, each of which also leaves the
// totals in the first L elements), or by svnet_slices_sum_* where no such kernel follows.  The hand-off between the adders and the
// reader is therefore a KERNEL BOUNDARY on one stream - the only inter-workgroup ordering HIP guarantees without a protocol.
// (Round 3 summed the slices in the reducing kernel's last workgroup to arrive, ordered by returning atomics + an arrival counter;
//  one run of its no-return form lost a share.  The memory-model-conforming form of that hand-off - an agent-scope release fence by
//  one thread per workgroup in front of the counter add, an acquire fence in the last arriver - measured +0.05 ms per step
//  (4.657 against 4.606 ms, three alternating runs on one box, gpurun_out/r04_ab_slices.log); this form needs no hand-off at all.)
template <typename T>
__device__ __forceinline__ T* svnet_slice_ptr(T* buf, int L) {
    const unsigned w = blockIdx.x + blockIdx.y * gridDim.x;
    return buf + (size_t)L * (1u + (w & (SVNET_RED_SLICES - 1)));
}
template <typename T>
__device__ __forceinline__ void svnet_slice_add(T* p, T v) { atomicAdd(p, v); }
// Sum i of a sliced accumulator a PREVIOUS kernel of the stream filled (fixed order: bit-reproducible for a given set of slice values).
template <typename T>
__device__ __forceinline__ T svnet_slices_total(const T* buf, int L, int i) {
    T s = 0;
#pragma unroll
    for (int sl = 0; sl < SVNET_RED_SLICES; ++sl) s += buf[(size_t)L * (1 + sl) + i];
    return s;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
// Single-instruction square root / reciprocal (v_sqrt_f32, v_rcp_f32: 1 ulp each) for the per-edge vector norms of the fused
// kernels, where the correctly rounded sequences (10 instructions each) were a fifth of the instruction stream.
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Value of lane (l ^ S) without the LDS crossbar: DPP modifiers inside a 16-lane row, the gfx950 row / half swaps across rows.
template <int CTRL>
__device__ __forceinline__ uint32_t svnet_dpp_u32(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false);
}
template <int S>
__device__ __forceinline__ uint32_t svnet_lane_xor_u32(uint32_t x, int lane) {
    if (S == 1) return svnet_dpp_u32<0xB1>(x);                       // quad_perm [1,0,3,2]
    if (S == 2) return svnet_dpp_u32<0x4E>(x);                       // quad_perm [2,3,0,1]
    if (S == 4) {                                                    // rotate the row by 4 either way, keep the one that is l ^ 4
        const uint32_t a = svnet_dpp_u32<0x124>(x), b = svnet_dpp_u32<0x12C>(x);
        return (lane & 4) ? a : b;
    }
    if (S == 8) return svnet_dpp_u32<0x128>(x);                      // row_ror:8
    if (S == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return (lane & 16) ? r[0] : r[1];
    }
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (lane & 32) ? r[0] : r[1];
}
// 32 x 32 bit-matrix transpose across 32 consecutive lanes (both halves of the wave at once): lane r of a half holds row r
// (bit b = element [r][b]); afterwards lane b holds column b (bit r = element [r][b]).  Five butterfly stages of ~7 instructions
// (swap the off-diagonal J x J blocks of every 2J x 2J block) instead of 32 ballots + selects.
template <int J>
__device__ __forceinline__ uint32_t svnet_bt_stage(uint32_t x, int lane) {
    constexpr uint32_t MLO = J == 16 ? 0x0000FFFFu : J == 8 ? 0x00FF00FFu : J == 4 ? 0x0F0F0F0Fu : J == 2 ? 0x33333333u : 0x55555555u;
    const uint32_t p = svnet_lane_xor_u32<J>(x, lane);
    const bool lower = (lane & J) != 0;
    const uint32_t keep = lower ? (x & ~MLO) : (x & MLO);
    const uint32_t take = lower ? ((p >> J) & MLO) : ((p & MLO) << J);
    return keep | take;
}
__device__ __forceinline__ uint32_t svnet_bit_transpose32(uint32_t x, int lane) {
    x = svnet_bt_stage<16>(x, lane);
    x = svnet_bt_stage<8>(x, lane);
    x = svnet_bt_stage<4>(x, lane);
    x = svnet_bt_stage<2>(x, lane);
    return svnet_bt_stage<1>(x, lane);
}

// A float at wave-uniform `base` + per-lane BYTE offset, loaded with the SGPR-base addressing mode (global_load v, voff, s[base]).
// The empty asm keeps the 32->64-bit extension of the lane offset next to the load: once it is hoisted out of a loop, instruction
// selection no longer sees it and falls back to a 64-bit VALU add per load.
__device__ __forceinline__ uint32_t keep_here(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ float ld_f32_sbase(const float* base, uint32_t byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + keep_here(byte_off));
}
__device__ __forceinline__ int16_t ld_i16_sbase(const int16_t* base, uint32_t byte_off) {
    return *reinterpret_cast<const int16_t*>(reinterpret_cast<const char*>(base) + keep_here(byte_off));
}
__device__ __forceinline__ void st_f32_sbase(float* base, uint32_t byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + keep_here(byte_off)) = v;
}

#endif
