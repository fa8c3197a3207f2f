// Shared helpers for the gfx950 kernels of libsvnet_hip.so.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

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
//   [L result | SVNET_RED_SLICES x L slices | arrival counter]       (SVNET_SLICED_LEN(L) elements, svnet_hip.h)
// workgroup w adds to slice w % SVNET_RED_SLICES, and the LAST workgroup to arrive adds the slices up in a fixed order into the
// first L elements - what every consumer reads, unchanged.
template <typename T>
__device__ __forceinline__ T* svnet_slice_ptr(T* buf, int L) {
    const unsigned w = blockIdx.x + blockIdx.y * gridDim.x;
    return buf + (size_t)L * (1u + (w & (SVNET_RED_SLICES - 1)));
}
// The add into a slice, RETURNING: the old value coming back means the read-modify-write HAS been performed at the point of
// coherence, so "wait for my memory operations" before the arrival is counted really orders the sums before the counter.  (A
// no-return atomic is acknowledged when the L2 has accepted it, not when the memory side has executed it: with those, one run in a
// few hundred summed a slice that was still missing a workgroup's share - a 6 % error in a BatchNorm gradient at step 4 of a five-step
// test, nothing in the other 245 tests.)
template <typename T>
__device__ __forceinline__ void svnet_slice_add(T* p, T v) {
    const T old = atomicAdd(p, v);
    asm volatile("" :: "v"(old));
}
// How the last workgroup learns that every share is in (the ordering argument, edge by edge, with the gfx950 ISA of this function:
// DESIGN.md 4.6, profiles/r04_slices_finish_isa.txt):
//   (1) a thread's slice adds -> its arrival at the barrier: the adds are RETURNING agent-scope atomics, the value coming back means the
//       read-modify-write has been performed at the memory side (float / double atomics execute there, MI355X_MICROARCH.md "Global float
//       atomics"); the compiler waits for it (s_waitcnt vmcnt(0)) before s_barrier because the value is used;
//   (2) every thread of the workgroup -> thread 0's counter add: the workgroup barrier (workgroup-scope release / acquire fences);
//   (3) thread 0's counter add -> the last workgroup's loads (SVNET_SLICES_ACQREL, the default): thread 0 runs an AGENT-scope release
//       fence (buffer_wbl2 sc1 + s_waitcnt vmcnt(0), the wait repeated in asm: ROCm 7.2 can drop the fence's own wait when the wave's
//       scoreboard is provably empty, cdna_hip_programming.md Guideline 16 pitfall 12) in front of its relaxed counter add, and the
//       thread that draws the last ticket an AGENT-scope acquire fence (buffer_inv sc1) behind it, before the second barrier.  So
//       adds -> barrier -> release fence -> counter RMW -> (modification order of the counter) -> last RMW -> acquire fence -> barrier
//       -> loads is a happens-before chain of the HSA / LLVM AMDGPU memory model (fence-fence synchronisation through the counter,
//       cumulative over (1) and (2)), whatever the hardware does with relaxed atomics.  One release per WORKGROUP (thread 0), not
//       per wave: the per-wave __threadfence() measured in round 3 cost 0.19 ms per step; this form is measured in DESIGN.md 4.6.
//       With SVNET_SLICES_ACQREL=0 there are no agent-scope fences: correct only because returned atomics are performed and sc1 loads
//       bypass the L1 on gfx950 (the "8-B agent atomics both sides" row of the guide's hand-off table) - kept as the A/B arm only.
#ifndef SVNET_SLICES_ACQREL
#define SVNET_SLICES_ACQREL 1
#endif
template <typename T>
__device__ __forceinline__ void svnet_slices_finish(T* buf, int L) {
    __shared__ int svnet_last_wg;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* counter = reinterpret_cast<unsigned*>(buf + (size_t)L * (1 + SVNET_RED_SLICES));
#if SVNET_SLICES_ACQREL
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        const unsigned arrived = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = arrived == gridDim.x * gridDim.y * gridDim.z - 1u;
#if SVNET_SLICES_ACQREL
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#endif
        svnet_last_wg = last;
    }
    __syncthreads();
    if (svnet_last_wg) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int i = threadIdx.x; i < L; i += blockDim.x) {
            T s = 0;
#pragma unroll
            for (int sl = 0; sl < SVNET_RED_SLICES; ++sl)
                s += __hip_atomic_load(&buf[(size_t)L * (1 + sl) + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            buf[i] = s;
        }
    }
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
