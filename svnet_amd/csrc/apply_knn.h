// The apply pass of a fused edge layer that ALSO prepares the exact k-NN's candidate table of its output.
//
// sv_dgcnn_cls.py:55-65 / sv_dgcnn_partseg.py:80-100: the pooled (s, v) of conv1 / conv2 / conv3 are read next by
// get_graph_feature_sv -> knn (sv_util.py:100-101, 19-25), whose first kernel (knn_prep_*) reads the rows back, sums their squares with
// ATen's recipe and writes them transposed.  The k-NN sits on the forward's critical path (nothing else can run beside it), so its
// preparation is done here, by the kernel that has the rows in flight: a 4-wave workgroup takes TP consecutive points of a cloud,
// evaluates their rows exactly as the plain apply kernel does (the SAME functor: same expressions, same contraction), stores them to
// s_out / v_out (+ the concatenation slices) AND to LDS; wave 0 then walks the LDS rows for ||x||^2 (knn_xx_walk, contiguous-row mode:
// the feature row is cat[s, v.view(3 Ov)], sv_util.py:100) and all four waves write the channel-major table, TP consecutive points of
// a channel per store instruction - what knn_prep_rows_kernel does, minus its read of the rows.
#pragma once
#include "common.h"
#include "knn_table.h"

namespace {

// Math: { float s(int64_t p, int o) const; float v(int64_t p, int64_t b, int q, int c) const; }  - q = dd * Ov + c
template <int TP, class Math>
__device__ __forceinline__ void apply_knn_tiles(const Math& m, int64_t P, int64_t N, int Os, int Ov, float* __restrict__ s_out,
                                                float* __restrict__ v_out, float* __restrict__ s_cat, int64_t s_ld,
                                                float* __restrict__ v_cat, int64_t v_ld, float* __restrict__ xT, float* __restrict__ xx,
                                                int64_t Cpad, float* staged) {
    const int C = Os + 3 * Ov, LD = C | 1;                                        // (odd row stride: a lane per row is conflict-free)
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tiles = P / TP;                                                 // (N % TP == 0, checked on the host: a tile lies in one cloud)
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t p0 = tile * TP, b = p0 / N, n0 = p0 - b * N;
        if (xT && tile != (int64_t)blockIdx.x) __syncthreads();                   // the previous tile has been read
        // the tile's TP x Os scalars and TP x 3 Ov vector entries are contiguous runs of s_out / v_out: flat element loops (coalesced
        // whatever the widths; four independent rows of loads in flight per thread), the (row, column) cursor advanced without divisions
        {
            const int w = Os, total = TP * w, adv = 256 / w, rem = 256 - adv * w;
            int row = tid / w, col = tid - row * w;
            for (int i0 = 0; i0 < total; i0 += 256 * 4) {
                float z[4];
                int rr[4], cc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool live = i0 + 256 * u + tid < total;
                    rr[u] = live ? row : -1; cc[u] = col;
                    z[u] = live ? m.s(p0 + row, col) : 0.f;
                    row += adv; col += rem;
                    if (col >= w) { col -= w; ++row; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (rr[u] >= 0) {
                        const int64_t p = p0 + rr[u];
                        s_out[p * Os + cc[u]] = z[u];
                        if (s_cat) s_cat[p * s_ld + cc[u]] = z[u];
                        if (xT) staged[rr[u] * LD + cc[u]] = z[u];
                    }
                }
            }
        }
        {
            const int w = 3 * Ov, total = TP * w, adv = 256 / w, rem = 256 - adv * w;
            int row = tid / w, col = tid - row * w;
            for (int i0 = 0; i0 < total; i0 += 256 * 4) {
                float z[4];
                int rr[4], cc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool live = i0 + 256 * u + tid < total;
                    rr[u] = live ? row : -1; cc[u] = col;
                    const int dd = col >= 2 * Ov ? 2 : (col >= Ov ? 1 : 0);
                    z[u] = live ? m.v(p0 + row, b, col, col - dd * Ov) : 0.f;
                    row += adv; col += rem;
                    if (col >= w) { col -= w; ++row; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (rr[u] >= 0) {
                        const int64_t p = p0 + rr[u];
                        const int q = cc[u], dd = q >= 2 * Ov ? 2 : (q >= Ov ? 1 : 0);
                        v_out[p * 3 * Ov + q] = z[u];
                        if (v_cat) v_cat[(p * 3 + dd) * v_ld + (q - dd * Ov)] = z[u];
                        if (xT) staged[rr[u] * LD + Os + q] = z[u];
                    }
                }
            }
        }
        if (!xT) continue;                                                        // (no table asked for: the apply pass alone; uniform)
        __syncthreads();
        if (wave == 0 && lane < TP) {
            struct Src { const float* a; __device__ __forceinline__ float operator[](int64_t off) const { return a[off]; } } src = {staged + lane * LD};
            struct Dst { __device__ __forceinline__ void put(int64_t, float) const {} } dst;
            xx[p0 + lane] = knn_xx_walk(src, dst, (int64_t)C, N, n0 + lane, 1, /*xx_mode=*/1);
        }
        constexpr int CPI = 64 / TP;
        const int pl = lane % TP, cl = lane / TP;
        float* out = xT + (size_t)b * Cpad * N + n0 + pl;
        for (int c = wave * CPI + cl; c < (int)Cpad; c += 4 * CPI) out[(size_t)c * N] = c < C ? staged[pl * LD + c] : 0.f;   // (rows past C: zeros)
    }
}

constexpr int APPLY_KNN_TP = 32;       // points per tile (knn_prep_rows_kernel's choice: more workgroups in flight)

__host__ inline size_t apply_knn_lds_bytes(int64_t Os, int64_t Ov) { return (size_t)APPLY_KNN_TP * ((size_t)(Os + 3 * Ov) | 1) * sizeof(float); }

// what the host entry points check before the fused launch: the table layout, whole tiles per cloud, the workspace
__host__ inline bool apply_knn_supported(int64_t P, int64_t N, int64_t Os, int64_t Ov, int64_t* Cpad) {
    const int64_t C = Os + 3 * Ov;
    return P > 0 && N % APPLY_KNN_TP == 0 && C >= 8 && C <= 384 && svnet_knn_table_is_channel_major(N, C, Cpad);
}

}  // namespace
