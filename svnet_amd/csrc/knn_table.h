// The ||x||^2 recipe of the exact k-NN (knn.hip) and the layout of its candidate table, shared with the kernels that PREPARE that table
// while they write the features it is built from (apply_knn.h: the pooled outputs of a fused edge layer, whose next reader is the k-NN
// of the following level - sv_dgcnn_cls.py:55-65).
//
// Every rounding below is intentional: the functions switch contraction off for their own bodies (`#pragma clang fp contract(off)` -
// kt_add(a, kt_mul(v, v)) IS contracted into an fma under hipcc's default -ffp-contract=fast: measured, 1 - 15 % of the rows got
// another ||x||^2), so the results do not depend on the including translation unit's setting; tests/test_hip_fused.py compares the tables
// of both producers bit for bit.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

// Layout of the table svnet_knn_f32 builds in its workspace for (N, C): channel-major xT[b][c][n] with *Cpad >= C rows per cloud (rows
// past C hold zeros), followed by xx[b][n].  Returns false when this build / environment uses the four-channel-interleaved layout instead
// (SVNET_KNN_MFMA diagnostic form), which the fused producers do not write.  Defined in knn.hip.
bool svnet_knn_table_is_channel_major(int64_t N, int64_t C, int64_t* Cpad);

namespace {

// one rounded sum / product.  (HIP's __fadd_rn / __fmul_rn are header functions compiled under the including file's contraction mode:
// inlined, their fmul + fadd pairs fuse.  These carry the pragma in their own bodies.)
__device__ __forceinline__ float kt_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float kt_mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}

struct Cascade {  // ATen multi_row_sum: 4 levels, level step 16
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
    __device__ __forceinline__ void add(float v) {
#pragma clang fp contract(off)
        a0 = kt_add(a0, v);
        ++i;
        if ((i & 15) == 0) {
            a1 = kt_add(a1, a0);
            a0 = 0.f;
            if ((i & 0xF0) == 0) {
                a2 = kt_add(a2, a1);
                a1 = 0.f;
                if ((i & 0xF00) == 0) {
                    a3 = kt_add(a3, a2);
                    a2 = 0.f;
                }
            }
        }
    }
    __device__ __forceinline__ float total() const {
#pragma clang fp contract(off)
        return kt_add(kt_add(kt_add(a0, a1), a2), a3);
    }
};

// The walk over one point's channels: ||x||^2 with ATen's exact recipe for the layout torch reduces (xx_mode), every value handed
// to dst.put on the way (the transposition; a no-op for callers that transpose separately).
template <class SrcT, class DstT>
__device__ __forceinline__ float knn_xx_walk(const SrcT& src, const DstT& dst, int64_t C, int64_t N, int64_t n, int64_t sc, int xx_mode) {
#pragma clang fp contract(off)
    float result;
    if (xx_mode == 0) {
        // outer-dim reduction: columns n < 32*floor(N/32) use one cascade, the rest ATen's row_sum (ilp 4)
        if (n < (N / 32) * 32) {
            Cascade cs;
            for (int64_t c = 0; c < C; ++c) {
                float v = src[c * sc];
                dst.put(c, v);
                cs.add(kt_mul(v, v));
            }
            result = cs.total();
        } else {
            Cascade part[4];
            const int64_t ng = C / 4;
            for (int64_t g = 0; g < ng; ++g) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = src[(4 * g + r) * sc];
                    dst.put((4 * g + r), v);
                    part[r].add(kt_mul(v, v));
                }
            }
            float p0 = part[0].total(), p1 = part[1].total(), p2 = part[2].total(), p3 = part[3].total();
            for (int64_t c = ng * 4; c < C; ++c) {
                float v = src[c * sc];
                dst.put(c, v);
                p0 = kt_add(p0, kt_mul(v, v));
            }
            result = kt_add(kt_add(kt_add(p0, p1), p2), p3);
        }
    } else if (C < 8) {
        // contiguous-dim reduction of a row shorter than one 8-lane vector: scalar row_sum (ilp 4)
        float part[4] = {0.f, 0.f, 0.f, 0.f};
        const int64_t ng = C / 4;
        for (int64_t g = 0; g < ng; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = src[(4 * g + r) * sc];
                dst.put((4 * g + r), v);
                part[r] = kt_add(part[r], kt_mul(v, v));
            }
        }
        for (int64_t c = ng * 4; c < C; ++c) {
            float v = src[c * sc];
            dst.put(c, v);
            part[0] = kt_add(part[0], kt_mul(v, v));
        }
        result = kt_add(kt_add(kt_add(part[0], part[1]), part[2]), part[3]);
    } else {
        // contiguous-dim reduction: 8-lane vectors, 4 interleaved vector accumulators (C <= 384 < 512,
        // so the inner cascade never spills a level), leftover vectors into accumulator 0,
        // lanes combined p0+p1+p2+p3, then scalar tail first, then the 8 lanes in order.
        float p[4][8];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int l = 0; l < 8; ++l) p[r][l] = 0.f;
        const int64_t nv = C / 8, ng = nv / 4;
        for (int64_t g = 0; g < ng; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int l = 0; l < 8; ++l) {
                    const int64_t c = (4 * g + r) * 8 + l;
                    float v = src[c * sc];
                    dst.put(c, v);
                    p[r][l] = kt_add(p[r][l], kt_mul(v, v));
                }
        }
        for (int64_t g = ng * 4; g < nv; ++g) {
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                const int64_t c = g * 8 + l;
                float v = src[c * sc];
                dst.put(c, v);
                p[0][l] = kt_add(p[0][l], kt_mul(v, v));
            }
        }
        float fin = 0.f;
        for (int64_t c = nv * 8; c < C; ++c) {
            float v = src[c * sc];
            dst.put(c, v);
            fin = kt_add(fin, kt_mul(v, v));
        }
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            float lane = kt_add(kt_add(kt_add(p[0][l], p[1][l]), p[2][l]), p[3][l]);
            fin = kt_add(fin, lane);
        }
        result = fin;
    }
    return result;
}

}  // namespace
