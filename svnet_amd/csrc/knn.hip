// k-NN in feature space, bit-exact against the reference's torch-CPU evaluation.
//
// Replaces  models/utils/sv_util.py:19-25 (knn):  -2*matmul(x^T,x), sum(x**2), pd, topk.
// Arithmetic contract (re-derived in oracle/knn_exact.c):
//   dot(i,j)  = sequential fp32 FMA chain over channels, first term a rounded product
//   xx        = ATen's cascade / ilp-4 / 8-lane vectorised sums of separately rounded squares
//   pd(i,j)   = fl( fl(-xx[j] + 2*dot) - xx[i] )
//   idx       = k largest pd, descending, ties -> lowest index
// This file must be compiled with -ffp-contract=off: every rounding is intentional.
//
// Layout in HBM: a prep kernel writes xT[b][c][n] (channel-major, so that the 64 lanes of a wave
// read 64 consecutive candidates of one channel = one 256-B line) and xx[b][n].  The main kernel
// never materialises the N x N matrix: a wave keeps Q query rows x (64*T) candidate distances in
// registers (Q*T accumulators per lane), query values arrive through scalar loads, and the top-k
// is selected straight from those registers with wave-wide arg-max reductions.
#include <stdlib.h>

#include "common.h"
#include "knn_table.h"

#ifndef SVNET_KNN_ABL
#define SVNET_KNN_ABL 0
#endif
#ifndef SVNET_KNN_FORCE_MF
#define SVNET_KNN_FORCE_MF 0
#endif
#ifndef SVNET_KNN_MF8
#define SVNET_KNN_MF8 1
#endif
#ifndef SVNET_KNN_NOSLOW
#define SVNET_KNN_NOSLOW 0      // (diagnostic builds: the four-query selection without its one-at-a-time fall-back)
#endif
#ifndef SVNET_KNN_TP
#define SVNET_KNN_TP 32
#endif


namespace {

// One thread per point: transpose to channel-major and compute ||x||^2 with the exact recipe.
// Two sources (x2 != nullptr): channel c < split comes from x, the rest from row p of x2 [B*N, C - split] - the feature rows
// cat[s, v.view(B,N,3Cv)] of get_graph_feature_sv (sv_util.py:100) read where they lie, without materialising the cat.
template <bool IL4>
__global__ __launch_bounds__(256) void knn_prep_kernel(const float* __restrict__ x, int64_t B, int64_t N, int64_t C,
                                                       int64_t sb, int64_t sn, int64_t sc, int xx_mode,
                                                       float* __restrict__ xT, float* __restrict__ xx,
                                                       const float* __restrict__ x2, int64_t split, int64_t Cpad) {
    constexpr int il4 = IL4 ? 1 : 0;
    const int64_t total = B * N;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = p / N, n = p % N;
        const float* src0 = x + b * sb + n * sn;
        const float* src1 = x2 ? x2 + p * (C - split) - split * sc : src0;      // so that src1[c * sc] is channel c >= split
        const int64_t cut = x2 ? split : C;
        struct Src {
            const float* a; const float* b2; int64_t cut, sc;
            __device__ __forceinline__ float operator[](int64_t off) const { return (off < cut * sc) ? a[off] : b2[off]; }
        } src = {src0, src1, cut, sc};
        // il4: channels interleaved in fours, element (c, n) at ((c >> 2) * N + n) * 4 + (c & 3) of the cloud's table of 4*ceil(C/4) channels
        // (what the matrix-core form of the main kernel reads: 16 candidates x 4 channels = one 256-byte run); else channel-major [c][n]
        const int64_t C4 = (C + 3) & ~(int64_t)3;
        struct Dst {
            float* base; int64_t N, n;
            __device__ __forceinline__ void put(int64_t c, float v) const {
                if (IL4) base[((c >> 2) * N + n) * 4 + (c & 3)] = v; else base[c * N + n] = v;
            }
        } dst = {xT + b * (il4 ? C4 : Cpad) * N, N, n};                              // (channel-major: Cpad >= C rows per cloud)
        for (int64_t c = C; c < (il4 ? C4 : Cpad); ++c) dst.put(c, 0.f);             // padding channels: zeros (never NaN bit patterns)
        xx[p] = knn_xx_walk(src, dst, C, N, n, sc, xx_mode);
    }
}

// Contiguous feature rows ([B*N, cut] (+ [B*N, C - cut]); N % 64 == 0): a thread walking its own row in global memory makes every load
// of its wave touch 64 cache lines - the generic kernel is bound by the CUs' address units there (22 us at C = 127 for 16.6 MB).  Here a
// 4-wave workgroup takes 64 consecutive points: their rows, one contiguous run per source, are copied into LDS with coalesced loads
// (row stride C | 1: a lane per row is conflict-free), wave 0 walks them for ||x||^2 - same arithmetic, same order - and all four
// waves write the channel-major table, 64 consecutive points of a channel per store instruction.
template <int TP>   // points per tile (32 or 64): smaller tiles = more workgroups in flight; the kernel is a chain of latencies, not of bytes
__global__ __launch_bounds__(256) void knn_prep_rows_kernel(const float* __restrict__ x, int64_t B, int64_t N, int64_t C, int xx_mode,
                                                            float* __restrict__ xT, float* __restrict__ xx,
                                                            const float* __restrict__ x2, int64_t split, int64_t Cpad) {
    extern __shared__ float staged[];                                             // [TP][C | 1]
    const int LD = (int)C | 1;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tiles = B * N / TP;
    const int cut = (int)(x2 ? split : C);
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t p0 = tile * TP;
        if (tile != (int64_t)blockIdx.x) __syncthreads();                         // the previous tile has been read
        for (int part = 0; part < (x2 ? 2 : 1); ++part) {
            const int w = part ? (int)C - cut : cut, c0 = part ? cut : 0, total = TP * w;
            const float* g = (part ? x2 : x) + p0 * w;
            int row = tid / w, col = tid - row * w;
            for (int i0 = 0; i0 < total; i0 += 256 * 16) {                        // sixteen coalesced loads in flight per thread
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int i = i0 + 256 * u + tid; v[u] = i < total ? g[i] : 0.f; }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (i0 + 256 * u + tid < total) staged[row * LD + c0 + col] = v[u];
                    col += 256;
                    while (col >= w) { col -= w; ++row; }
                }
            }
        }
        __syncthreads();
        const int64_t b = p0 / N, n0 = p0 - b * N;
        if (wave == 0 && lane < TP) {
            struct Src { const float* a; __device__ __forceinline__ float operator[](int64_t off) const { return a[off]; } } src = {staged + lane * LD};
            struct Dst { __device__ __forceinline__ void put(int64_t, float) const {} } dst;
            xx[p0 + lane] = knn_xx_walk(src, dst, C, N, n0 + lane, 1, xx_mode);
        }
        // the channel-major table: TP consecutive points of a channel per store instruction (64 / TP channels per wave and instruction)
        constexpr int CPI = 64 / TP;
        const int pl = lane % TP, cl = lane / TP;
        float* out = xT + (size_t)b * Cpad * N + n0 + pl;
        for (int c = wave * CPI + cl; c < (int)Cpad; c += 4 * CPI) out[(size_t)c * N] = c < (int)C ? staged[pl * LD + c] : 0.f;   // (rows past C: zeros)
    }
}

// (value, index) arg-max across the wave; larger value wins, equal values -> smaller index.
__device__ __forceinline__ void wave_argmax(float& v, int& j) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oj = __shfl_xor(j, off, 64);
        const bool take = (ov > v) || (ov == v && oj < j);
        v = take ? ov : v;
        j = take ? oj : j;
    }
}

// "a ranks before b": larger value first, equal values -> smaller index first
__device__ __forceinline__ bool ranks_before(float va, int ja, float vb, int jb) { return (va > vb) || (va == vb && ja < jb); }

// Value of lane (l ^ S) without the LDS crossbar: DPP modifiers inside a 16-lane row, the gfx950 row / half swaps across rows
// (a bitonic sort is a chain of 21 dependent exchanges, so the ~100-cycle ds_bpermute round trip was its whole cost).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);      // (every lane has a source: bound_ctrl only spares the v_mov of "old")
}
template <int S>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t x, int lane) {
    if (S == 1) return dpp_u32<0xB1>(x);                       // quad_perm [1,0,3,2]
    if (S == 2) return dpp_u32<0x4E>(x);                       // quad_perm [2,3,0,1]
    if (S == 4) {                                              // rotate the row by 4 either way, keep the one that is l ^ 4
        const uint32_t a = dpp_u32<0x124>(x), b = dpp_u32<0x12C>(x);   // row_ror:4 reads lane l-4, row_ror:12 reads lane l+4 (mod 16)
        return (lane & 4) ? a : b;
    }
    if (S == 8) return dpp_u32<0x128>(x);                      // row_ror:8
    if (S == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return (lane & 16) ? r[0] : r[1];
    }
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (lane & 32) ? r[0] : r[1];
}

// Bitonic sort of one (value, index) pair per lane across the wave; afterwards lane 0 holds the best pair, lane 63 the worst.
template <int K2, int S2>
__device__ __forceinline__ void sort_step(float& v, int& j, int lane) {
    const float ov = __uint_as_float(lane_xor_u32<S2>(__float_as_uint(v), lane));
    const int oj = (int)lane_xor_u32<S2>((uint32_t)j, lane);
    const bool desc = (lane & K2) == 0 || K2 == 64;   // final merge: whole wave descending
    const bool lower = (lane & S2) == 0;
    const bool other_first = (ov > v) || (ov == v && oj < j);
    // in a descending block the lower lane keeps the pair that ranks first, the upper lane the other one
    const bool take = (lower == desc) ? other_first : !other_first;
    v = take ? ov : v;
    j = take ? oj : j;
}
__device__ __forceinline__ void wave_sort_desc(float& v, int& j, int lane) {
    sort_step<2, 1>(v, j, lane);
    sort_step<4, 2>(v, j, lane); sort_step<4, 1>(v, j, lane);
    sort_step<8, 4>(v, j, lane); sort_step<8, 2>(v, j, lane); sort_step<8, 1>(v, j, lane);
    sort_step<16, 8>(v, j, lane); sort_step<16, 4>(v, j, lane); sort_step<16, 2>(v, j, lane); sort_step<16, 1>(v, j, lane);
    sort_step<32, 16>(v, j, lane); sort_step<32, 8>(v, j, lane); sort_step<32, 4>(v, j, lane); sort_step<32, 2>(v, j, lane);
    sort_step<32, 1>(v, j, lane);
    sort_step<64, 32>(v, j, lane); sort_step<64, 16>(v, j, lane); sort_step<64, 8>(v, j, lane); sort_step<64, 4>(v, j, lane);
    sort_step<64, 2>(v, j, lane); sort_step<64, 1>(v, j, lane);
}
// The same two sorts on integer keys.  ord_key() maps a float to a uint32 whose unsigned order is the float order (-0 is first
// made +0, so equal floats have equal keys); a (value, index) pair becomes the 64-bit composite (ord_key(value) << 32) | ~index,
// whose unsigned order is "larger value first, equal values: smaller index first".  One exchange step is then two DPP moves, ONE
// 64-bit compare, an exclusive-or with a lane pattern that depends on the step only, and two selects - the float version's
// three compares and their and / or / select went through the scalar unit (800 of its 1 100 instructions per four queries), which
// the CU's four SIMDs share.  Rankings are identical (no NaNs: distances of finite points).
__device__ __forceinline__ uint32_t ord_key(float v) {
    const uint32_t u = __float_as_uint(v + 0.f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_val(uint32_t key) {
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key);
}
template <int K2, int S2>
__device__ __forceinline__ bool sort_flip(int lane) {   // lower lane of a descending block / upper lane of an ascending one keeps the first-ranked
    return (((lane & K2) == 0 || K2 == 64) != ((lane & S2) == 0));
}
template <int K2, int S2>
__device__ __forceinline__ void sort_step_key(uint32_t& key, int lane) {
    const uint32_t ok = lane_xor_u32<S2>(key, lane);
    const bool take = (ok > key) != sort_flip<K2, S2>(lane);
    key = take ? ok : key;
}
template <int K2, int S2>
__device__ __forceinline__ void sort_step_pair(uint32_t& hi, uint32_t& lo, int lane) {
    const uint32_t ohi = lane_xor_u32<S2>(hi, lane), olo = lane_xor_u32<S2>(lo, lane);
    const bool take = ((((uint64_t)ohi << 32) | olo) > (((uint64_t)hi << 32) | lo)) != sort_flip<K2, S2>(lane);
    hi = take ? ohi : hi;
    lo = take ? olo : lo;
}
// The same exchanges with the direction folded into the DATA: during stage K2 the lanes of the ascending blocks (lane & K2) hold the
// COMPLEMENT of their key, so that every block sorts the same way - the lower lane of a pair keeps the larger word, the upper lane the
// smaller - and the lane pattern of an exchange depends on its distance S2 alone: 6 patterns instead of 21 (with four networks side
// by side the 21 live scalar pairs no longer fitted: 168 v_writelane / v_readlane per four queries).  xform<K2>() moves the data from
// stage K2 / 2's form to stage K2's (one exclusive-or with a lane constant); stage 64 has no ascending blocks, so the network ends
// on the plain keys.
template <int K2>
__device__ __forceinline__ uint32_t stage_mask(int lane) {            // complemented lanes of stage K2 (none for K2 = 64 and before the first stage)
    return (K2 >= 64 || K2 < 2) ? 0u : (uint32_t)(0 - ((lane / K2) & 1));
}
template <int K2>
__device__ __forceinline__ uint32_t xform(int lane) { return stage_mask<K2 / 2>(lane) ^ stage_mask<K2>(lane); }   // (K2 = 2: from the plain keys)
template <int S2>
__device__ __forceinline__ void cx_key(uint32_t& key, int lane) {
    const uint32_t ok = lane_xor_u32<S2>(key, lane);
    const bool take = (ok > key) != ((lane & S2) != 0);
    key = take ? ok : key;
}
template <int S2>
__device__ __forceinline__ void cx_pair(uint32_t& hi, uint32_t& lo, int lane) {
    const uint32_t ohi = lane_xor_u32<S2>(hi, lane), olo = lane_xor_u32<S2>(lo, lane);
    const bool take = ((((uint64_t)ohi << 32) | olo) > (((uint64_t)hi << 32) | lo)) != ((lane & S2) != 0);
    hi = take ? ohi : hi;
    lo = take ? olo : lo;
}
#define SVNET_SORT_NETWORK(STEP, ...)                                                                                            \
    STEP<2, 1>(__VA_ARGS__);                                                                                                      \
    STEP<4, 2>(__VA_ARGS__); STEP<4, 1>(__VA_ARGS__);                                                                             \
    STEP<8, 4>(__VA_ARGS__); STEP<8, 2>(__VA_ARGS__); STEP<8, 1>(__VA_ARGS__);                                                    \
    STEP<16, 8>(__VA_ARGS__); STEP<16, 4>(__VA_ARGS__); STEP<16, 2>(__VA_ARGS__); STEP<16, 1>(__VA_ARGS__);                       \
    STEP<32, 16>(__VA_ARGS__); STEP<32, 8>(__VA_ARGS__); STEP<32, 4>(__VA_ARGS__); STEP<32, 2>(__VA_ARGS__); STEP<32, 1>(__VA_ARGS__); \
    STEP<64, 32>(__VA_ARGS__); STEP<64, 16>(__VA_ARGS__); STEP<64, 8>(__VA_ARGS__); STEP<64, 4>(__VA_ARGS__); STEP<64, 2>(__VA_ARGS__); \
    STEP<64, 1>(__VA_ARGS__)
// descending sort of one key per lane / of one (value key, ~index) pair per lane: lane 0 ends up with the first-ranked
__device__ __forceinline__ void wave_sort_keys(uint32_t& key, int lane) { SVNET_SORT_NETWORK(sort_step_key, key, lane); }
__device__ __forceinline__ void wave_sort_pairs(uint32_t& hi, uint32_t& lo, int lane) { SVNET_SORT_NETWORK(sort_step_pair, hi, lo, lane); }

// Selection: a wave holds the inner products of Q queries with ALL the cloud's candidates (candidate j = lane + 64 t in acc[q][t]),
// turns them into pd(i, j) and writes the k first-ranked ids of each query.  cand_v / cand_j: this wave's CAP slots of LDS.
template <int T>
__device__ __forceinline__ void knn_pd(float (&row)[T], const float (&xxj)[T], float xxi, int N, int lane) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int j = lane + 64 * t;
        const float inner = -2.0f * row[t];                    // exact
        const float t1 = __fsub_rn(-xxj[t], inner);            // fl(-xx[j] - inner)
        const float v = __fsub_rn(t1, xxi);                    // fl(.. - xx[i])
        row[t] = (j < N) ? v : -INFINITY;
    }
}
// one query: pd[t] = distance to candidate lane + 64 t; returns this lane's neighbour id (lane s: the s-th ranked)
template <int T, int CAP>
__device__ __forceinline__ int knn_select_one(float (&pd)[T], int k, int lane, float* cand_v, int* cand_j) {
// ---- top-k selection.  Threshold pass: the k-th largest of the 64 lane maxima is a lower bound of the k-th
    // largest distance, so every winner is >= it; those few candidates (typically < 2k) are compacted into LDS
    // with ballot prefix sums and sorted across the wave.  If more than 64 qualify (heavy ties) fall back to k
    // rounds of wave-wide arg-max.
    int mine = 0;
    float lm = pd[0];
#pragma unroll
    for (int t = 1; t < T; ++t) lm = fmaxf(lm, pd[t]);
    {
        uint32_t key = ord_key(lm);
        wave_sort_keys(key, lane);
        lm = ord_val((uint32_t)__shfl((int)key, k - 1, 64));   // threshold (wave-uniform): the k-th largest lane maximum
    }
    int count = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const bool in = pd[t] >= lm;
        const uint64_t m = __ballot(in);
        const int pos = count + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (in && pos < CAP) {
            cand_v[pos] = pd[t];
            cand_j[pos] = lane + 64 * t;
        }
        count += __popcll(m);
    }
    if (count <= CAP) {  // wave-uniform
        const float cv = (lane < count) ? cand_v[lane] : -INFINITY;
        const int cj = (lane < count) ? cand_j[lane] : 0x7fffffff;
        uint32_t hi = ord_key(cv), lo = ~(uint32_t)cj;
        wave_sort_pairs(hi, lo, lane);
        // More than 64 qualified (k close to 64: the k-th largest of 64 lane maxima is a weak bound - at N = 2048, k = 40 the
        // expected count is 62): every further chunk of 64 is sorted the same way and merged in - max(A[i], B[63 - i]) of two
        // descending runs holds the 64 first-ranked of their union as a bitonic sequence, which the last stage of the network sorts.
        for (int c0 = 64; c0 < count; c0 += 64) {
            const float cv2 = (c0 + lane < count) ? cand_v[c0 + lane] : -INFINITY;
            const int cj2 = (c0 + lane < count) ? cand_j[c0 + lane] : 0x7fffffff;
            uint32_t hi2 = ord_key(cv2), lo2 = ~(uint32_t)cj2;
            wave_sort_pairs(hi2, lo2, lane);
            const uint32_t rh = (uint32_t)__shfl((int)hi2, 63 - lane, 64), rl = (uint32_t)__shfl((int)lo2, 63 - lane, 64);
            const bool take = ((((uint64_t)rh << 32) | rl) > (((uint64_t)hi << 32) | lo));
            hi = take ? rh : hi;
            lo = take ? rl : lo;
            sort_step_pair<64, 32>(hi, lo, lane); sort_step_pair<64, 16>(hi, lo, lane); sort_step_pair<64, 8>(hi, lo, lane);
            sort_step_pair<64, 4>(hi, lo, lane); sort_step_pair<64, 2>(hi, lo, lane); sort_step_pair<64, 1>(hi, lo, lane);
        }
        mine = (int)~lo;
    } else {
        for (int s = 0; s < k; ++s) {
            float bv = pd[0];
            int bt = 0;
#pragma unroll
            for (int t = 1; t < T; ++t) {
                const bool g = pd[t] > bv;  // strict: first (lowest j) wins inside a lane
                bv = g ? pd[t] : bv;
                bt = g ? t : bt;
            }
            int bj = lane + 64 * bt;
            wave_argmax(bv, bj);
            if (lane == s) mine = bj;
#pragma unroll
            for (int t = 0; t < T; ++t) pd[t] = (bj == lane + 64 * t) ? -INFINITY : pd[t];
        }
    }
    return mine;
}
template <int T, int Q, int CAP>
__device__ __forceinline__ void knn_select(float (&acc)[Q][T], const float* __restrict__ xxb, int N, int k, int lane, float* cand_v, int* cand_j,
                                           const int (&qid)[Q], int64_t* __restrict__ idx_cloud) {
    float xxj[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int j = lane + 64 * t;
        xxj[t] = (j < N) ? xxb[j] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (qid[q] >= N) continue;  // wave-uniform
        knn_pd<T>(acc[q], xxj, xxb[qid[q]], N, lane);
        const int mine = knn_select_one<T, CAP>(acc[q], k, lane, cand_v, cand_j);
        // (NaN distances compare false everywhere and can leave slots unfilled: never hand an out-of-range id to the gathers)
        if (lane < k) idx_cloud[(size_t)qid[q] * k + lane] = ((unsigned)mine < (unsigned)N) ? mine : qid[q];
    }
}

// Inclusive prefix sum of one integer per lane across the wave: four shifts inside the 16-lane rows, then the row totals are carried
// over with the two row broadcasts (DPP modifiers of the adds; lanes without a source add 0).
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// Four queries at once.  The steps are those of knn_select_one; what changes is how they are issued:
//  * the four queries' sorting networks run side by side, stage by stage - a network is a chain of 21 dependent exchanges, each a DPP move
//    that has to wait two cycles for the select before it (481 s_nop in the one-at-a-time form) - four independent chains fill those slots;
//  * the candidates >= the threshold are compacted lane-locally: a lane counts its own, ONE prefix sum over the lanes gives it its first
//    slot, and it writes its candidates there one after the other (a select of the address, not an exec-mask branch, keeps the lanes
//    that have none out: they write to a slot of their own behind the list) - the ballot / mbcnt / s_bcnt1 / branch per register of the
//    old form was 12 instructions per candidate register, half of them scalar;
//  * a query with more than 64 candidates (heavy ties, k near 64) sends all four through knn_select_one.
// scratch: 4 x 2 x SVNET_KNN_SLOTS words of this wave's LDS.
constexpr int SVNET_KNN_SLOTS = 192;        // 64 list slots + 64 lanes' spare slots (+ 64: knn_select_one's 128)
template <int T>
__device__ __forceinline__ void knn_select4(float (&acc)[4][T], const float* __restrict__ xxb, int N, int k, int lane, float* scratch,
                                            const int (&qid)[4], int64_t* __restrict__ idx_cloud) {
    float xxj[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int j = lane + 64 * t;
        xxj[t] = (j < N) ? xxb[j] : 0.f;
    }
    uint32_t key[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        knn_pd<T>(acc[q], xxj, xxb[min(qid[q], N - 1)], N, lane);
        float lm = acc[q][0];
#pragma unroll
        for (int t = 1; t < T; ++t) lm = fmaxf(lm, acc[q][t]);
        key[q] = ord_key(lm);
    }
#define SVNET_X4_KEY(K2) do { const uint32_t m_ = xform<K2>(lane); key[0] ^= m_; key[1] ^= m_; key[2] ^= m_; key[3] ^= m_; } while (0)
#define SVNET_CX4_KEY(S2) do { cx_key<S2>(key[0], lane); cx_key<S2>(key[1], lane); cx_key<S2>(key[2], lane); cx_key<S2>(key[3], lane); } while (0)
    SVNET_X4_KEY(2); SVNET_CX4_KEY(1);
    SVNET_X4_KEY(4); SVNET_CX4_KEY(2); SVNET_CX4_KEY(1);
    SVNET_X4_KEY(8); SVNET_CX4_KEY(4); SVNET_CX4_KEY(2); SVNET_CX4_KEY(1);
    SVNET_X4_KEY(16); SVNET_CX4_KEY(8); SVNET_CX4_KEY(4); SVNET_CX4_KEY(2); SVNET_CX4_KEY(1);
    SVNET_X4_KEY(32); SVNET_CX4_KEY(16); SVNET_CX4_KEY(8); SVNET_CX4_KEY(4); SVNET_CX4_KEY(2); SVNET_CX4_KEY(1);
    SVNET_X4_KEY(64); SVNET_CX4_KEY(32); SVNET_CX4_KEY(16); SVNET_CX4_KEY(8); SVNET_CX4_KEY(4); SVNET_CX4_KEY(2); SVNET_CX4_KEY(1);
#undef SVNET_X4_KEY
#undef SVNET_CX4_KEY
    float thr[4];
    int pos[4], total[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        thr[q] = ord_val((uint32_t)__builtin_amdgcn_readlane((int)key[q], k - 1));     // the k-th largest lane maximum (wave-uniform)
        int cnt = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) cnt += (acc[q][t] >= thr[q]) ? 1 : 0;
        const int incl = wave_incl_scan(cnt);
        total[q] = __builtin_amdgcn_readlane(incl, 63);
        pos[q] = incl - cnt;
    }
    const int most = max(max(total[0], total[1]), max(total[2], total[3]));
    int mine[4];
    if (most <= 64) {   // (wave-uniform)
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* sv = scratch + q * 2 * SVNET_KNN_SLOTS;
            int* sj = reinterpret_cast<int*>(sv + SVNET_KNN_SLOTS);
            int at = pos[q];
            float th = thr[q];
            asm volatile("" : "+s"(th));                                 // (or the 64 compares of the counting pass are kept alive as scalar pairs, spilled lane by lane)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const bool in = acc[q][t] >= th;
                const int slot = in ? at : 64 + lane;
                sv[slot] = acc[q][t];
                sj[slot] = lane + 64 * t;
                at += in ? 1 : 0;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* sv = scratch + q * 2 * SVNET_KNN_SLOTS;
            const int* sj = reinterpret_cast<const int*>(sv + SVNET_KNN_SLOTS);
            const float cv = sv[lane];
            const int cj = sj[lane];
            hi[q] = ord_key((lane < total[q]) ? cv : -INFINITY);
            lo[q] = ~(uint32_t)((lane < total[q]) ? cj : 0x7fffffff);
        }
#define SVNET_X4_PAIR(K2) do { const uint32_t m_ = xform<K2>(lane); _Pragma("unroll") for (int q = 0; q < 4; ++q) { hi[q] ^= m_; lo[q] ^= m_; } } while (0)
#define SVNET_CX4_PAIR(S2) do { cx_pair<S2>(hi[0], lo[0], lane); cx_pair<S2>(hi[1], lo[1], lane); cx_pair<S2>(hi[2], lo[2], lane); cx_pair<S2>(hi[3], lo[3], lane); } while (0)
        SVNET_X4_PAIR(2); SVNET_CX4_PAIR(1);
        SVNET_X4_PAIR(4); SVNET_CX4_PAIR(2); SVNET_CX4_PAIR(1);
        SVNET_X4_PAIR(8); SVNET_CX4_PAIR(4); SVNET_CX4_PAIR(2); SVNET_CX4_PAIR(1);
        SVNET_X4_PAIR(16); SVNET_CX4_PAIR(8); SVNET_CX4_PAIR(4); SVNET_CX4_PAIR(2); SVNET_CX4_PAIR(1);
        SVNET_X4_PAIR(32); SVNET_CX4_PAIR(16); SVNET_CX4_PAIR(8); SVNET_CX4_PAIR(4); SVNET_CX4_PAIR(2); SVNET_CX4_PAIR(1);
        SVNET_X4_PAIR(64); SVNET_CX4_PAIR(32); SVNET_CX4_PAIR(16); SVNET_CX4_PAIR(8); SVNET_CX4_PAIR(4); SVNET_CX4_PAIR(2); SVNET_CX4_PAIR(1);
#undef SVNET_X4_PAIR
#undef SVNET_CX4_PAIR
#pragma unroll
        for (int q = 0; q < 4; ++q) mine[q] = (int)~lo[q];
    } else {
        // (rare: ONE copy of the one-at-a-time code, walked by a loop that is not unrolled, and fed through LDS - it takes a query's
        //  distances from this wave's 4 KB, not from the accumulator registers: four inlined copies reading the registers cost the
        //  common path 6 - 9 us per call in spilled registers)
        mine[0] = mine[1] = mine[2] = mine[3] = 0;
        float* row = scratch + 2 * SVNET_KNN_SLOTS;                      // [64 T] behind knn_select_one's 128 + 128 slots (at T = 16: 6 KB of the wave's 6)
#pragma unroll 1
        for (int q = 0; q < (SVNET_KNN_NOSLOW ? 0 : 4); ++q) {
            if (q == 0) { _Pragma("unroll") for (int t = 0; t < T; ++t) row[64 * t + lane] = acc[0][t]; }
            else if (q == 1) { _Pragma("unroll") for (int t = 0; t < T; ++t) row[64 * t + lane] = acc[1][t]; }
            else if (q == 2) { _Pragma("unroll") for (int t = 0; t < T; ++t) row[64 * t + lane] = acc[2][t]; }
            else { _Pragma("unroll") for (int t = 0; t < T; ++t) row[64 * t + lane] = acc[3][t]; }
            float pd[T];
#pragma unroll
            for (int t = 0; t < T; ++t) pd[t] = row[64 * t + lane];
            const int m = knn_select_one<T, 128>(pd, k, lane, scratch, reinterpret_cast<int*>(scratch + SVNET_KNN_SLOTS));
            mine[0] = q == 0 ? m : mine[0]; mine[1] = q == 1 ? m : mine[1]; mine[2] = q == 2 ? m : mine[2]; mine[3] = q == 3 ? m : mine[3];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)   // (NaN distances compare false everywhere and can leave slots unfilled: never hand an out-of-range id to the gathers)
        if (qid[q] < N && lane < k) idx_cloud[(size_t)qid[q] * k + lane] = ((unsigned)mine[q] < (unsigned)N) ? mine[q] : qid[q];
}

// T candidates per lane (64*T >= N), Q query rows per wave, 4 waves per workgroup.
// STAGE: the four waves of a workgroup share the candidate rows through LDS (CC channels at a time, N % 4 == 0): every wave
// needs the whole [C, N] table of its cloud, so without sharing the L2 -> CU traffic is 4x what the arithmetic can hide.
constexpr int KNN_CC = 8;
// SPLIT (T = 16, Q = 4, N % 16 == 0): in the distance loop the four waves split the CANDIDATES (256 each) and every wave carries
// all 16 queries of the workgroup, so a channel costs a wave 1 KB of LDS reads instead of 4 KB for the same 32 packed FMAs (the
// loop was co-limited by LDS bytes); the finished inner products are then handed over through LDS to the layout the selection
// works on (wave = 4 queries x all candidates).  The fmaf chain of every (query, candidate) pair is unchanged.
// WPB = waves per workgroup (8 with SPLIT: 32 queries share one pass over the cloud's [C, N] table - every workgroup stages the
// WHOLE table through LDS, 520 KB at C = 127, so the L2 -> LDS traffic of a call is (N / queries per workgroup) tables per cloud).
// DIRECT (with SPLIT): no staging of the candidates at all - a wave needs only ITS 64*T/WPB candidates of a channel (four coalesced
// 256-byte loads), so it takes them from L2 into a four-channel register ring; only the workgroup's 16 query rows go through LDS
// (once).  No barrier inside the channel loop.
// MF (with SPLIT and DIRECT, 16 queries per workgroup): the inner products on v_mfma_f32_16x16x4_f32 - 16 queries x 16 candidates x 4
// channels per instruction, whose result is bit for bit the k-ordered fmaf chain the contract asks for (one rounding per product,
// accumulator = C input; MI355X_MICROARCH.md / cdna_hip_programming.md §3 "FP32-input MFMA").  The f32 matrix rate equals the
// f32 vector rate, so the loop itself gains little - but it runs on the MATRIX pipe, one operand VGPR per lane and instruction,
// and leaves the vector ALUs to the waves that are in their selection phase (half of this kernel's time).
template <int T, int Q, bool STAGE, bool SPLIT = false, int WPB = 4, bool DIRECT = false, bool MF = false>
// (T = 32, N <= 2048: 128 accumulators per lane - without the bound the allocator takes 268 registers, one wave per SIMD; with it 240, two)
__global__ __launch_bounds__(64 * WPB, (T == 32 && SPLIT && !MF) ? 2 : 1) void knn_main_kernel(const float* __restrict__ xT, const float* __restrict__ xx,
                                                            int N, int C, int k, int64_t* __restrict__ idx_out, int xcd_blocks_per_cloud) {
    constexpr int CAP = T >= 32 ? 256 : (T >= 16 ? 128 : 64);         // candidate slots per wave (see the selection below)
    __shared__ float cand_v[WPB * CAP];
    __shared__ int cand_j[WPB * CAP];
    extern __shared__ __attribute__((aligned(16))) float rows[];      // STAGE: [KNN_CC][64 * T]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // XCD-aware order (workgroups w and w+8 share an XCD): with B % 8 == 0 the launch is one-dimensional and XCD x walks the clouds
    // x, x+8, ... one after the other, so the cloud's [C, N] table - which EVERY workgroup of the cloud reads in full - is served
    // by one XCD's L2 instead of being resident in all eight
    int bx = blockIdx.x, b = blockIdx.y;
    if (gridDim.y == 1 && xcd_blocks_per_cloud > 0) {
        const int w = blockIdx.x, per = xcd_blocks_per_cloud;
        b = (w / (per * 8)) * 8 + (w & 7);
        bx = (w >> 3) % per;
    }
    const int q0 = (bx * WPB + wave) * Q;
    if (!STAGE && q0 >= N) return;  // wave-uniform (no workgroup barriers in the unstaged variant: the LDS slices are per wave)

    const float* __restrict__ xb = xT + (size_t)b * (MF ? ((C + 3) & ~3) : C) * N;     // (MF: the interleaved table has 4*ceil(C/4) channels)
    const float* __restrict__ xxb = xx + (size_t)b * N;

    float acc[Q][T];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int t = 0; t < T; ++t) acc[q][t] = 0.f;

    int qi[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) qi[q] = min(q0 + q, N - 1);
    float (&accs)[Q][T] = acc;                                       // SPLIT: the same 64 registers, indexed [q % 4][(q / 4) * T/4 + t]

    if (STAGE || (SPLIT && DIRECT)) {
        // chunk i+1 travels global -> registers while chunk i is consumed from LDS; the FMA chain over c keeps its order
        constexpr int NP = 64 * T;
        constexpr int NTH = 64 * WPB;
        constexpr int F4 = (KNN_CC * NP / 4 / NTH) > 0 ? (KNN_CC * NP / 4 / NTH) : 1;   // float4 per thread per chunk (STAGE needs T >= 4)
        float4 stg[F4];
        const int n4 = N >> 2;
#define SVNET_KNN_FETCH(C0)                                                                         \
    do {                                                                                            \
        _Pragma("unroll") for (int u = 0; u < F4; ++u) {                                            \
            const int e_ = u * NTH + threadIdx.x;                                                   \
            const int rw_ = e_ / (NP / 4), c4_ = e_ - rw_ * (NP / 4);                               \
            stg[u] = ((C0) + rw_ < C && c4_ < n4) ? *reinterpret_cast<const float4*>(xb + (size_t)((C0) + rw_) * N + 4 * c4_) \
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);               \
        }                                                                                           \
    } while (0)
        if constexpr (SPLIT && DIRECT) {
            constexpr int TS = T / WPB > 0 ? T / WPB : 1, QB = WPB * Q;
            const int qb0 = bx * QB;
            const int C4 = (C + 3) & ~3;
            for (int e = threadIdx.x; e < C4 * QB; e += NTH) {          // rows[c * QB + i] = x[c][qb0 + i]; channels past C are zeros
                const int c = e / QB, i = e - c * QB;
                if (MF) rows[e] = xb[((size_t)(c >> 2) * N + qb0 + i) * 4 + (c & 3)];        // (interleaved table, zero padded)
                else rows[e] = c < C ? xb[(size_t)c * N + qb0 + i] : 0.f;
            }
            if constexpr (MF) {
                // tile tt of this wave: candidates 64*TS*wave + 16*tt .. +15; lane l = (k = l >> 4, j = l & 15) holds B[k][j] = x[4s + k][candidate j]
                // and A[i = l & 15][k] = x[4s + k][query qb0 + i] = rows[64 s + l]  (the query rows were laid out [c][16])
                typedef __attribute__((ext_vector_type(4))) float f32x4;
                constexpr int NTL = 4 * TS;                                 // 16-candidate tiles per wave
                static_assert(QB == 16, "the MFMA form carries 16 queries per workgroup");
                f32x4 dacc[NTL];
#pragma unroll
                for (int tt = 0; tt < NTL; ++tt) dacc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int kl = lane >> 4, jl = lane & 15;
                int coff[NTL];
#pragma unroll
                for (int tt = 0; tt < NTL; ++tt) coff[tt] = 4 * min(64 * TS * wave + 16 * tt + jl, N - 1);
                constexpr int RING = NTL <= 16 ? 2 : 1;                    // k-steps of candidates in flight
                float bn[RING][NTL];
// (interleaved table: the 16 candidates x 4 channels of a tile are one 256-byte run, the wave's tiles of a k-step 1 KB x TS)
#define SVNET_KNN_LOADB(SLOT, S4)                                                                   \
    do {                                                                                            \
        const float* r_ = xb + (size_t)(S4) * N * 4 + kl;                                           \
        _Pragma("unroll") for (int tt = 0; tt < NTL; ++tt) bn[SLOT][tt] = r_[coff[tt]];             \
    } while (0)
                const int nks = C4 >> 2;
#pragma unroll
                for (int u = 0; u < RING; ++u) SVNET_KNN_LOADB(u, min(u, nks - 1));
                __syncthreads();
                for (int s0 = 0; s0 < nks; s0 += RING) {
#pragma unroll
                    for (int u = 0; u < RING; ++u) {
                        const int s4 = s0 + u;
                        if (s4 >= nks) break;                              // (wave-uniform)
                        const float av = rows[64 * s4 + lane];             // zeros past C: those products add nothing
                        float bv[NTL];
#pragma unroll
                        for (int tt = 0; tt < NTL; ++tt) bv[tt] = bn[u][tt];
                        SVNET_KNN_LOADB(u, min(s4 + RING, nks - 1));       // unconditional, clamped: RING k-steps ahead
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int tt = 0; tt < NTL; ++tt) dacc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[tt], dacc[tt], 0, 0, 0);
                    }
                }
#undef SVNET_KNN_LOADB
                // D reg g of tile tt in lane l: query 4 (l >> 4) + g, candidate 64*TS*wave + 16 tt + (l & 15) -> the flat accumulator order
                // of the vector form is not used here: the hand-over below takes dacc directly
                constexpr int NP_ = 64 * T;
                constexpr int PQ_ = (8192 / NP_) > 0 ? (8192 / NP_) : 1;
                float mine_[Q][T];
#pragma unroll
                for (int pass = 0; pass < QB / PQ_; ++pass) {
                    __syncthreads();
                    // queries of this pass: pass * PQ_ .. + PQ_ - 1; this lane holds queries 4 kl .. 4 kl + 3
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int qq = 4 * kl + g;
                        if (qq / PQ_ == pass) {
#pragma unroll
                            for (int tt = 0; tt < NTL; ++tt)
                                rows[(qq - pass * PQ_) * NP_ + 64 * TS * wave + 16 * tt + jl] = dacc[tt][g];
                        }
                    }
                    __syncthreads();
                    if ((wave * Q) / PQ_ == pass) {
#pragma unroll
                        for (int q = 0; q < Q; ++q)
#pragma unroll
                            for (int t = 0; t < T; ++t) mine_[q][t] = rows[((wave * Q) % PQ_ + q) * NP_ + 64 * t + lane];
                    }
                }
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[q][t] = mine_[q][t];
            } else {
            // lane l carries the candidates 64*TS*wave + 256*(t/4) + 4*l + (t%4) of its wave's slice: four consecutive floats of a channel
            // per load instruction (the L2 -> CU path was the loop's limit with one float per lane and load: 7.8 TB/s at every width),
            // pairs (t, t+1) in adjacent registers as the packed FMAs want them.  The hand-over below restores the order lane + 64*t.
            static_assert(TS % 4 == 0, "four candidates per lane and load");
            constexpr int TV = TS / 4;
            int joff[TV];
    #pragma unroll
                for (int v = 0; v < TV; ++v) joff[v] = min(64 * TS * wave + 256 * v + 4 * lane, N - 4);     // (N % 16 == 0)
                float4 cn[4][TV];
    #define SVNET_KNN_LOADC(SLOT, CH)                                                                   \
        do {                                                                                            \
            const float* r_ = xb + (size_t)(SVNET_KNN_ABL == 3 ? 0 : min((CH), C - 1)) * N;               \
            _Pragma("unroll") for (int v = 0; v < TV; ++v) cn[SLOT][v] = *reinterpret_cast<const float4*>(r_ + joff[v]);   \
        } while (0)
                SVNET_KNN_LOADC(0, 0); SVNET_KNN_LOADC(1, 1); SVNET_KNN_LOADC(2, 2);
                __syncthreads();
                for (int c4 = 0; c4 < C4; c4 += 4) {
    #pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = c4 + u;
                        SVNET_KNN_LOADC((u + 3) & 3, c + 3);               // unconditional, clamped: three channels ahead
                        __builtin_amdgcn_sched_barrier(0);
                        float qv[QB];
    #pragma unroll
                        for (int i = 0; i < QB / 4; ++i) {
                            const float4 q4 = *reinterpret_cast<const float4*>(rows + c * QB + 4 * i);   // broadcast read (zeros past C)
                            qv[4 * i] = q4.x; qv[4 * i + 1] = q4.y; qv[4 * i + 2] = q4.z; qv[4 * i + 3] = q4.w;
                        }
                        float cv[TS];
    #pragma unroll
                        for (int v = 0; v < TV; ++v) { cv[4 * v] = cn[u][v].x; cv[4 * v + 1] = cn[u][v].y; cv[4 * v + 2] = cn[u][v].z; cv[4 * v + 3] = cn[u][v].w; }
    #pragma unroll
                        for (int q = 0; q < QB; ++q)
    #pragma unroll
                            for (int t = 0; t < TS; ++t)
                                accs[(q * TS + t) / T][(q * TS + t) % T] = __builtin_fmaf(qv[q], cv[t], accs[(q * TS + t) / T][(q * TS + t) % T]);
                    }
                }
#undef SVNET_KNN_LOADC
            }
        } else {
        SVNET_KNN_FETCH(0);
        for (int c0 = 0; c0 < C; c0 += KNN_CC) {
            __syncthreads();                                         // the previous chunk has been consumed
#pragma unroll
            for (int u = 0; u < F4; ++u) *reinterpret_cast<float4*>(&rows[4 * (u * NTH + threadIdx.x)]) = stg[u];
            __syncthreads();
            if (c0 + KNN_CC < C) SVNET_KNN_FETCH(c0 + KNN_CC);
            const int cc_end = min(KNN_CC, C - c0);
            if (SPLIT) {
                constexpr int TS = T / WPB > 0 ? T / WPB : 1, QB = WPB * Q;   // candidates per lane, queries per workgroup
                const int qb0 = bx * QB;                              // N % QB == 0: the workgroup's queries all exist, 16-byte aligned
                for (int cc = 0; cc < cc_end; ++cc) {
                    const float* row = rows + cc * NP;
                    float cand[TS], qv[QB];
#pragma unroll
                    for (int t = 0; t < TS; ++t) cand[t] = row[64 * TS * wave + lane + 64 * t];
#pragma unroll
                    for (int i = 0; i < QB / 4; ++i) {
                        const float4 q4 = *reinterpret_cast<const float4*>(row + qb0 + 4 * i);   // broadcast read
                        qv[4 * i] = q4.x; qv[4 * i + 1] = q4.y; qv[4 * i + 2] = q4.z; qv[4 * i + 3] = q4.w;
                    }
#pragma unroll
                    for (int q = 0; q < QB; ++q)
#pragma unroll
                        for (int t = 0; t < TS; ++t) accs[(q * TS + t) / T][(q * TS + t) % T] = __builtin_fmaf(qv[q], cand[t], accs[(q * TS + t) / T][(q * TS + t) % T]);
                }
                continue;
            }
            for (int cc = 0; cc < cc_end; ++cc) {
                const float* row = rows + cc * NP;
                float cand[T], qv[Q];
#pragma unroll
                for (int t = 0; t < T; ++t) cand[t] = row[lane + 64 * t];
#pragma unroll
                for (int q = 0; q < Q; ++q) qv[q] = row[qi[q]];
#pragma unroll
                for (int q = 0; q < Q; ++q)
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[q][t] = __builtin_fmaf(qv[q], cand[t], acc[q][t]);
            }
        }
        }
#undef SVNET_KNN_FETCH
        if (SPLIT && !MF) {
            // accs (the registers of acc, flat index q * TS + t) = inner product of query qb0 + q with candidate 64 * TS * wave + 64 * t + lane.
            // Hand-over in passes of PQ queries (PQ x 64T floats = 32 KB of LDS): the waves that own them read.
            constexpr int TS = T / WPB > 0 ? T / WPB : 1, QB = WPB * Q;
            constexpr int PQ = (8192 / NP) > 0 ? (8192 / NP) : 1;      // 8 at N <= 1024, 4 at N <= 2048
            float mine[Q][T];
#pragma unroll
            for (int pass = 0; pass < QB / PQ; ++pass) {
                __syncthreads();                                     // rows[] is free (last chunk consumed / previous pass read)
#pragma unroll
                for (int q = 0; q < PQ; ++q) {
                    const int qq = pass * PQ + q;
#pragma unroll
                    for (int t = 0; t < TS; ++t)
                        rows[q * NP + 64 * TS * wave + (DIRECT ? 256 * (t >> 2) + 4 * lane + (t & 3) : 64 * t + lane)] = accs[(qq * TS + t) / T][(qq * TS + t) % T];
                }
                __syncthreads();
                if ((wave * Q) / PQ == pass) {                       // the waves whose Q queries lie in this pass
#pragma unroll
                    for (int q = 0; q < Q; ++q)
#pragma unroll
                        for (int t = 0; t < T; ++t) mine[q][t] = rows[((wave * Q) % PQ + q) * NP + 64 * t + lane];
                }
            }
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int t = 0; t < T; ++t) acc[q][t] = mine[q][t];
        }
        if (q0 >= N) return;                                         // idle waves only helped with the staging
    } else {
    // channel c+1's candidate row and query values are requested before channel c's FMAs (the chain over c is what fixes
    // the rounding, so the loop cannot be reordered, only overlapped)
    int jc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) jc[t] = min(lane + 64 * t, N - 1);    // clamped: lanes past N are overwritten with -inf below
    float cn[T], qn[Q];
#pragma unroll
    for (int t = 0; t < T; ++t) cn[t] = xb[jc[t]];
#pragma unroll
    for (int q = 0; q < Q; ++q) qn[q] = xb[qi[q]];                     // wave-uniform address -> scalar load
    for (int c = 0; c < C; ++c) {
        float cand[T], qv[Q];
#pragma unroll
        for (int t = 0; t < T; ++t) cand[t] = cn[t];
#pragma unroll
        for (int q = 0; q < Q; ++q) qv[q] = qn[q];
        if (c + 1 < C) {
            const float* __restrict__ row = xb + (size_t)(c + 1) * N;
#pragma unroll
            for (int t = 0; t < T; ++t) cn[t] = row[jc[t]];
#pragma unroll
            for (int q = 0; q < Q; ++q) qn[q] = row[qi[q]];
        }
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[q][t] = __builtin_fmaf(qv[q], cand[t], acc[q][t]);
    }
    }

#if SVNET_KNN_ABL == 1 || SVNET_KNN_ABL == 3   // (3: every channel's candidates from channel 0's row - the loop without its L2 traffic)
  // diagnostic build (tools/knn_lab.py): the distance phase alone - no selection, the sums keep the accumulators alive
    {
        float sm = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int t = 0; t < T; ++t) sm += acc[q][t];
        if (lane < k) idx_out[((size_t)b * N + q0) * k + lane] = (int64_t)(sm > 0.f);
        return;
    }
#endif
    int qid[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) qid[q] = q0 + q;
    knn_select<T, Q, CAP>(acc, xxb, N, k, lane, cand_v + wave * CAP, cand_j + wave * CAP, qid, idx_out + (size_t)b * N * k);
}

// ---- Matrix-core form for 512 < N <= 1024 (N % 16 == 0): 32 queries per 8-wave workgroup.
// Where the time of the vector form went (tools/knn_lab.py, B = 32, N = 1024, k = 20): its distance loop runs at 70 % of what the
// packed-FMA pipe sustains (119 TFLOP/s at the clock the chip holds under this load, not the 157 of the data sheet) whether its
// candidates come from L2 or from L1, so it cannot get shorter on the vector ALUs - and the selection needs those same ALUs.  Round 3's
// matrix-core form was slower for a different reason: one query tile per wave means one operand register from L2 per MFMA (4 TB/s).
// Here the table goes through LDS in chunks of KC channels (coalesced 16-byte loads, each byte fetched once per 32 queries), a wave
// multiplies 2 query tiles x NTL candidate tiles (10 operand reads from LDS per 16 MFMAs), and the finished inner products are handed
// over through the same LDS to the layout the selection works on.  v_mfma_f32_16x16x4_f32 adds its four products in channel order with
// one rounding each: the fmaf chain of the contract, bit for bit.  Two workgroups per CU: one's MFMA phase beside the other's selection.
template <int T>
__global__ __launch_bounds__(512, 4) void knn_mf8_kernel(const float* __restrict__ xT, const float* __restrict__ xx, int N, int C, int k,
                                                         int64_t* __restrict__ idx_out, int xcd_blocks_per_cloud) {
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    constexpr int NP = 64 * T;                 // candidate columns of a staged row (>= N)
    constexpr int LDS_S = NP + 16;             // staged row stride: the four channel rows of a k-step start 16 banks apart
    constexpr int LDS_H = NP + 4;              // hand-over row stride: the four query rows a tile's lanes write lie 16 banks apart
    constexpr int KC = 4, NB = 4;              // channels per staged k-step of the 16x16x4 product, buffers in the ring
    constexpr int WPB = 8, QB = 32, Q = 4;
    constexpr int CW = NP / WPB;               // candidates per wave in the distance phase
    constexpr int NTL = CW / 16;               // 16-candidate tiles per wave
    static_assert(16 * LDS_H <= NB * KC * LDS_S && WPB * 8 * SVNET_KNN_SLOTS <= NB * KC * LDS_S && 2 * SVNET_KNN_SLOTS + 64 * T <= 8 * SVNET_KNN_SLOTS, "the hand-over and the selection's lists fit in the staging buffers");
    extern __shared__ __attribute__((aligned(16))) float rows[];          // [NB][KC][LDS_S]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int bx = blockIdx.x, b = blockIdx.y;
    if (gridDim.y == 1 && xcd_blocks_per_cloud > 0) {                     // XCD-aware cloud order, as in knn_main_kernel
        const int w = blockIdx.x, per = xcd_blocks_per_cloud;
        b = (w / (per * 8)) * 8 + (w & 7);
        bx = (w >> 3) % per;
    }
    const int qb0 = bx * QB;
    const float* __restrict__ xxb = xx + (size_t)b * N;
    const int kl = lane >> 4, jl = lane & 15;

    // Staging: LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes = 256 consecutive candidates of one channel row, no registers) into
    // a ring of NB buffers of one k-step (4 channel rows) each.  The table's clouds are C8 = 8 ceil(C / 8) rows apart and the rows past C
    // hold zeros, so every k-step is whole.  At the top of step s: k-steps <= s + 1 have landed and been seen by every wave (the wait and
    // barrier that ended step s - 1), k-step s + 2 is in flight, k-step s's operands are in registers.  Step s then requests k-step s + 3
    // (into the buffer whose operands were read during step s - 2), reads the operands of s + 1 and issues the products of s - nothing
    // in it waits for a round trip.  Every wave issues DMA_PER instructions per k-step whatever N is (columns past N: a clamped source),
    // so that the counted wait means the same in all of them; a run that starts before N and ends past it reads on into the next row
    // (the table is followed by the ||x||^2 array: always mapped), and what lands past column N only reaches distances the selection masks.
    const int C8 = (C + 7) / 8 * 8;
    const float* __restrict__ xb8 = xT + (size_t)b * C8 * N;
    typedef __attribute__((address_space(3))) float lds_f32;
    typedef const __attribute__((address_space(1))) float glb_f32;
    constexpr int DMA_PER = KC * (NP / 256) / WPB;                       // 2 at N <= 1024
    static_assert(DMA_PER * WPB == KC * (NP / 256) && DMA_PER == 2, "the counted waits below assume two DMA instructions per wave and k-step");
    const int nks = (C + 3) >> 2;
#define SVNET_KNN_DMA4(S)                                                                           \
    do {                                                                                            \
        const int s_ = min((S), nks - 1);                                                           \
        float* buf_ = rows + ((S) & (NB - 1)) * (KC * LDS_S);                                       \
        _Pragma("unroll") for (int u = 0; u < DMA_PER; ++u) {                                       \
            const int e_ = wave * DMA_PER + u, rw_ = e_ / (NP / 256), i_ = e_ - rw_ * (NP / 256);  \
            __builtin_amdgcn_global_load_lds((glb_f32*)(xb8 + (size_t)(4 * s_ + rw_) * N + (256 * i_ < N ? 256 * i_ : N - 256) + 4 * lane),     \
                                             (lds_f32*)(buf_ + rw_ * LDS_S + 256 * i_), 16, 0, 0);                                  \
        }                                                                                           \
    } while (0)
    // lane l = (kl = l >> 4, jl = l & 15): A[i = jl][kl] = x[4 s + kl][qb0 + 16 a + jl], B[kl][j = jl] = x[4 s + kl][candidate]
#define SVNET_KNN_OPERANDS(S, AV, BV)                                                               \
    do {                                                                                            \
        const float* rk_ = rows + ((S) & (NB - 1)) * (KC * LDS_S) + kl * LDS_S + jl;                \
        AV[0] = rk_[qb0]; AV[1] = rk_[qb0 + 16];                                                    \
        _Pragma("unroll") for (int tt = 0; tt < NTL; ++tt) BV[tt] = rk_[CW * wave + 16 * tt];       \
    } while (0)
#define SVNET_KNN_PRODUCTS_HALF(AV, BV, H)                                                          \
    do {                                                                                            \
        _Pragma("unroll") for (int tt = (H) * (NTL / 2); tt < ((H) + 1) * (NTL / 2); ++tt) {        \
            dacc[0][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[0], BV[tt], dacc[0][tt], 0, 0, 0); \
            dacc[1][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[1], BV[tt], dacc[1][tt], 0, 0, 0); \
        }                                                                                           \
    } while (0)
    // end of a step: all but the newest k-step's DMA have landed (this wave's share; the barrier makes it everyone's); the wave's own LDS
    // reads are waited for by the compiler where their values are first used (the operand copies behind the barrier)
#define SVNET_KNN_STEP_END()                                                                        \
    do {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                            \
        __builtin_amdgcn_s_barrier();                                                               \
        asm volatile("" ::: "memory");                                                              \
    } while (0)

    f32x4 dacc[2][NTL];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int tt = 0; tt < NTL; ++tt) dacc[a][tt] = f32x4{0.f, 0.f, 0.f, 0.f};

    float av[2], bv[NTL], avn[2], bvn[NTL];
#if SVNET_KNN_ABL == 4   // diagnostic build (tools/knn_floor.sh): NO distance phase - the hand-over and the selection on zero inner products
    const int nks_run = 0;
    (void)nks;
#else
    const int nks_run = nks;
#endif
    SVNET_KNN_DMA4(0); SVNET_KNN_DMA4(1); SVNET_KNN_DMA4(2);
    SVNET_KNN_STEP_END();                                                // k-steps 0 and 1 have landed
    SVNET_KNN_OPERANDS(0, av, bv);
    for (int s = 0; s < nks_run; ++s) {
        SVNET_KNN_DMA4(s + 3);
        __builtin_amdgcn_sched_barrier(0);
        SVNET_KNN_PRODUCTS_HALF(av, bv, 0);
        __builtin_amdgcn_sched_barrier(0);
        SVNET_KNN_OPERANDS(s + 1, avn, bvn);                             // (past the last k-step: a landed buffer, values unused)
        __builtin_amdgcn_sched_barrier(0);
        SVNET_KNN_PRODUCTS_HALF(av, bv, 1);
        SVNET_KNN_STEP_END();
        av[0] = avn[0]; av[1] = avn[1];
#pragma unroll
        for (int tt = 0; tt < NTL; ++tt) bv[tt] = bvn[tt];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the ring's last requests, before the hand-over reuses the buffers
    __syncthreads();
#undef SVNET_KNN_DMA4
#undef SVNET_KNN_OPERANDS
#undef SVNET_KNN_PRODUCTS_HALF
#undef SVNET_KNN_STEP_END

    // hand-over, 16 queries per pass: D register g of tile (a, tt) in lane l = query 16 a + 4 kl + g, candidate CW wave + 16 tt + jl;
    // every wave then takes two queries of the pass - after both passes it holds 4 queries x all candidates, candidate lane + 64 t
    float acc[Q][T];
    int qid[Q];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();                                       // the first pass has been read
#pragma unroll
        for (int tt = 0; tt < NTL; ++tt)
#pragma unroll
            for (int g = 0; g < 4; ++g) rows[(4 * kl + g) * LDS_H + CW * wave + 16 * tt + jl] = dacc[pass][tt][g];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            qid[2 * pass + r] = qb0 + 16 * pass + 2 * wave + r;
#pragma unroll
            for (int t = 0; t < T; ++t) acc[2 * pass + r][t] = rows[(2 * wave + r) * LDS_H + 64 * t + lane];
        }
    }
#if SVNET_KNN_ABL >= 1 && SVNET_KNN_ABL <= 3   // diagnostic builds: no selection
    {
        float sm = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int t = 0; t < T; ++t) sm += acc[q][t];
        if (lane < k) idx_out[((size_t)b * N + qid[0]) * k + lane] = (int64_t)(sm > 0.f);
        return;
    }
#endif
    __syncthreads();                                                     // the hand-over has been read: its LDS now holds the selection's lists
    knn_select4<T>(acc, xxb, N, k, lane, rows + wave * (8 * SVNET_KNN_SLOTS), qid, idx_out + (size_t)b * N * k);
}

template <int T, int Q>
void launch_main(const float* xT, const float* xx, int64_t B, int N, int C, int k, int64_t* idx, hipStream_t st, bool mf8 = false) {
    dim3 grid((unsigned)svnet_cdiv(N, 4 * Q), (unsigned)B);
    int per = 0;
    if ((B & 7) == 0) { per = (int)grid.x; grid = dim3((unsigned)(grid.x * B), 1u); }   // XCD-aware cloud order (see the kernel)
    constexpr size_t stage_bytes = (size_t)KNN_CC * 64 * T * sizeof(float);
    // (WPB = 8 - 32 queries per pass over the cloud's table - was measured: 521 us against 425 for the three feature-space graphs of
    //  the bench; one 8-wave workgroup per CU loses more to its barriers than it saves in staging traffic)
    // SVNET_KNN_MFMA=1: the distance loop on the f32 matrix cores.  Measured (round 3, B=32 N=1024, the three feature-space graphs of
    // the bench): bit-identical neighbour lists on all 33 parity cases, and 183 us per call against 111 us for the vector form -
    // so the vector form stays the default.
    static const bool valu = getenv("SVNET_KNN_MFMA") == nullptr && !SVNET_KNN_FORCE_MF;
    if constexpr (T == 16 && Q == 4) {
        if (mf8) {                            // 32 queries per workgroup on the f32 matrix cores (channel-major table)
            constexpr size_t lds = (size_t)2 * 8 * (64 * T + 16) * sizeof(float);
            bool ok = true;                   // (per device; a failed opt-in is in svnet_last_error and the launch below fails)
            SVNET_LDS_OPTIN(ok, lds, "knn_mf8_kernel", (const void*)knn_mf8_kernel<T>);
            (void)ok;
            dim3 g8((unsigned)svnet_cdiv(N, 32), (unsigned)B);
            int per8 = 0;
            if ((B & 7) == 0) { per8 = (int)g8.x; g8 = dim3((unsigned)(g8.x * B), 1u); }
            hipLaunchKernelGGL((knn_mf8_kernel<T>), g8, dim3(512), lds, st, xT, xx, N, C, k, idx, per8);
            return;
        }
    }
    if constexpr ((T == 16 || T == 32) && Q == 4) {
        if ((N & 15) == 0 && !valu) {                                   // inner products on the f32 matrix cores
            hipLaunchKernelGGL((knn_main_kernel<T, Q, (T >= 4 && T <= 16), true, 4, true, true>), grid, dim3(256), (size_t)32768, st, xT, xx, N, C, k,
                               idx, per);
            return;
        }
        if ((N & 15) == 0) {   // (any C: the direct form stages nothing; 32 KB of LDS for the hand-over)
            hipLaunchKernelGGL((knn_main_kernel<T, Q, (T >= 4 && T <= 16), true, 4, true>), grid, dim3(256), (size_t)32768, st, xT, xx, N, C, k, idx,
                               per);
            return;
        }
    }
    if (T >= 4 && T <= 16 && (N & 3) == 0 && C >= 8)       // (T < 4: the staging chunk would not fill the 256 threads' float4 slots)
        hipLaunchKernelGGL((knn_main_kernel<T, Q, (T >= 4 && T <= 16)>), grid, dim3(256), stage_bytes, st, xT, xx, N, C, k, idx, per);
    else
        hipLaunchKernelGGL((knn_main_kernel<T, Q, false>), grid, dim3(256), 0, st, xT, xx, N, C, k, idx, per);
}

}  // namespace

// Which form of the main kernel a call takes decides the table's layout (one place: the fused producers of apply_knn.h ask it too)
struct KnnLayout { int il4; bool mf8; int64_t Cpad; };
static KnnLayout knn_layout(int64_t N, int64_t C) {
    // the matrix-core form of the main kernel (512 < N <= 2048, N % 16 == 0) reads the table with its channels interleaved in fours
    static const bool valu = getenv("SVNET_KNN_MFMA") == nullptr && !SVNET_KNN_FORCE_MF;
    static const bool mf8_on = getenv("SVNET_KNN_NO_MF8") == nullptr && SVNET_KNN_MF8;
    KnnLayout l;
    l.il4 = (N > 512 && N <= 2048 && (N & 15) == 0 && !valu) ? 1 : 0;
    l.mf8 = mf8_on && valu && N > 512 && N <= 1024 && (N & 15) == 0;              // knn_mf8_kernel: the table's clouds are C8 rows apart
    l.Cpad = l.il4 ? ((C + 3) & ~(int64_t)3) : (l.mf8 ? (C + 7) / 8 * 8 : C);
    return l;
}

bool svnet_knn_table_is_channel_major(int64_t N, int64_t C, int64_t* Cpad) {
    const KnnLayout l = knn_layout(N, C);
    if (Cpad) *Cpad = l.Cpad;
    return !l.il4;
}

static int knn_run_main(const float* xT, const float* xx, int64_t B, int64_t N, int64_t C, int k, int64_t* idx_out, bool mf8, hipStream_t st);

extern "C" int svnet_knn_table_fusable(int64_t B, int64_t N, int64_t C) {
    return B > 0 && N > 0 && (N % 32) == 0 && C >= 8 && C <= 384 && N <= 4096 && svnet_knn_table_is_channel_major(N, C, nullptr) ? 1 : 0;
}

extern "C" int svnet_knn_from_table_f32(const void* workspace, size_t workspace_bytes, int64_t B, int64_t N, int64_t C, int k,
                                        int64_t* idx_out, void* stream) {
    SVNET_REQUIRE(workspace && idx_out, SVNET_E_ARG, "svnet_knn_from_table_f32: null pointer");
    SVNET_REQUIRE(B >= 0 && N > 0 && C > 0 && k > 0 && k <= N, SVNET_E_ARG, "svnet_knn_from_table_f32: bad sizes B=%lld N=%lld C=%lld k=%d",
                  (long long)B, (long long)N, (long long)C, k);
    SVNET_REQUIRE(C <= 384 && N <= 4096 && k <= 64, SVNET_E_UNSUPPORTED, "svnet_knn_from_table_f32: outside C<=384, N<=4096, k<=64");
    SVNET_REQUIRE(workspace_bytes >= svnet_knn_workspace_bytes(B, N, C), SVNET_E_WORKSPACE, "svnet_knn_from_table_f32: workspace too small");
    const KnnLayout lay = knn_layout(N, C);
    SVNET_REQUIRE(!lay.il4, SVNET_E_UNSUPPORTED, "svnet_knn_from_table_f32: this configuration reads an interleaved table (svnet_knn_table_fusable)");
    if (B == 0) return SVNET_OK;
    const float* xT = (const float*)workspace;
    const float* xx = xT + B * N * ((C + 7) / 8 * 8);
    return knn_run_main(xT, xx, B, N, C, k, idx_out, lay.mf8, (hipStream_t)stream);
}

extern "C" size_t svnet_knn_workspace_bytes(int64_t B, int64_t N, int64_t C) {
    if (B < 0 || N < 0 || C < 0) return 0;
    return (size_t)(B * N * ((C + 7) / 8 * 8) + B * N) * sizeof(float) + 256;     // (channels padded to a multiple of 8: whole chunks of the matrix-core form)
}

static int knn_impl(const float* x, const float* x2, int64_t split, int64_t B, int64_t N, int64_t C, int64_t sb, int64_t sn, int64_t sc,
                    int xx_mode, int k, int64_t* idx_out, void* workspace, size_t workspace_bytes, void* stream);

extern "C" int svnet_knn_f32(const float* x, int64_t B, int64_t N, int64_t C, int64_t sb, int64_t sn, int64_t sc,
                             int xx_mode, int k, int64_t* idx_out, void* workspace, size_t workspace_bytes,
                             void* stream) {
    return knn_impl(x, nullptr, 0, B, N, C, sb, sn, sc, xx_mode, k, idx_out, workspace, workspace_bytes, stream);
}

extern "C" int svnet_knn_sv_f32(const float* s, int64_t Cs, const float* v, int64_t Cv3, int64_t B, int64_t N, int k, int64_t* idx_out,
                                void* workspace, size_t workspace_bytes, void* stream) {
    SVNET_REQUIRE(s && v && Cs > 0 && Cv3 > 0, SVNET_E_ARG, "svnet_knn_sv_f32: bad arguments");
    // rows of cat[s, v.flat] ([B,N,Cs+Cv3], channels contiguous): the transposed view knn() receives at sv_util.py:101
    return knn_impl(s, v, Cs, B, N, Cs + Cv3, N * Cs, Cs, 1, /*xx_mode=*/1, k, idx_out, workspace, workspace_bytes, stream);
}

static int knn_impl(const float* x, const float* x2, int64_t split, int64_t B, int64_t N, int64_t C, int64_t sb, int64_t sn, int64_t sc,
                    int xx_mode, int k, int64_t* idx_out, void* workspace, size_t workspace_bytes, void* stream) {
    SVNET_REQUIRE(x && idx_out && workspace, SVNET_E_ARG, "svnet_knn_f32: null pointer");
    SVNET_REQUIRE(B >= 0 && N > 0 && C > 0 && k > 0 && k <= N, SVNET_E_ARG, "svnet_knn_f32: bad sizes B=%lld N=%lld C=%lld k=%d",
                  (long long)B, (long long)N, (long long)C, k);
    SVNET_REQUIRE(xx_mode == 0 || xx_mode == 1, SVNET_E_ARG, "svnet_knn_f32: xx_mode must be 0 or 1");
    SVNET_REQUIRE(C <= 384, SVNET_E_UNSUPPORTED, "svnet_knn_f32: C=%lld > 384 (bit-exact contract ends where MKL splits K)", (long long)C);
    SVNET_REQUIRE(N <= 4096 && k <= 64, SVNET_E_UNSUPPORTED, "svnet_knn_f32: N=%lld k=%d outside N<=4096, k<=64", (long long)N, k);
    SVNET_REQUIRE(workspace_bytes >= svnet_knn_workspace_bytes(B, N, C), SVNET_E_WORKSPACE, "svnet_knn_f32: workspace too small");
    if (B == 0) return SVNET_OK;
    hipStream_t st = (hipStream_t)stream;
    float* xT = (float*)workspace;
    const int64_t C8 = (C + 7) / 8 * 8;
    float* xx = xT + B * N * C8;
    const KnnLayout lay = knn_layout(N, C);
    const int il4 = lay.il4;
    const bool mf8 = lay.mf8;
    // one wave per workgroup: a thread walks its point's row, so every load instruction of a wave touches 64 cache lines - the kernel is
    // bound by the CUs' address units, and 32 768 points in 256-thread workgroups put four such waves on each of only 128 CUs (61 -> 47 us
    // for the four calls of a step; staging the rows through LDS with coalesced loads was slower - 33 us per call whatever C: one wave
    // per SIMD and two dependent phases leave nothing to overlap)
    // (ROWS: see the kernel; the channel-first coordinate graph - sn == 1 - is coalesced as it is)
    const int64_t cut = x2 ? split : C;
    static const bool rows_on = getenv("SVNET_KNN_NO_ROWS") == nullptr;
    const bool rows = rows_on && !il4 && sc == 1 && sn == cut && sb == N * cut && (N & 63) == 0 && C >= 96 && C <= 255;   // (below ~96 channels the generic kernel is as fast: 11 us either way at C = 62)
    if (il4) hipLaunchKernelGGL(knn_prep_kernel<true>, dim3(svnet_grid(B * N, 64)), dim3(64), 0, st, x, B, N, C, sb, sn, sc, xx_mode, xT, xx, x2, split, C);
    else if (rows) hipLaunchKernelGGL(knn_prep_rows_kernel<SVNET_KNN_TP>, dim3(svnet_grid(B * N / SVNET_KNN_TP, 1)), dim3(256),
                                      (size_t)SVNET_KNN_TP * ((size_t)C | 1) * sizeof(float), st, x, B, N, C, xx_mode, xT, xx, x2, split, mf8 ? C8 : C);
    else hipLaunchKernelGGL(knn_prep_kernel<false>, dim3(svnet_grid(B * N, 64)), dim3(64), 0, st, x, B, N, C, sb, sn, sc, xx_mode, xT, xx, x2, split, mf8 ? C8 : C);
    SVNET_CHECK_LAUNCH("knn_prep_kernel");
#if SVNET_KNN_ABL == 5   // diagnostic build: the table preparation alone
    return SVNET_OK;
#endif
    return knn_run_main(xT, xx, B, N, C, k, idx_out, mf8, st);
}

static int knn_run_main(const float* xT, const float* xx, int64_t B, int64_t N, int64_t C, int k, int64_t* idx_out, bool mf8, hipStream_t st) {
    const int n = (int)N, c = (int)C;
    if (N <= 64) launch_main<1, 8>(xT, xx, B, n, c, k, idx_out, st);
    else if (N <= 128) launch_main<2, 8>(xT, xx, B, n, c, k, idx_out, st);
    else if (N <= 256) launch_main<4, 8>(xT, xx, B, n, c, k, idx_out, st);
    else if (N <= 512) launch_main<8, 8>(xT, xx, B, n, c, k, idx_out, st);
    else if (N <= 1024) launch_main<16, 4>(xT, xx, B, n, c, k, idx_out, st, mf8);   // 4 queries per wave: <= 128 VGPRs, 4 waves per SIMD
    else if (N <= 2048) launch_main<32, 4>(xT, xx, B, n, c, k, idx_out, st);
    else launch_main<64, 2>(xT, xx, B, n, c, k, idx_out, st);
    SVNET_CHECK_LAUNCH("knn_main_kernel");
    return SVNET_OK;
}
