/*
 * ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Exact-arithmetic CPU restatement of the reference k-NN
 *   /root/reference/models/utils/sv_util.py:19-25   (knn)
 * as torch 2.10 CPU evaluates it:
 *
 *   inner = -2 * matmul(x^T, x)            sv_util.py:20   sequential fp32 FMA chain over channels
 *   xx    = sum(x**2, dim=1)               sv_util.py:21   ATen cascade / vectorised row sum
 *   pd    = -xx - inner - xx^T             sv_util.py:22   two fp32 subtractions, left to right
 *   idx   = pd.topk(k)[1]                  sv_util.py:24   descending; ties -> lowest index (ours)
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC   (see oracle/Makefile)
 * -ffp-contract=off matters: every rounding below is intentional.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define XX_OUTER 0 /* reduction over a non-contiguous (outer) dim: x is a contiguous [B,C,N] tensor */
#define XX_INNER 1 /* reduction over the contiguous dim: x is a transposed view of [B,N,C]          */

/* ATen multi_row_sum cascade for one row: 4 levels, level step 16 (sizes < 2^20). */
static float cascade_sum(const float *sq, int64_t stride, int64_t size) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t level_power = 4, level_step = 16, level_mask = 15;
    int64_t i = 0;
    while (i + level_step <= size) {
        for (int64_t j = 0; j < level_step; ++j, ++i) acc[0] += sq[i * stride];
        for (int j = 1; j < 4; ++j) {
            acc[j] += acc[j - 1];
            acc[j - 1] = 0.f;
            if ((i & (level_mask << (j * level_power))) != 0) break;
        }
    }
    for (; i < size; ++i) acc[0] += sq[i * stride];
    for (int j = 1; j < 4; ++j) acc[0] += acc[j];
    return acc[0];
}

/* ATen row_sum: 4 interleaved partial sums (each a cascade over size/4 terms), leftovers into
 * partial 0, then partial0 + partial1 + partial2 + partial3. */
static float ilp4_sum(const float *sq, int64_t stride, int64_t size) {
    float part[4];
    const int64_t ng = size / 4;
    for (int r = 0; r < 4; ++r) part[r] = cascade_sum(sq + r * stride, 4 * stride, ng);
    for (int64_t i = ng * 4; i < size; ++i) part[0] += sq[i * stride];
    for (int r = 1; r < 4; ++r) part[0] += part[r];
    return part[0];
}

/* ATen vectorized_outer_sum: columns are processed 32 at a time (4 vectors of 8) with a plain
 * cascade per column; the remaining N % 32 columns go through row_sum (ilp4_sum). */
static float outer_col_sum(const float *sq, int64_t C, int64_t n, int64_t N) {
    return (n < (N / 32) * 32) ? cascade_sum(sq, 1, C) : ilp4_sum(sq, 1, C);
}

/* ATen vectorized_inner_sum for one contiguous row of `size` floats, 8-lane vectors, ilp 4. */
static float inner_row_sum(const float *sq, int64_t size) {
    enum { V = 8, ILP = 4 };
    if (size < V) return ilp4_sum(sq, 1, size); /* rows shorter than one vector take the scalar row_sum */
    const int64_t nv = size / V;       /* whole vectors                */
    const int64_t ng = nv / ILP;       /* groups of 4 vectors          */
    float lane[V];
    for (int l = 0; l < V; ++l) {
        /* per lane: 4 interleaved partial sums, each a cascade over ng terms with stride 4 vectors */
        float part[ILP];
        for (int r = 0; r < ILP; ++r) part[r] = cascade_sum(sq + (int64_t)r * V + l, (int64_t)ILP * V, ng);
        for (int64_t g = ng * ILP; g < nv; ++g) part[0] += sq[g * V + l];
        for (int r = 1; r < ILP; ++r) part[0] += part[r];
        lane[l] = part[0];
    }
    float fin = 0.f;
    for (int64_t c = nv * V; c < size; ++c) fin += sq[c];
    for (int l = 0; l < V; ++l) fin += lane[l];
    return fin;
}

/*
 * x is addressed as x[b*sb + n*sn + c*sc].
 * idx_out: [B,N,k] int64 (cloud-local neighbour index, nearest first).
 * pd_out : optional [B,N,N] float32 (may be NULL).
 * returns 0 on success, negative on bad arguments.
 */
int svnet_oracle_knn(const float *x, int64_t B, int64_t N, int64_t C, int64_t sb, int64_t sn, int64_t sc,
                     int xx_mode, int k, int64_t *idx_out, float *pd_out) {
    if (!x || !idx_out || B < 0 || N <= 0 || C <= 0 || k <= 0 || k > N) return -1;
    if (C > 384) return -4; /* MKL splits the K loop beyond 384: the single FMA chain no longer describes torch's matmul */
    if (xx_mode != XX_OUTER && xx_mode != XX_INNER) return -2;

    float *xx = (float *)malloc(sizeof(float) * (size_t)(B * N));
    if (!xx) return -3;

#pragma omp parallel
    {
        float *sq = (float *)malloc(sizeof(float) * (size_t)C);
        float *row = (float *)malloc(sizeof(float) * (size_t)N);
#pragma omp for schedule(static)
        for (int64_t bn = 0; bn < B * N; ++bn) {
            const float *p = x + (bn / N) * sb + (bn % N) * sn;
            for (int64_t c = 0; c < C; ++c) {
                float v = p[c * sc];
                sq[c] = v * v; /* x**2: a separate, rounded multiply */
            }
            xx[bn] = (xx_mode == XX_OUTER) ? outer_col_sum(sq, C, bn % N, N) : inner_row_sum(sq, C);
        }
#pragma omp for schedule(dynamic, 16)
        for (int64_t bi = 0; bi < B * N; ++bi) {
            const int64_t b = bi / N;
            const float *xi = x + b * sb + (bi % N) * sn;
            const float xxi = xx[bi];
            for (int64_t j = 0; j < N; ++j) {
                const float *xj = x + b * sb + j * sn;
                float acc = xi[0] * xj[0];
                for (int64_t c = 1; c < C; ++c) acc = fmaf(xi[c * sc], xj[c * sc], acc);
                float inner = -2.0f * acc;
                float t = -xx[b * N + j] - inner;
                row[j] = t - xxi;
            }
            if (pd_out) memcpy(pd_out + bi * N, row, sizeof(float) * (size_t)N);
            int64_t *out = idx_out + bi * k;
            for (int s = 0; s < k; ++s) { /* k passes of arg-max; ties -> lowest index */
                int64_t best = -1;
                float bv = 0.f;
                for (int64_t j = 0; j < N; ++j) {
                    float v = row[j];
                    if (isnan(v)) continue;
                    if (best < 0 || v > bv) { best = j; bv = v; }
                }
                if (best < 0) best = 0;
                out[s] = best;
                row[best] = NAN; /* consumed */
            }
        }
        free(sq);
        free(row);
    }
    free(xx);
    return 0;
}
