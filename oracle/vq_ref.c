/* CPU restatement (plain C) of the two vector-quantiser searches on the path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see oracle/__init__.py): built by
 * oracle/Makefile into oracle/libvq_ref.so and loaded only by tests/, smoke()
 * and bench.py's cpu_baseline leg.
 *
 * 1. FSQ (reference quantizer.py:109-114 bound, :128-140 quantize, :177-181
 *    codes->index, :183-187 index->codes).  Arithmetic: t = z + shift (f32);
 *    th = tanh(t) evaluated in double and rounded once to f32; m = th * half_l
 *    (f32); b = m - offset (f32); q = rint(b) half-to-even; code = q / half_width;
 *    index = sum_j (q_j + half_width_j) * basis_j.  The per-dimension constants
 *    are produced on the host exactly as the reference produces them and passed in.
 *    The reference (torch CPU) evaluates tanh in f32 (<= 1 ulp); the two can differ
 *    by one level only where b lies within ~1 ulp of a half-integer.
 *
 * 2. Generic L2 nearest neighbour (SURVEY.md section 8a row Q3 - no reference
 *    code; this file is the definition, PARITY UNPINNED beyond the FSQ grid):
 *       dot(n,k)  = fma-chain over d = 0..D-1, acc = fmaf(x[n][d], c[k][d], acc), acc0 = 0
 *       cn(k)     = fma-chain over d of c[k][d]^2
 *       dist(n,k) = fmaf(-2, dot(n,k), cn(k))            (|x|^2 dropped: constant per row)
 *       idx(n)    = argmin_k dist(n,k), ties -> lowest k.
 *    This is bit-for-bit what an f32 MFMA (v_mfma_f32_32x32x2_f32) k-ordered
 *    accumulation produces, so the HIP kernel can be required to match exactly.
 */
#include <math.h>
#include <stdint.h>

void fsq_ref_forward(const float *z, int64_t n, int d, const float *half_l, const float *offset,
                     const float *shift, const int32_t *half_width, const int32_t *basis,
                     float *codes, int32_t *idx)
{
    for (int64_t i = 0; i < n; ++i) {
        int32_t id = 0;
        for (int j = 0; j < d; ++j) {
            volatile float t = z[i * d + j] + shift[j];
            volatile float th = (float)tanh((double)t);
            volatile float m = th * half_l[j];
            volatile float b = m - offset[j];
            float q = rintf(b); /* default rounding mode: half to even */
            if (codes) codes[i * d + j] = q / (float)half_width[j];
            id += ((int32_t)q + half_width[j]) * basis[j];
        }
        idx[i] = id;
    }
}

void fsq_ref_decode(const int64_t *idx, int64_t n, int d, const int32_t *levels, const int32_t *half_width,
                    const int32_t *basis, float *codes)
{
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < d; ++j) {
            int32_t l = (int32_t)((idx[i] / basis[j]) % levels[j]);
            codes[i * d + j] = (float)(l - half_width[j]) / (float)half_width[j];
        }
}

void l2_codebook_norms_ref(const float *c, int k, int d, float *cn)
{
    for (int j = 0; j < k; ++j) {
        float acc = 0.0f;
        for (int t = 0; t < d; ++t) acc = fmaf(c[(int64_t)j * d + t], c[(int64_t)j * d + t], acc);
        cn[j] = acc;
    }
}

/* rows [n0, n1) - callers thread over row ranges */
void l2_argmin_ref(const float *x, const float *c, const float *cn, int64_t n0, int64_t n1, int k, int d,
                   int32_t *idx, float *best_dist)
{
    for (int64_t i = n0; i < n1; ++i) {
        const float *xi = x + i * d;
        float best = INFINITY;
        int32_t bi = 0;
        for (int j = 0; j < k; ++j) {
            const float *cj = c + (int64_t)j * d;
            float acc = 0.0f;
            for (int t = 0; t < d; ++t) acc = fmaf(xi[t], cj[t], acc);
            float dist = fmaf(-2.0f, acc, cn[j]);
            if (dist < best) { best = dist; bi = j; }
        }
        idx[i] = bi;
        if (best_dist) best_dist[i] = best;
    }
}
