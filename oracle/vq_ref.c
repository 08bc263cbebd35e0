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
 *       idx(n)    = argmin_k dist(n,k), ties -> lowest k; a NaN distance counts as -inf
 *                   (the first NaN column wins, as torch.argmin orders NaN).
 *    This is bit-for-bit what an f32 MFMA (v_mfma_f32_32x32x2_f32) k-ordered
 *    accumulation produces, so the HIP kernel can be required to match exactly.
 *
 * 3. Codebook EMA (row Q3, no reference code either - this is the definition):
 *       cnt(c)    = #{i : idx[i] == c}                       (rows with idx outside [0,K) are ignored)
 *       part(s,c) = sum of x[i] over the rows i of segment s = [s*SEG, (s+1)*SEG) with idx[i] == c,
 *                   added in increasing i (f32), SEG = 2048
 *       sum(c)    = part(0,c) + part(1,c) + ...              (f32, increasing s)
 *       cs(c)     = fmaf(1-decay, cnt(c), cs(c)*decay)       ("1-decay" rounded once to f32)
 *       es(c)     = fmaf(1-decay, sum(c), es(c)*decay)
 *       tot       = lanes l = 0..1023 each add cs(c), c = l, l+1024, ... in increasing c; then the
 *                   1024 lane sums are added in increasing l (all f32)
 *       sm(c)     = ((cs(c) + eps) / (tot + K*eps)) * tot    (Laplace smoothing; dead codes keep es/sm)
 *       code(c)   = es(c) / sm(c)
 *    Every step is a single correctly rounded f32 operation, so a GPU kernel that follows the same
 *    order is bit-exact.
 */
#include <math.h>
#include <stddef.h>
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
            if (dist != dist) dist = -INFINITY;
            if (dist < best) { best = dist; bi = j; }
        }
        idx[i] = bi;
        if (best_dist) best_dist[i] = best;
    }
}

#define EMA_SEG 2048

/* in-place update of cluster_size[k], embed_sum[k][d], codebook[k][d]; sum_ws is scratch of k*d floats */
void codebook_ema_ref(const float *x, const int32_t *idx, int64_t n, int k, int d, float decay, float eps,
                      float *cluster_size, float *embed_sum, float *codebook, float *sum_ws /* k*d */)
{
    const float omd = 1.0f - decay;
    for (int64_t t = 0; t < (int64_t)k * d; ++t) sum_ws[t] = 0.0f;
    float *part = (float *)__builtin_alloca(sizeof(float) * (size_t)d);
    for (int c = 0; c < k; ++c) {
        float cnt = 0.0f;
        for (int64_t s0 = 0; s0 < n; s0 += EMA_SEG) {
            const int64_t s1 = s0 + EMA_SEG < n ? s0 + EMA_SEG : n;
            for (int t = 0; t < d; ++t) part[t] = 0.0f;
            for (int64_t i = s0; i < s1; ++i)
                if (idx[i] == c) {
                    cnt += 1.0f;
                    for (int t = 0; t < d; ++t) part[t] = part[t] + x[i * d + t];
                }
            for (int t = 0; t < d; ++t) sum_ws[(int64_t)c * d + t] = sum_ws[(int64_t)c * d + t] + part[t];
        }
        cluster_size[c] = fmaf(omd, cnt, cluster_size[c] * decay);
        for (int t = 0; t < d; ++t)
            embed_sum[(int64_t)c * d + t] = fmaf(omd, sum_ws[(int64_t)c * d + t], embed_sum[(int64_t)c * d + t] * decay);
    }
    float lanes[1024];
    for (int l = 0; l < 1024; ++l) {
        float s = 0.0f;
        for (int c = l; c < k; c += 1024) s = s + cluster_size[c];
        lanes[l] = s;
    }
    float tot = 0.0f;
    for (int l = 0; l < 1024; ++l) tot = tot + lanes[l];
    const float den = tot + (float)k * eps;
    for (int c = 0; c < k; ++c) {
        const float sm = ((cluster_size[c] + eps) / den) * tot;
        for (int t = 0; t < d; ++t) codebook[(int64_t)c * d + t] = embed_sum[(int64_t)c * d + t] / sm;
    }
}
