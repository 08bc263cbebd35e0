// Effective-weight preparation for the implicit GEMM: layout change (reference (K, C, R*S) ->
// KRSC and CRSK) fused with weight norm / spectral norm, plus the matching backward passes.
#include "common.h"
#include <cstring>

namespace mq {

// one block per output channel k.  v (K, C, RS).
__global__ void weight_prep_kernel(int mode, const float *__restrict__ v, const float *__restrict__ g, int K, int C, int RS,
                                   float *__restrict__ w_krsc, float *__restrict__ w_crsk, float *__restrict__ inv_norm)
{
    __shared__ float red[4];
    const int k = blockIdx.x;
    const int fan = C * RS;
    const float *vk = v + (size_t)k * fan;
    float scale = 1.0f;
    if (mode == 1) {
        float s = 0.f;
        for (int i = threadIdx.x; i < fan; i += 256) s += vk[i] * vk[i];
        s = block_sum<256>(s, red);
        const float inv = 1.0f / sqrtf(s);
        if (threadIdx.x == 0 && inv_norm) inv_norm[k] = inv;
        scale = g[k] * inv;
    }
    const int cbc = chan_block(C), cbk = chan_block(K);
    for (int i = threadIdx.x; i < fan; i += 256) {
        const int c = i / RS, rs = i - c * RS;
        const float w = vk[i] * scale;
        if (w_krsc) w_krsc[(size_t)k * fan + kidx(rs, c, RS, cbc)] = w;
        if (w_crsk) w_crsk[(size_t)c * RS * K + kidx(rs, k, RS, cbk)] = w;
    }
}

__global__ void weight_prep_bwd_kernel(int mode, const float *__restrict__ dw_krsc, const float *__restrict__ v,
                                       const float *__restrict__ g, const float *__restrict__ inv_norm, int K, int C, int RS,
                                       float *__restrict__ dv, float *__restrict__ dg)
{
    __shared__ float red[4];
    const int k = blockIdx.x;
    const int fan = C * RS;
    const float *vk = v + (size_t)k * fan;
    const int cbc = chan_block(C);
    const float *dwk = dw_krsc + (size_t)k * fan;
    if (mode == 0) {
        for (int i = threadIdx.x; i < fan; i += 256) {
            const int c = i / RS, rs = i - c * RS;
            dv[(size_t)k * fan + i] = dwk[kidx(rs, c, RS, cbc)];
        }
        return;
    }
    float dot = 0.f;
    for (int i = threadIdx.x; i < fan; i += 256) {
        const int c = i / RS, rs = i - c * RS;
        dot += dwk[kidx(rs, c, RS, cbc)] * vk[i];
    }
    dot = block_sum<256>(dot, red);
    const float inv = inv_norm[k], gk = g[k];
    if (threadIdx.x == 0) dg[k] = dot * inv;
    if (fan == 1) {   // w = g * sign(v): the gradient wrt v is exactly zero (reference: measured 0)
        if (threadIdx.x == 0) dv[k] = 0.f;
        return;
    }
    const float a = gk * inv, b = gk * dot * inv * inv * inv;
    for (int i = threadIdx.x; i < fan; i += 256) {
        const int c = i / RS, rs = i - c * RS;
        dv[(size_t)k * fan + i] = a * dwk[kidx(rs, c, RS, cbc)] - b * vk[i];
    }
}

// ------------------------------------------------------------------------------------------ spectral norm
// t[j] = sum_k W[k][j] * u[k].  Block = 32 columns x 8 row lanes: lane r of a column adds rows r, r + 8, ... (four loads in flight),
// the eight partial sums are combined in the order r = 0 .. 7.  (One thread per column walking all K rows - the first version -
// left a 13 MB matrix to 50 blocks of dependent loads: 190 us where the bytes need 3.)
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const float *__restrict__ W, const float *__restrict__ u, int K, int fan,
                                                      float *__restrict__ t)
{
    __shared__ float part[8][33];
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < fan) {
        int k = r;
        for (; k + 24 < K; k += 32) {
            s0 += W[(size_t)k * fan + j] * u[k];
            s1 += W[(size_t)(k + 8) * fan + j] * u[k + 8];
            s2 += W[(size_t)(k + 16) * fan + j] * u[k + 16];
            s3 += W[(size_t)(k + 24) * fan + j] * u[k + 24];
        }
        for (; k < K; k += 8) s0 += W[(size_t)k * fan + j] * u[k];
    }
    part[r][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (r == 0 && j < fan) {
        float a = part[0][c];
#pragma unroll
        for (int i = 1; i < 8; ++i) a += part[i][c];
        t[j] = a;
    }
}
// s[k] = sum_j W[k][j] * v[j]      (block per k)
__global__ void sn_w_v_kernel(const float *__restrict__ W, const float *__restrict__ v, int fan, float *__restrict__ s)
{
    __shared__ float red[4];
    const int k = blockIdx.x;
    float a = 0.f;
    for (int j = threadIdx.x; j < fan; j += 256) a += W[(size_t)k * fan + j] * v[j];
    a = block_sum<256>(a, red);
    if (threadIdx.x == 0) s[k] = a;
}
// out = in / max(||in||, eps)   (single block of 1024)
__global__ void sn_normalize_kernel(const float *__restrict__ in, int n, float eps, float *__restrict__ out)
{
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) s += in[i] * in[i];
    s = block_sum<1024>(s, red);
    const float inv = 1.0f / fmaxf(sqrtf(s), eps);
    for (int i = threadIdx.x; i < n; i += 1024) out[i] = in[i] * inv;
}
// sigma = dot(u, s)   (single block)
__global__ void sn_sigma_kernel(const float *__restrict__ u, const float *__restrict__ s, int K, float *__restrict__ sigma)
{
    __shared__ float red[16];
    float a = 0.f;
    for (int i = threadIdx.x; i < K; i += 1024) a += u[i] * s[i];
    a = block_sum<1024>(a, red);
    if (threadIdx.x == 0) sigma[0] = a;
}
__global__ void sn_scale_kernel(const float *__restrict__ W, const float *__restrict__ sigma, int K, int C, int RS,
                                float *__restrict__ w_krsc, float *__restrict__ w_crsk)
{
    const int fan = C * RS;
    const int64_t n = (int64_t)K * fan;
    const float inv = 1.0f / sigma[0];
    const int cbc = chan_block(C), cbk = chan_block(K);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / fan), r = (int)(i - (int64_t)k * fan);
        const int c = r / RS, rs = r - c * RS;
        const float w = W[i] * inv;
        if (w_krsc) w_krsc[(size_t)k * fan + kidx(rs, c, RS, cbc)] = w;
        if (w_crsk) w_crsk[(size_t)c * RS * K + kidx(rs, k, RS, cbk)] = w;
    }
}
// partial[b] = sum over a slice of dW (KRSC) * W (KCRS)
__global__ void sn_bwd_dot_kernel(const float *__restrict__ dw_krsc, const float *__restrict__ W, int K, int C, int RS,
                                  float *__restrict__ partial)
{
    __shared__ float red[4];
    const int fan = C * RS;
    const int64_t n = (int64_t)K * fan;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i / fan), r = (int)(i - (int64_t)k * fan);
        const int c = r / RS, rs = r - c * RS;
        s += dw_krsc[(size_t)k * fan + kidx(rs, c, RS, chan_block(C))] * W[i];
    }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void sn_bwd_apply_kernel(const float *__restrict__ dw_krsc, const float *__restrict__ u, const float *__restrict__ v,
                                    const float *__restrict__ sigma, const float *__restrict__ partial, int nparts, int K,
                                    int C, int RS, float *__restrict__ dW)
{
    const int fan = C * RS;
    const int64_t n = (int64_t)K * fan;
    float dot = 0.f;
    for (int i = 0; i < nparts; ++i) dot += partial[i];
    const float inv = 1.0f / sigma[0];
    const float coef = dot * inv * inv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / fan), r = (int)(i - (int64_t)k * fan);
        const int c = r / RS, rs = r - c * RS;
        dW[i] = dw_krsc[(size_t)k * fan + kidx(rs, c, RS, chan_block(C))] * inv - coef * u[k] * v[r];
    }
}

// ---- the same layer of up to MQ_SN_MAX_BANDS band discriminators in one launch each (no power iteration: eval-mode
// discriminators, i.e. every step after the first generator step of an epoch).  Band g reads its own weight_orig / u / v
// and writes rows [g*K, (g+1)*K) of the stacked KRSC (and [g*C, (g+1)*C) of the stacked CRSK) filters.
struct SnBands {
    const float *w[MQ_SN_MAX_BANDS];
    const float *u[MQ_SN_MAX_BANDS];
    const float *v[MQ_SN_MAX_BANDS];
};
struct SnBandsOut {
    float *dw[MQ_SN_MAX_BANDS];
};
// s[g][k] = sum_j W_g[k][j] * v_g[j]      grid (K, G)
__global__ void sn_w_v_bands_kernel(SnBands b, int K, int fan, float *__restrict__ s)
{
    __shared__ float red[4];
    const int k = blockIdx.x, g = blockIdx.y;
    const float *W = b.w[g], *v = b.v[g];
    float a = 0.f;
    for (int j = threadIdx.x; j < fan; j += 256) a += W[(size_t)k * fan + j] * v[j];
    a = block_sum<256>(a, red);
    if (threadIdx.x == 0) s[(size_t)g * K + k] = a;
}
// sigma_g = dot(u_g, s_g) recomputed by every block (K <= a few hundred values), then the scaled KRSC / CRSK filters
__global__ __launch_bounds__(256) void sn_scale_bands_kernel(SnBands b, const float *__restrict__ s, int K, int C, int RS,
                                                             float *__restrict__ w_krsc, float *__restrict__ w_crsk,
                                                             float *__restrict__ sigma_out)
{
    __shared__ float red[4];
    const int g = blockIdx.y;
    const float *W = b.w[g], *u = b.u[g];
    float a = 0.f;
    for (int i = threadIdx.x; i < K; i += 256) a += u[i] * s[(size_t)g * K + i];
    a = block_sum<256>(a, red);
    __shared__ float sig;
    if (threadIdx.x == 0) sig = a;
    __syncthreads();
    const float sigma = sig;
    if (blockIdx.x == 0 && threadIdx.x == 0) sigma_out[g] = sigma;
    const float inv = 1.0f / sigma;
    const int fan = C * RS;
    const int64_t n = (int64_t)K * fan;
    const int cbc = chan_block(C), cbk = chan_block(K);
    float *ok = w_krsc ? w_krsc + (size_t)g * n : nullptr;
    float *oc = w_crsk ? w_crsk + (size_t)g * n : nullptr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / fan), r = (int)(i - (int64_t)k * fan);
        const int c = r / RS, rs = r - c * RS;
        const float w = W[i] * inv;
        if (ok) ok[(size_t)k * fan + kidx(rs, c, RS, cbc)] = w;
        if (oc) oc[(size_t)c * RS * K + kidx(rs, k, RS, cbk)] = w;
    }
}
// partial[g][b] = sum over a slice of dW_g (stacked KRSC) * W_g (KCRS)      grid (nparts, G)
__global__ void sn_bwd_dot_bands_kernel(const float *__restrict__ dw_krsc, SnBands b, int K, int C, int RS, float *__restrict__ partial)
{
    __shared__ float red[4];
    const int g = blockIdx.y;
    const int fan = C * RS;
    const int64_t n = (int64_t)K * fan;
    const float *dw = dw_krsc + (size_t)g * n, *W = b.w[g];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i / fan), r = (int)(i - (int64_t)k * fan);
        const int c = r / RS, rs = r - c * RS;
        s += dw[(size_t)k * fan + kidx(rs, c, RS, chan_block(C))] * W[i];
    }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) partial[(size_t)g * gridDim.x + blockIdx.x] = s;
}
__global__ void sn_bwd_apply_bands_kernel(const float *__restrict__ dw_krsc, SnBands b, const float *__restrict__ sigma,
                                          const float *__restrict__ partial, int nparts, int K, int C, int RS, SnBandsOut o)
{
    const int g = blockIdx.y;
    const int fan = C * RS;
    const int64_t n = (int64_t)K * fan;
    const float *dw = dw_krsc + (size_t)g * n, *u = b.u[g], *v = b.v[g];
    float *dW = o.dw[g];
    float dot = 0.f;
    for (int i = 0; i < nparts; ++i) dot += partial[(size_t)g * nparts + i];
    const float inv = 1.0f / sigma[g];
    const float coef = dot * inv * inv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / fan), r = (int)(i - (int64_t)k * fan);
        const int c = r / RS, rs = r - c * RS;
        dW[i] = dw[(size_t)k * fan + kidx(rs, c, RS, chan_block(C))] * inv - coef * u[k] * v[r];
    }
}

}  // namespace mq

using namespace mq;

static int sn_bands_nparts(int K, int C, int RS)
{
    int nparts = (int)cdiv((int64_t)K * C * RS, 256 * 16);
    return nparts > 256 ? 256 : (nparts < 1 ? 1 : nparts);
}

extern "C" int mq_spectral_prep_bands(const float *const *w_orig, const float *const *u, const float *const *v, int bands, int K,
                                      int C, int RS, float *w_krsc, float *w_crsk, float *sigma_out, float *scratch,
                                      mqStream stream)
{
    MQ_REQUIRE(w_orig && u && v && sigma_out && scratch && (w_krsc || w_crsk) && K > 0 && C > 0 && RS > 0,
               "mq_spectral_prep_bands: bad arguments");
    MQ_REQUIRE(bands > 0 && bands <= MQ_SN_MAX_BANDS, "mq_spectral_prep_bands: 1 .. %d bands, got %d", MQ_SN_MAX_BANDS, bands);
    SnBands b;
    memset(&b, 0, sizeof(b));
    for (int g = 0; g < bands; ++g) {
        MQ_REQUIRE(w_orig[g] && u[g] && v[g], "mq_spectral_prep_bands: null pointer for band %d", g);
        b.w[g] = w_orig[g]; b.u[g] = u[g]; b.v[g] = v[g];
    }
    const int fan = C * RS;
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(sn_w_v_bands_kernel, dim3(K, bands), dim3(256), 0, st, b, K, fan, scratch);
    int gx = ew_grid((int64_t)K * fan, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(sn_scale_bands_kernel, dim3(gx, bands), dim3(256), 0, st, b, scratch, K, C, RS, w_krsc, w_crsk, sigma_out);
    MQ_LAUNCH_CHECK("mq_spectral_prep_bands");
    return 0;
}

extern "C" int mq_spectral_prep_bands_bwd(const float *dw_krsc, const float *const *w_orig, const float *const *u,
                                          const float *const *v, const float *sigma, int bands, int K, int C, int RS,
                                          float *const *dw_orig, float *scratch, mqStream stream)
{
    MQ_REQUIRE(dw_krsc && w_orig && u && v && sigma && dw_orig && scratch && K > 0 && C > 0 && RS > 0,
               "mq_spectral_prep_bands_bwd: bad arguments");
    MQ_REQUIRE(bands > 0 && bands <= MQ_SN_MAX_BANDS, "mq_spectral_prep_bands_bwd: 1 .. %d bands, got %d", MQ_SN_MAX_BANDS, bands);
    SnBands b;
    SnBandsOut o;
    memset(&b, 0, sizeof(b));
    memset(&o, 0, sizeof(o));
    for (int g = 0; g < bands; ++g) {
        MQ_REQUIRE(w_orig[g] && u[g] && v[g] && dw_orig[g], "mq_spectral_prep_bands_bwd: null pointer for band %d", g);
        b.w[g] = w_orig[g]; b.u[g] = u[g]; b.v[g] = v[g]; o.dw[g] = dw_orig[g];
    }
    const int nparts = sn_bands_nparts(K, C, RS);
    int gx = ew_grid((int64_t)K * C * RS, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(sn_bwd_dot_bands_kernel, dim3(nparts, bands), dim3(256), 0, S(stream), dw_krsc, b, K, C, RS, scratch);
    hipLaunchKernelGGL(sn_bwd_apply_bands_kernel, dim3(gx, bands), dim3(256), 0, S(stream), dw_krsc, b, sigma, scratch, nparts, K, C,
                       RS, o);
    MQ_LAUNCH_CHECK("mq_spectral_prep_bands_bwd");
    return 0;
}

extern "C" int mq_weight_prep(int mode, const float *v, const float *g, int K, int C, int RS, float *w_krsc, float *w_crsk,
                              float *inv_norm, mqStream stream)
{
    MQ_REQUIRE(v && K > 0 && C > 0 && RS > 0 && (w_krsc || w_crsk), "mq_weight_prep: bad arguments");
    MQ_REQUIRE(mode == 0 || (mode == 1 && g && inv_norm), "mq_weight_prep: mode 1 needs g and inv_norm");
    hipLaunchKernelGGL(weight_prep_kernel, dim3(K), dim3(256), 0, S(stream), mode, v, g, K, C, RS, w_krsc, w_crsk, inv_norm);
    MQ_LAUNCH_CHECK("mq_weight_prep");
    return 0;
}

extern "C" int mq_weight_prep_bwd(int mode, const float *dw_krsc, const float *v, const float *g, const float *inv_norm,
                                  int K, int C, int RS, float *dv, float *dg, mqStream stream)
{
    MQ_REQUIRE(dw_krsc && v && dv && K > 0 && C > 0 && RS > 0, "mq_weight_prep_bwd: bad arguments");
    MQ_REQUIRE(mode == 0 || (mode == 1 && g && inv_norm && dg), "mq_weight_prep_bwd: mode 1 needs g, inv_norm, dg");
    hipLaunchKernelGGL(weight_prep_bwd_kernel, dim3(K), dim3(256), 0, S(stream), mode, dw_krsc, v, g, inv_norm, K, C, RS, dv,
                       dg);
    MQ_LAUNCH_CHECK("mq_weight_prep_bwd");
    return 0;
}

extern "C" int mq_spectral_prep(const float *w_orig, float *u, float *v, int K, int C, int RS, int power_iteration, float eps,
                                float *w_krsc, float *w_crsk, float *sigma_out, float *scratch, mqStream stream)
{
    MQ_REQUIRE(w_orig && u && v && sigma_out && scratch && K > 0 && C > 0 && RS > 0, "mq_spectral_prep: bad arguments");
    const int fan = C * RS;
    float *t = scratch;        // fan
    float *s = scratch + fan;  // K
    hipStream_t st = S(stream);
    if (power_iteration) {
        hipLaunchKernelGGL(sn_wt_u_kernel, dim3((unsigned)cdiv(fan, 32)), dim3(256), 0, st, w_orig, u, K, fan, t);
        hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, st, t, fan, eps, v);
        hipLaunchKernelGGL(sn_w_v_kernel, dim3(K), dim3(256), 0, st, w_orig, v, fan, s);
        hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, st, s, K, eps, u);
    } else {
        hipLaunchKernelGGL(sn_w_v_kernel, dim3(K), dim3(256), 0, st, w_orig, v, fan, s);
    }
    hipLaunchKernelGGL(sn_sigma_kernel, dim3(1), dim3(1024), 0, st, u, s, K, sigma_out);
    hipLaunchKernelGGL(sn_scale_kernel, dim3(ew_grid((int64_t)K * fan, 256)), dim3(256), 0, st, w_orig, sigma_out, K, C, RS,
                       w_krsc, w_crsk);
    MQ_LAUNCH_CHECK("mq_spectral_prep");
    return 0;
}

extern "C" int mq_spectral_prep_bwd(const float *dw_krsc, const float *w_orig, const float *u, const float *v,
                                    const float *sigma, int K, int C, int RS, float *dw_orig, float *scratch, mqStream stream)
{
    MQ_REQUIRE(dw_krsc && w_orig && u && v && sigma && dw_orig && scratch && K > 0 && C > 0 && RS > 0,
               "mq_spectral_prep_bwd: bad arguments");
    const int64_t n = (int64_t)K * C * RS;
    int nparts = (int)cdiv(n, 256 * 16);
    if (nparts > 256) nparts = 256;
    if (nparts < 1) nparts = 1;
    hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(nparts), dim3(256), 0, S(stream), dw_krsc, w_orig, K, C, RS, scratch);
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), dw_krsc, u, v, sigma, scratch,
                       nparts, K, C, RS, dw_orig);
    MQ_LAUNCH_CHECK("mq_spectral_prep_bwd");
    return 0;
}
