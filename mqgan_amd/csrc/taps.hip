// Thin convolutions.  A conv whose input (C = 1: the first layer of each discriminator and of the refiner) or
// output (K = 1: the logits convs, the refiner's last conv) has a single channel wastes 31/32 of an MFMA tile
// and, as an implicit GEMM, re-reads the wide tensor once per tap.  They run instead as
//     C = 1:  cols = gather(img)            (N,P,Q,T)   then a 1x1 GEMM  T -> K   on the vector path
//     K = 1:  u = 1x1 GEMM  C -> T          (N,H,W,T)   then y = reduce(u) (+ bias)
// with T = R*S rounded up to a multiple of 4, so the wide tensor is read exactly once and the single-channel
// side (a few MB) is expanded / folded by the four kernels below.  scatter / expand are the transposes used by
// the backward pass, written in gather form (one thread per output element, fixed summation order).
#include "common.h"

namespace mq {

struct TapGeom {
    int N, H, W, P, Q, R, S, sh, sw, ph, pw, dh, dw, T, RS, n_per_group;
};

// cols[n,p,q,t] = img[n, p*sh - ph + r*dh, q*sw - pw + s*dw]   (t = r*S + s < RS; 0 elsewhere)
__global__ void taps_gather_kernel(TapGeom g, const float *__restrict__ img, float *__restrict__ cols, int64_t total)
{
    const int T4 = g.T >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t4 = (int)(i % T4);
        int64_t pix = i / T4;
        const int q = (int)(pix % g.Q);
        pix /= g.Q;
        const int p = (int)(pix % g.P), n = (int)(pix / g.P);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t = 4 * t4 + e;
            const int r = t / g.S, s = t - r * g.S;
            const int ih = p * g.sh - g.ph + r * g.dh, iw = q * g.sw - g.pw + s * g.dw;
            const bool ok = t < g.RS && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
            v[e] = ok ? img[((int64_t)n * g.H + ih) * g.W + iw] : 0.f;
        }
        *reinterpret_cast<float4 *>(cols + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// img[n,h,w] = sum_t cols[n,p,q,t]  over the (p,q,t) with  p*sh - ph + r*dh == h,  q*sw - pw + s*dw == w
__global__ void taps_scatter_kernel(TapGeom g, const float *__restrict__ cols, float *__restrict__ img, int64_t total)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % g.W);
        int64_t rest = i / g.W;
        const int h = (int)(rest % g.H), n = (int)(rest / g.H);
        float acc = 0.f;
        for (int r = 0; r < g.R; ++r) {
            const int th = h + g.ph - r * g.dh;
            if (th < 0 || th % g.sh != 0) continue;
            const int p = th / g.sh;
            if (p >= g.P) continue;
            for (int s = 0; s < g.S; ++s) {
                const int tw = w + g.pw - s * g.dw;
                if (tw < 0 || tw % g.sw != 0) continue;
                const int q = tw / g.sw;
                if (q >= g.Q) continue;
                acc += cols[(((int64_t)n * g.P + p) * g.Q + q) * g.T + r * g.S + s];
            }
        }
        img[i] = acc;
    }
}

// y[n,p,q] = bias[group(n)] + sum_t u[n, p*sh - ph + r*dh, q*sw - pw + s*dw, t]
__global__ void taps_reduce_kernel(TapGeom g, const float *__restrict__ u, const float *__restrict__ bias,
                                   float *__restrict__ y, int64_t total)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % g.Q);
        int64_t rest = i / g.Q;
        const int p = (int)(rest % g.P), n = (int)(rest / g.P);
        float acc = bias ? bias[n / g.n_per_group] : 0.f;
        for (int r = 0; r < g.R; ++r) {
            const int ih = p * g.sh - g.ph + r * g.dh;
            if ((unsigned)ih >= (unsigned)g.H) continue;
            for (int s = 0; s < g.S; ++s) {
                const int iw = q * g.sw - g.pw + s * g.dw;
                if ((unsigned)iw >= (unsigned)g.W) continue;
                acc += u[(((int64_t)n * g.H + ih) * g.W + iw) * g.T + r * g.S + s];
            }
        }
        y[i] = acc;
    }
}

// du[n,h,w,t] = dy[n,p,q]  where  p*sh - ph + r*dh == h,  q*sw - pw + s*dw == w   (0 if no such output pixel)
__global__ void taps_expand_kernel(TapGeom g, const float *__restrict__ dy, float *__restrict__ du, int64_t total)
{
    const int T4 = g.T >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t4 = (int)(i % T4);
        int64_t pix = i / T4;
        const int w = (int)(pix % g.W);
        pix /= g.W;
        const int h = (int)(pix % g.H), n = (int)(pix / g.H);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t = 4 * t4 + e;
            const int r = t / g.S, s = t - r * g.S;
            const int th = h + g.ph - r * g.dh, tw = w + g.pw - s * g.dw;
            bool ok = t < g.RS && th >= 0 && tw >= 0 && th % g.sh == 0 && tw % g.sw == 0;
            const int p = th / g.sh, q = tw / g.sw;
            ok = ok && p < g.P && q < g.Q;
            v[e] = ok ? dy[((int64_t)n * g.P + p) * g.Q + q] : 0.f;
        }
        *reinterpret_cast<float4 *>(du + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

static int tap_geom(const mqConvDesc *d, int T, TapGeom &g, const char *who)
{
    MQ_REQUIRE(d != nullptr, "%s: null descriptor", who);
    MQ_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->P > 0 && d->Q > 0 && d->R > 0 && d->S > 0 && d->groups > 0 &&
                   d->N % d->groups == 0,
               "%s: bad dimensions", who);
    MQ_REQUIRE(d->stride_h > 0 && d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0 && d->pad_h >= 0 && d->pad_w >= 0,
               "%s: bad stride/dilation/padding", who);
    MQ_REQUIRE(T % 4 == 0 && T >= d->R * d->S, "%s: T=%d must be a multiple of 4 and >= R*S=%d", who, T, d->R * d->S);
    g.N = d->N; g.H = d->H; g.W = d->W; g.P = d->P; g.Q = d->Q; g.R = d->R; g.S = d->S;
    g.sh = d->stride_h; g.sw = d->stride_w; g.ph = d->pad_h; g.pw = d->pad_w; g.dh = d->dil_h; g.dw = d->dil_w;
    g.T = T; g.RS = d->R * d->S; g.n_per_group = d->N / d->groups;
    return 0;
}

}  // namespace mq

using namespace mq;

extern "C" int mq_taps_gather(const mqConvDesc *d, int T, const float *img, float *cols, mqStream stream)
{
    TapGeom g;
    if (tap_geom(d, T, g, "mq_taps_gather")) return 1;
    MQ_REQUIRE(img && cols && (uintptr_t)cols % 16 == 0, "mq_taps_gather: null or unaligned pointer");
    const int64_t total = (int64_t)g.N * g.P * g.Q * (T / 4);
    hipLaunchKernelGGL(taps_gather_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, S(stream), g, img, cols, total);
    MQ_LAUNCH_CHECK("mq_taps_gather");
    return 0;
}

extern "C" int mq_taps_scatter(const mqConvDesc *d, int T, const float *cols, float *img, mqStream stream)
{
    TapGeom g;
    if (tap_geom(d, T, g, "mq_taps_scatter")) return 1;
    MQ_REQUIRE(img && cols, "mq_taps_scatter: null pointer");
    const int64_t total = (int64_t)g.N * g.H * g.W;
    hipLaunchKernelGGL(taps_scatter_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, S(stream), g, cols, img, total);
    MQ_LAUNCH_CHECK("mq_taps_scatter");
    return 0;
}

extern "C" int mq_taps_reduce(const mqConvDesc *d, int T, const float *u, const float *bias, float *y, mqStream stream)
{
    TapGeom g;
    if (tap_geom(d, T, g, "mq_taps_reduce")) return 1;
    MQ_REQUIRE(u && y, "mq_taps_reduce: null pointer");
    const int64_t total = (int64_t)g.N * g.P * g.Q;
    hipLaunchKernelGGL(taps_reduce_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, S(stream), g, u, bias, y, total);
    MQ_LAUNCH_CHECK("mq_taps_reduce");
    return 0;
}

extern "C" int mq_taps_expand(const mqConvDesc *d, int T, const float *dy, float *du, mqStream stream)
{
    TapGeom g;
    if (tap_geom(d, T, g, "mq_taps_expand")) return 1;
    MQ_REQUIRE(dy && du && (uintptr_t)du % 16 == 0, "mq_taps_expand: null or unaligned pointer");
    const int64_t total = (int64_t)g.N * g.H * g.W * (T / 4);
    hipLaunchKernelGGL(taps_expand_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, S(stream), g, dy, du, total);
    MQ_LAUNCH_CHECK("mq_taps_expand");
    return 0;
}
