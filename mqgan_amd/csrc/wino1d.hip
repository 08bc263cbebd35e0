// One-dimensional Winograd F(2, 3) along the image ROWS for convolutions with 3 taps and stride 1 in that direction and
// any taps / stride 1 or 2 along the columns: the multi-bin discriminator's (3 x k) layers (reference
// discriminators.py:251-290: kernels (3, k), strides (1, 1) then (1, 2), on 10-bin x T-frame band images).
//
//   rows 2t, 2t+1 of the output from input rows 2t-1 .. 2t+2 (d0..d3; zero outside the image: pad 1):
//     V = B^T d :  v0 = d0 - d2,  v1 = d1 + d2,  v2 = d2 - d1,  v3 = d1 - d3
//     U = G g   :  u0 = g0,  u1 = (g0 + g1 + g2) / 2,  u2 = (g0 - g1 + g2) / 2,  u3 = g2        (per filter column s)
//     M_xi      =  conv_{1 x S}(V_xi, U_xi)    xi = 0..3: four independent ROW convolutions (all column taps, all channels)
//     Y = A^T M :  y0 = m0 + m1 + m2,  y1 = m1 - m2 - m3
//
// 4 row products per 2 output rows instead of 6: two thirds of the multiplications.  The four positions run as ONE
// batch-grouped convolution on the implicit-GEMM kernels of igemm.hip (images = (xi, n, tile) rows of 1 x W, groups =
// 4 x the layer's groups, filters U): exact f32 products, f32 accumulation, the same strided-dgrad / split-K kernels.
// This file holds the HBM-bound transforms; channels-last f32, 4 channels per thread.
//
// Backward is the exact adjoint:  Z = A dY (z0 = y0, z1 = y0 + y1, z2 = y0 - y1, z3 = -y1);  dV_xi = dgrad(Z_xi, U_xi);
// dx = B dV (each input row collects from the one or two tiles that contain it);  dU_xi = wgrad(V_xi, Z_xi);  dg = G^T dU.
// The discriminators' epilogues ride on the transforms: bias + LeakyReLU + frame mask in the output transform, the producer
// layer's LeakyReLU' + mask (ops.conv_leaky's "gated" contract) in the dx transform, the bias gradient (deterministic
// per-block column sums) in the Z transform.
#include "common.h"

namespace mq {

__device__ __forceinline__ float4 w1_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 w1_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 w1_neg(float4 a) { return make_float4(-a.x, -a.y, -a.z, -a.w); }
__device__ __forceinline__ float4 w1_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// V[xi][n][t][w][c]
__global__ __launch_bounds__(256) void wino1d_input_kernel(const float *__restrict__ x, int64_t N, int H, int W, int C4, int th,
                                                           float *__restrict__ V)
{
    const int64_t plane = N * th * W * C4;                  // float4 elements of one xi plane
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    float4 *V4 = reinterpret_cast<float4 *>(V);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t r = i / C4;
        const int w = (int)(r % W);
        r /= W;
        const int t = (int)(r % th);
        const int64_t n = r / th;
        float4 d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int h = 2 * t - 1 + j;
            d[j] = (h >= 0 && h < H) ? x4[((n * H + h) * W + w) * C4 + c4] : w1_zero();
        }
        V4[i] = w1_sub(d[0], d[2]);
        V4[plane + i] = w1_add(d[1], d[2]);
        V4[2 * plane + i] = w1_sub(d[2], d[1]);
        V4[3 * plane + i] = w1_sub(d[1], d[3]);
    }
}

// y = epilogue(A^T M + bias).  M[xi][n][t][q][k]; act 0: none, 2: LeakyReLU(slope) then zero where padded[(m / d1) * d2 + m % d3]
// (m = output pixel index), as mq_conv_fprop_leaky.
__global__ __launch_bounds__(256) void wino1d_output_kernel(const float *__restrict__ M, const float *__restrict__ bias, int64_t N, int H,
                                                            int Q, int K4, int th, int groups, int act, float slope,
                                                            const uint8_t *__restrict__ padded, int64_t d1, int64_t d2, int64_t d3,
                                                            float *__restrict__ y)
{
    const int64_t plane = N * th * Q * K4;
    const float4 *M4 = reinterpret_cast<const float4 *>(M);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    const int64_t per_group = N / groups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (int64_t)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % K4);
        int64_t r = i / K4;
        const int q = (int)(r % Q);
        r /= Q;
        const int t = (int)(r % th);
        const int64_t n = r / th;
        const float4 m0 = M4[i], m1 = M4[plane + i], m2 = M4[2 * plane + i], m3 = M4[3 * plane + i];
        const float4 bv = bias ? reinterpret_cast<const float4 *>(bias)[(n / per_group) * K4 + k4] : w1_zero();
        float4 o[2] = {w1_add(w1_add(w1_add(m0, m1), m2), bv), w1_add(w1_sub(w1_sub(m1, m2), m3), bv)};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int h = 2 * t + a;
            if (h >= H) continue;
            const int64_t pix = (n * H + h) * Q + q;
            float4 v = o[a];
            if (act == 2) {
                v.x = v.x > 0.f ? v.x : v.x * slope;
                v.y = v.y > 0.f ? v.y : v.y * slope;
                v.z = v.z > 0.f ? v.z : v.z * slope;
                v.w = v.w > 0.f ? v.w : v.w * slope;
                if (padded && padded[(pix / d1) * d2 + pix % d3]) v = w1_zero();
            }
            y4[pix * K4 + k4] = v;
        }
    }
}

// Z[xi][n][t][q][k] = A dY, plus per-block column sums of dy for the bias gradient.  grid (blocks, groups): block (b, g)
// walks group g's rows; the launch keeps (blocks * 256) % K4 == 0, so a thread owns one channel group for the whole loop.
__global__ __launch_bounds__(256) void wino1d_dy_kernel(const float *__restrict__ dy, int64_t N, int H, int Q, int K4, int th, int groups,
                                                        float *__restrict__ Z, float *__restrict__ partial)
{
    __shared__ float4 s_sum[256];
    const int64_t per_group = N / groups;
    const int g = blockIdx.y;
    const int64_t plane = N * th * Q * K4;
    const int64_t gsize = per_group * th * Q * K4, gbase = (int64_t)g * gsize;
    const float4 *y4 = reinterpret_cast<const float4 *>(dy);
    float4 *Z4 = reinterpret_cast<float4 *>(Z);
    float4 bsum = w1_zero();
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < gsize; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = gbase + j;
        const int k4 = (int)(i % K4);
        int64_t r = i / K4;
        const int q = (int)(r % Q);
        r /= Q;
        const int t = (int)(r % th);
        const int64_t n = r / th;
        const int h0 = 2 * t, h1 = 2 * t + 1;
        const float4 y0 = y4[((n * H + h0) * Q + q) * K4 + k4];
        const float4 y1 = h1 < H ? y4[((n * H + h1) * Q + q) * K4 + k4] : w1_zero();
        bsum = w1_add(bsum, w1_add(y0, y1));
        Z4[i] = y0;
        Z4[plane + i] = w1_add(y0, y1);
        Z4[2 * plane + i] = w1_sub(y0, y1);
        Z4[3 * plane + i] = w1_neg(y1);
    }
    if (partial == nullptr) return;
    s_sum[threadIdx.x] = bsum;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < K4) {
        float4 acc = s_sum[t];
        for (int j = t + K4; j < 256; j += K4) acc = w1_add(acc, s_sum[j]);
        // gsize and gbase are multiples of K4, so the channel group of element gbase + b * 256 + t is (b * 256 + t) % K4
        const int k4 = (int)(((int64_t)blockIdx.x * 256 + t) % K4);
        reinterpret_cast<float4 *>(partial)[((int64_t)blockIdx.x * gridDim.y + g) * K4 + k4] = acc;      // [block][group][K]
    }
}

// dx = B dV, optionally gated: dx *= gate > 0 ? 1 : slope, zero where padded_in[(m / d1) * d2 + m % d3] (m = input pixel)
__global__ __launch_bounds__(256) void wino1d_dx_kernel(const float *__restrict__ dV, int64_t N, int H, int W, int C4, int th,
                                                        const float *__restrict__ gate, float slope, const uint8_t *__restrict__ padded,
                                                        int64_t d1, int64_t d2, int64_t d3, float *__restrict__ dx)
{
    const int64_t plane = N * th * W * C4;
    const int64_t total = N * H * W * C4;
    const float4 *V4 = reinterpret_cast<const float4 *>(dV);
    float4 *x4 = reinterpret_cast<float4 *>(dx);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t r = i / C4;
        const int w = (int)(r % W);
        r /= W;
        const int h = (int)(r % H);
        const int64_t n = r / H;
        const int t = h >> 1;
        const int64_t e = (((n * th + t) * W) + w) * C4 + c4;               // element of tile t inside a plane
        const int64_t trow = (int64_t)W * C4;                               // distance between consecutive tiles
        float4 v;
        if ((h & 1) == 0) {                                                 // d1 of tile t, d3 of tile t - 1
            v = w1_add(w1_sub(V4[plane + e], V4[2 * plane + e]), V4[3 * plane + e]);
            if (t > 0) v = w1_sub(v, V4[3 * plane + e - trow]);
        } else {                                                            // d2 of tile t, d0 of tile t + 1
            v = w1_add(w1_sub(V4[plane + e], V4[e]), V4[2 * plane + e]);
            if (t + 1 < th) v = w1_add(v, V4[e + trow]);
        }
        if (gate) {
            const float4 gv = reinterpret_cast<const float4 *>(gate)[i];
            v.x *= gv.x > 0.f ? 1.f : slope;
            v.y *= gv.y > 0.f ? 1.f : slope;
            v.z *= gv.z > 0.f ? 1.f : slope;
            v.w *= gv.w > 0.f ? 1.f : slope;
            const int64_t pix = (n * H + h) * W + w;
            if (padded && padded[(pix / d1) * d2 + pix % d3]) v = w1_zero();
        }
        x4[i] = v;
    }
}

// U = G g per filter column.  w is KRSC (GK, 3 * S, C) in the channel-blocked order of common.h (GK = groups * K).
// U_krsc (4 * GK, S, C) for fprop / wgrad; U_crsk (4 * groups * C, S, K), K-blocked, for dgrad (may be null).
__global__ __launch_bounds__(256) void wino1d_weight_kernel(const float *__restrict__ w, int GK, int K, int C, int S,
                                                            float *__restrict__ U_krsc, float *__restrict__ U_crsk)
{
    const int64_t total = (int64_t)GK * S * C;
    const int cbc = chan_block(C), cbk = chan_block(K);
    const int groups = GK / K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int s = (int)((i / C) % S);
        const int gk = (int)(i / ((int64_t)C * S));
        const float *wr = w + (size_t)gk * 3 * S * C;
        const float g0 = wr[kidx(0 * S + s, c, 3 * S, cbc)], g1 = wr[kidx(1 * S + s, c, 3 * S, cbc)], g2 = wr[kidx(2 * S + s, c, 3 * S, cbc)];
        const float u[4] = {g0, 0.5f * ((g0 + g1) + g2), 0.5f * ((g0 - g1) + g2), g2};
        const int grp = gk / K, k = gk - grp * K;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            U_krsc[((size_t)xi * GK + gk) * S * C + kidx(s, c, S, cbc)] = u[xi];
            if (U_crsk) U_crsk[(((size_t)xi * groups + grp) * C + c) * S * K + kidx(s, k, S, cbk)] = u[xi];
        }
    }
}

// dw = G^T dU:  dU (4 * GK, S, C) -> dw KRSC (GK, 3 * S, C), channel-blocked
__global__ __launch_bounds__(256) void wino1d_wgrad_inverse_kernel(const float *__restrict__ dU, int GK, int C, int S, float *__restrict__ dw)
{
    const int64_t total = (int64_t)GK * S * C;
    const int cbc = chan_block(C);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int s = (int)((i / C) % S);
        const int gk = (int)(i / ((int64_t)C * S));
        float p[4];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) p[xi] = dU[((size_t)xi * GK + gk) * S * C + kidx(s, c, S, cbc)];
        float *o = dw + (size_t)gk * 3 * S * C;
        o[kidx(0 * S + s, c, 3 * S, cbc)] = p[0] + 0.5f * (p[1] + p[2]);
        o[kidx(1 * S + s, c, 3 * S, cbc)] = 0.5f * (p[1] - p[2]);
        o[kidx(2 * S + s, c, 3 * S, cbc)] = 0.5f * (p[1] + p[2]) + p[3];
    }
}

static int w1_blocks_for(int64_t work_items, int K4)
{
    int g = 256, a = K4;
    while (a) { const int t = g % a; g = a; a = t; }
    const int m = K4 / g;
    int64_t want = ew_grid(work_items, 256);
    want = (want / m) * m;
    if (want < m) want = m;
    return (int)want;
}

}  // namespace mq

using namespace mq;

static int w1_args(const void *a, const void *b, int N, int H, int W, int C, const char *who)
{
    MQ_REQUIRE(a && b && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "%s: bad arguments (channels must be a multiple of 4, got %d)", who, C);
    MQ_REQUIRE(((uintptr_t)a | (uintptr_t)b) % 16 == 0, "%s: tensors must be 16-byte aligned", who);
    return 0;
}

extern "C" int mq_wino1d_input_transform(const float *x, int N, int H, int W, int C, float *V, mqStream stream)
{
    if (w1_args(x, V, N, H, W, C, "mq_wino1d_input_transform")) return 1;
    const int th = (H + 1) / 2;
    hipLaunchKernelGGL(wino1d_input_kernel, dim3(ew_grid((int64_t)N * th * W * (C / 4), 256)), dim3(256), 0, S(stream), x, (int64_t)N, H, W,
                       C / 4, th, V);
    MQ_LAUNCH_CHECK("mq_wino1d_input_transform");
    return 0;
}

extern "C" int mq_wino1d_output_transform(const float *M, const float *bias, int N, int H, int Q, int K, int groups, int act, float slope,
                                          const uint8_t *padded, int64_t d1, int64_t d2, int64_t d3, float *y, mqStream stream)
{
    if (w1_args(M, y, N, H, Q, K, "mq_wino1d_output_transform")) return 1;
    MQ_REQUIRE(groups > 0 && N % groups == 0 && (act == 0 || act == 2), "mq_wino1d_output_transform: groups must divide N, act 0 or 2");
    MQ_REQUIRE(!bias || (uintptr_t)bias % 16 == 0, "mq_wino1d_output_transform: bias must be 16-byte aligned");
    MQ_REQUIRE(!padded || (d1 > 0 && d3 > 0), "mq_wino1d_output_transform: mask index map");
    const int th = (H + 1) / 2;
    hipLaunchKernelGGL(wino1d_output_kernel, dim3(ew_grid((int64_t)N * th * Q * (K / 4), 256)), dim3(256), 0, S(stream), M, bias, (int64_t)N,
                       H, Q, K / 4, th, groups, act, slope, padded, d1, d2, d3, y);
    MQ_LAUNCH_CHECK("mq_wino1d_output_transform");
    return 0;
}

extern "C" int mq_wino1d_dy_rows(int N, int H, int Q, int K, int groups)
{
    if (N <= 0 || H <= 0 || Q <= 0 || K <= 0 || K % 4 != 0 || K > 1024 || groups <= 0 || N % groups != 0) return 0;
    return w1_blocks_for((int64_t)(N / groups) * ((H + 1) / 2) * Q * (K / 4), K / 4);
}

extern "C" int mq_wino1d_dy_transform(const float *dy, int N, int H, int Q, int K, int groups, float *Z, float *bias_partial, mqStream stream)
{
    if (w1_args(dy, Z, N, H, Q, K, "mq_wino1d_dy_transform")) return 1;
    const int blocks = mq_wino1d_dy_rows(N, H, Q, K, groups);
    MQ_REQUIRE(blocks > 0, "mq_wino1d_dy_transform: shape not supported (K <= 1024, groups must divide N)");
    MQ_REQUIRE(!bias_partial || (uintptr_t)bias_partial % 16 == 0, "mq_wino1d_dy_transform: bias_partial must be 16-byte aligned");
    const int th = (H + 1) / 2;
    hipLaunchKernelGGL(wino1d_dy_kernel, dim3(blocks, groups), dim3(256), 0, S(stream), dy, (int64_t)N, H, Q, K / 4, th, groups, Z,
                       bias_partial);
    MQ_LAUNCH_CHECK("mq_wino1d_dy_transform");
    return 0;
}

extern "C" int mq_wino1d_dx_transform(const float *dV, int N, int H, int W, int C, const float *gate, float slope, const uint8_t *padded,
                                      int64_t d1, int64_t d2, int64_t d3, float *dx, mqStream stream)
{
    if (w1_args(dV, dx, N, H, W, C, "mq_wino1d_dx_transform")) return 1;
    MQ_REQUIRE(!gate || (uintptr_t)gate % 16 == 0, "mq_wino1d_dx_transform: gate must be 16-byte aligned");
    MQ_REQUIRE(!padded || (gate && d1 > 0 && d3 > 0), "mq_wino1d_dx_transform: a mask needs the gate tensor and an index map");
    const int th = (H + 1) / 2;
    hipLaunchKernelGGL(wino1d_dx_kernel, dim3(ew_grid((int64_t)N * H * W * (C / 4), 256)), dim3(256), 0, S(stream), dV, (int64_t)N, H, W,
                       C / 4, th, gate, slope, padded, d1, d2, d3, dx);
    MQ_LAUNCH_CHECK("mq_wino1d_dx_transform");
    return 0;
}

extern "C" int mq_wino1d_weight_transform(const float *w_krsc, int groups, int K, int C, int Sw, float *U_krsc, float *U_crsk, mqStream stream)
{
    MQ_REQUIRE(w_krsc && U_krsc && groups > 0 && K > 0 && C > 0 && Sw > 0, "mq_wino1d_weight_transform: bad arguments");
    hipLaunchKernelGGL(wino1d_weight_kernel, dim3(ew_grid((int64_t)groups * K * Sw * C, 256)), dim3(256), 0, S(stream), w_krsc, groups * K, K, C,
                       Sw, U_krsc, U_crsk);
    MQ_LAUNCH_CHECK("mq_wino1d_weight_transform");
    return 0;
}

extern "C" int mq_wino1d_wgrad_inverse(const float *dU, int groups, int K, int C, int Sw, float *dw_krsc, mqStream stream)
{
    MQ_REQUIRE(dU && dw_krsc && groups > 0 && K > 0 && C > 0 && Sw > 0, "mq_wino1d_wgrad_inverse: bad arguments");
    hipLaunchKernelGGL(wino1d_wgrad_inverse_kernel, dim3(ew_grid((int64_t)groups * K * Sw * C, 256)), dim3(256), 0, S(stream), dU, groups * K, C,
                       Sw, dw_krsc);
    MQ_LAUNCH_CHECK("mq_wino1d_wgrad_inverse");
    return 0;
}
