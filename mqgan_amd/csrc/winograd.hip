// Winograd F(2x2, 3x3) for the refiner's 3x3 / stride-1 / pad-1 convolutions (reference preencoder.py:86-102, nn.Conv2d).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d: 4x4 input tile, g: 3x3 filter, Y: 2x2 outputs
//
// 16 multiplications per 4 outputs instead of 36: the 16 transform positions are 16 independent GEMMs
// [tiles x C] x [C x K], which run as ONE batch-grouped 1x1 convolution (groups = 16) on the existing implicit-GEMM
// kernels (igemm.hip) - f32 MFMA with exact products, f32 accumulation.  This file holds the HBM-bound transforms around
// that GEMM; everything is f32, channels-last, 4 channels per thread.
//
//   forward   V = B^T d B  ->  M_xi = V_xi U_xi (U = G g G^T)  ->  Y = A^T M A (+ bias)
//   dgrad     the same with dy as input and the filter rotated by 180 degrees and transposed (a "same" convolution)
//   wgrad     dg = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G: V as above, Z = A dY A^T, P_xi = Z_xi^T V_xi on the
//             weight-gradient GEMM kernels (groups = 16), then the 4x4 -> 3x3 inverse transform
//
// Tile (n, ti, tj) covers outputs (2ti .. 2ti+1, 2tj .. 2tj+1) and inputs (2ti-1 .. 2ti+2, 2tj-1 .. 2tj+2); everything
// outside the image is zero (pad 1), odd heights / widths leave the last tile half used.
#include "common.h"

namespace mq {

__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator-(float4 a) { return make_float4(-a.x, -a.y, -a.z, -a.w); }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// V[xi][tile][c] = (B^T d B)[xi],  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
__global__ __launch_bounds__(256) void wino_input_kernel(const float *__restrict__ x, int64_t N, int H, int W, int C4, int th, int tw,
                                                         float *__restrict__ V)
{
    const int64_t tiles = N * th * tw;
    const int64_t total = tiles * C4;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    float4 *V4 = reinterpret_cast<float4 *>(V);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const int64_t tile = i / C4;
        const int tj = (int)(tile % tw);
        const int64_t r = tile / tw;
        const int ti = (int)(r % th);
        const int64_t n = r / th;
        float4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int h = 2 * ti - 1 + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int w = 2 * tj - 1 + b;
                d[a][b] = (h >= 0 && h < H && w >= 0 && w < W) ? x4[((n * H + h) * W + w) * C4 + c4] : f4zero();
            }
        }
        float4 t[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            t[0][b] = d[0][b] - d[2][b];
            t[1][b] = d[1][b] + d[2][b];
            t[2][b] = d[2][b] - d[1][b];
            t[3][b] = d[1][b] - d[3][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 v0 = t[a][0] - t[a][2], v1 = t[a][1] + t[a][2], v2 = t[a][2] - t[a][1], v3 = t[a][1] - t[a][3];
            V4[((int64_t)(4 * a + 0) * tiles + tile) * C4 + c4] = v0;
            V4[((int64_t)(4 * a + 1) * tiles + tile) * C4 + c4] = v1;
            V4[((int64_t)(4 * a + 2) * tiles + tile) * C4 + c4] = v2;
            V4[((int64_t)(4 * a + 3) * tiles + tile) * C4 + c4] = v3;
        }
    }
}

// Z[xi][tile][k] = (A dY A^T)[xi],  A = [1 0; 1 1; 1 -1; 0 -1]   (the output-gradient side of the weight gradient)
__global__ __launch_bounds__(256) void wino_dy_kernel(const float *__restrict__ dy, int64_t N, int H, int W, int K4, int th, int tw,
                                                      float *__restrict__ Z)
{
    const int64_t tiles = N * th * tw;
    const int64_t total = tiles * K4;
    const float4 *y4 = reinterpret_cast<const float4 *>(dy);
    float4 *Z4 = reinterpret_cast<float4 *>(Z);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % K4);
        const int64_t tile = i / K4;
        const int tj = (int)(tile % tw);
        const int64_t r = tile / tw;
        const int ti = (int)(r % th);
        const int64_t n = r / th;
        float4 y[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int h = 2 * ti + a, w = 2 * tj + b;
                y[a][b] = (h < H && w < W) ? y4[((n * H + h) * W + w) * K4 + k4] : f4zero();
            }
        float4 t[4][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            t[0][b] = y[0][b];
            t[1][b] = y[0][b] + y[1][b];
            t[2][b] = y[0][b] - y[1][b];
            t[3][b] = -y[1][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            Z4[((int64_t)(4 * a + 0) * tiles + tile) * K4 + k4] = t[a][0];
            Z4[((int64_t)(4 * a + 1) * tiles + tile) * K4 + k4] = t[a][0] + t[a][1];
            Z4[((int64_t)(4 * a + 2) * tiles + tile) * K4 + k4] = t[a][0] - t[a][1];
            Z4[((int64_t)(4 * a + 3) * tiles + tile) * K4 + k4] = -t[a][1];
        }
    }
}

// Backward of a Winograd layer reads dy for three things: Z = A dY A^T (weight gradient), V' = B^T dy B (input gradient: the
// "same" convolution of dy with the rotated filter) and the bias gradient (column sums).  One pass: the 4x4 window of a tile
// holds its 2x2 outputs in the middle.  The column sums are deterministic: the launch is sized so that a thread keeps its
// channel group over the whole grid-stride loop (threads % K4 == 0), sums it in registers, the block adds its threads' sums
// in a fixed order and writes one row of partial[blocks][K]; the caller column-sums that small matrix.
__global__ __launch_bounds__(256) void wino_dy_fused_kernel(const float *__restrict__ dy, int64_t N, int H, int W, int K4, int th, int tw,
                                                            float *__restrict__ Z, float *__restrict__ V, float *__restrict__ partial)
{
    __shared__ float4 s_sum[256];
    const int64_t tiles = N * th * tw;
    const int64_t total = tiles * K4;
    const float4 *y4 = reinterpret_cast<const float4 *>(dy);
    float4 *Z4 = reinterpret_cast<float4 *>(Z);
    float4 *V4 = reinterpret_cast<float4 *>(V);
    float4 bsum = f4zero();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % K4);
        const int64_t tile = i / K4;
        const int tj = (int)(tile % tw);
        const int64_t r = tile / tw;
        const int ti = (int)(r % th);
        const int64_t n = r / th;
        float4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int h = 2 * ti - 1 + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int w = 2 * tj - 1 + b;
                d[a][b] = (h >= 0 && h < H && w >= 0 && w < W) ? y4[((n * H + h) * W + w) * K4 + k4] : f4zero();
            }
        }
        bsum = bsum + ((d[1][1] + d[1][2]) + (d[2][1] + d[2][2]));
        {   // Z = A y A^T of the middle 2x2 (wino_dy_kernel)
            float4 t[4][2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                t[0][b] = d[1][1 + b];
                t[1][b] = d[1][1 + b] + d[2][1 + b];
                t[2][b] = d[1][1 + b] - d[2][1 + b];
                t[3][b] = -d[2][1 + b];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                Z4[((int64_t)(4 * a + 0) * tiles + tile) * K4 + k4] = t[a][0];
                Z4[((int64_t)(4 * a + 1) * tiles + tile) * K4 + k4] = t[a][0] + t[a][1];
                Z4[((int64_t)(4 * a + 2) * tiles + tile) * K4 + k4] = t[a][0] - t[a][1];
                Z4[((int64_t)(4 * a + 3) * tiles + tile) * K4 + k4] = -t[a][1];
            }
        }
        {   // V' = B^T d B (wino_input_kernel)
            float4 t[4][4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                t[0][b] = d[0][b] - d[2][b];
                t[1][b] = d[1][b] + d[2][b];
                t[2][b] = d[2][b] - d[1][b];
                t[3][b] = d[1][b] - d[3][b];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                V4[((int64_t)(4 * a + 0) * tiles + tile) * K4 + k4] = t[a][0] - t[a][2];
                V4[((int64_t)(4 * a + 1) * tiles + tile) * K4 + k4] = t[a][1] + t[a][2];
                V4[((int64_t)(4 * a + 2) * tiles + tile) * K4 + k4] = t[a][2] - t[a][1];
                V4[((int64_t)(4 * a + 3) * tiles + tile) * K4 + k4] = t[a][1] - t[a][3];
            }
        }
    }
    // block sums per channel group: thread t owns group (blockIdx * 256 + t) % K4 for the whole loop
    s_sum[threadIdx.x] = bsum;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < K4) {                                   // first member of its group inside this block (K4 >= 256: every thread)
        float4 acc = s_sum[t];
        for (int j = t + K4; j < 256; j += K4) acc = acc + s_sum[j];
        const int k4 = (int)(((int64_t)blockIdx.x * 256 + t) % K4);
        reinterpret_cast<float4 *>(partial)[(int64_t)blockIdx.x * K4 + k4] = acc;
    }
}

// y = A^T M A (+ bias),  A^T = [1 1 1 0; 0 1 -1 -1]
__global__ __launch_bounds__(256) void wino_output_kernel(const float *__restrict__ M, const float *__restrict__ bias, int64_t N, int H,
                                                          int W, int K4, int th, int tw, float *__restrict__ y)
{
    const int64_t tiles = N * th * tw;
    const int64_t total = tiles * K4;
    const float4 *M4 = reinterpret_cast<const float4 *>(M);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % K4);
        const int64_t tile = i / K4;
        const int tj = (int)(tile % tw);
        const int64_t r = tile / tw;
        const int ti = (int)(r % th);
        const int64_t n = r / th;
        float4 m[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) m[a][b] = M4[((int64_t)(4 * a + b) * tiles + tile) * K4 + k4];
        float4 s[2][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            s[0][b] = (m[0][b] + m[1][b]) + m[2][b];
            s[1][b] = (m[1][b] - m[2][b]) - m[3][b];
        }
        const float4 bv = bias ? reinterpret_cast<const float4 *>(bias)[k4] : f4zero();
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int h = 2 * ti + a;
            if (h >= H) continue;
            const float4 o0 = ((s[a][0] + s[a][1]) + s[a][2]) + bv, o1 = ((s[a][1] - s[a][2]) - s[a][3]) + bv;
            const int w = 2 * tj;
            y4[((n * H + h) * W + w) * K4 + k4] = o0;
            if (w + 1 < W) y4[((n * H + h) * W + w + 1) * K4 + k4] = o1;
        }
    }
}

// U = G g G^T,  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1].  w is KRSC (K, 9, C) in the channel-blocked order of common.h.
// flip = 0: U[xi][k][c] (forward);  flip = 1: U[xi][c][k] of the filter rotated by 180 degrees (input gradient)
__global__ __launch_bounds__(256) void wino_weight_kernel(const float *__restrict__ w, int K, int C, int flip, float *__restrict__ U)
{
    const int64_t total = (int64_t)K * C;
    const int cb = chan_block(C);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), k = (int)(i / C);
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int rs = flip ? (2 - r) * 3 + (2 - s) : r * 3 + s;
                g[r][s] = w[(size_t)k * 9 * C + kidx(rs, c, 9, cb)];
            }
        float t[4][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            t[0][s] = g[0][s];
            t[1][s] = 0.5f * ((g[0][s] + g[1][s]) + g[2][s]);
            t[2][s] = 0.5f * ((g[0][s] - g[1][s]) + g[2][s]);
            t[3][s] = g[2][s];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float u[4] = {t[a][0], 0.5f * ((t[a][0] + t[a][1]) + t[a][2]), 0.5f * ((t[a][0] - t[a][1]) + t[a][2]), t[a][2]};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xi = 4 * a + b;
                if (flip) U[((size_t)xi * C + c) * K + k] = u[b];
                else U[((size_t)xi * K + k) * C + c] = u[b];
            }
        }
    }
}

// dw = G^T P G per (k, c):  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1];  P[xi][k][c] -> KRSC (K, 9, C), channel-blocked
__global__ __launch_bounds__(256) void wino_wgrad_inverse_kernel(const float *__restrict__ P, int K, int C, float *__restrict__ dw)
{
    const int64_t total = (int64_t)K * C;
    const int cb = chan_block(C);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), k = (int)(i / C);
        float p[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) p[a][b] = P[((size_t)(4 * a + b) * K + k) * C + c];
        float q[3][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            q[0][b] = p[0][b] + 0.5f * (p[1][b] + p[2][b]);
            q[1][b] = 0.5f * (p[1][b] - p[2][b]);
            q[2][b] = 0.5f * (p[1][b] + p[2][b]) + p[3][b];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float o[3] = {q[r][0] + 0.5f * (q[r][1] + q[r][2]), 0.5f * (q[r][1] - q[r][2]), 0.5f * (q[r][1] + q[r][2]) + q[r][3]};
#pragma unroll
            for (int s = 0; s < 3; ++s) dw[(size_t)k * 9 * C + kidx(r * 3 + s, c, 9, cb)] = o[s];
        }
    }
}

}  // namespace mq

using namespace mq;

static int wino_args(const void *a, const void *b, int N, int H, int W, int C, const char *who)
{
    MQ_REQUIRE(a && b && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "%s: bad arguments (channels must be a multiple of 4, got %d)", who, C);
    MQ_REQUIRE(((uintptr_t)a | (uintptr_t)b) % 16 == 0, "%s: tensors must be 16-byte aligned", who);
    return 0;
}

extern "C" int mq_wino_input_transform(const float *x, int N, int H, int W, int C, float *V, mqStream stream)
{
    if (wino_args(x, V, N, H, W, C, "mq_wino_input_transform")) return 1;
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    hipLaunchKernelGGL(wino_input_kernel, dim3(ew_grid((int64_t)N * th * tw * (C / 4), 256)), dim3(256), 0, S(stream), x, (int64_t)N, H, W,
                       C / 4, th, tw, V);
    MQ_LAUNCH_CHECK("mq_wino_input_transform");
    return 0;
}
extern "C" int mq_wino_dy_transform(const float *dy, int N, int H, int W, int K, float *Z, mqStream stream)
{
    if (wino_args(dy, Z, N, H, W, K, "mq_wino_dy_transform")) return 1;
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    hipLaunchKernelGGL(wino_dy_kernel, dim3(ew_grid((int64_t)N * th * tw * (K / 4), 256)), dim3(256), 0, S(stream), dy, (int64_t)N, H, W,
                       K / 4, th, tw, Z);
    MQ_LAUNCH_CHECK("mq_wino_dy_transform");
    return 0;
}
// blocks of mq_wino_dy_fused: enough to fill the chip, a multiple of K4 / gcd(K4, 256) so that (blocks * 256) % K4 == 0, and -
// for K4 > 256 - at least K4 / 256 rounded up to that multiple (every channel group needs a thread)
static int wino_fused_blocks(int64_t work_items, int K4)
{
    int g = 256, a = K4;
    while (a) { const int t = g % a; g = a; a = t; }          // gcd(256, K4)
    const int m = K4 / g;
    int64_t want = ew_grid(work_items, 256);
    want = (want / m) * m;
    if (want < m) want = m;
    return (int)want;
}
extern "C" int mq_wino_dy_fused_rows(int N, int H, int W, int K)
{
    if (N <= 0 || H <= 0 || W <= 0 || K <= 0 || K % 4 != 0 || K > 1024) return 0;
    return wino_fused_blocks((int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (K / 4), K / 4);
}
extern "C" int mq_wino_dy_fused(const float *dy, int N, int H, int W, int K, float *Z, float *Vd, float *bias_partial, mqStream stream)
{
    if (wino_args(dy, Z, N, H, W, K, "mq_wino_dy_fused")) return 1;
    MQ_REQUIRE(Vd && bias_partial && ((uintptr_t)Vd | (uintptr_t)bias_partial) % 16 == 0, "mq_wino_dy_fused: Vd / bias_partial null or unaligned");
    MQ_REQUIRE(K <= 1024, "mq_wino_dy_fused: K <= 1024 (every block must cover every channel group; got %d)", K);
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    const int blocks = mq_wino_dy_fused_rows(N, H, W, K);
    MQ_REQUIRE(blocks > 0 && ((int64_t)blocks * 256) % (K / 4) == 0, "mq_wino_dy_fused: launch plan");
    hipLaunchKernelGGL(wino_dy_fused_kernel, dim3(blocks), dim3(256), 0, S(stream), dy, (int64_t)N, H, W, K / 4, th, tw, Z, Vd, bias_partial);
    MQ_LAUNCH_CHECK("mq_wino_dy_fused");
    return 0;
}
extern "C" int mq_wino_output_transform(const float *M, const float *bias, int N, int H, int W, int K, float *y, mqStream stream)
{
    if (wino_args(M, y, N, H, W, K, "mq_wino_output_transform")) return 1;
    MQ_REQUIRE(!bias || (uintptr_t)bias % 16 == 0, "mq_wino_output_transform: bias must be 16-byte aligned");
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    hipLaunchKernelGGL(wino_output_kernel, dim3(ew_grid((int64_t)N * th * tw * (K / 4), 256)), dim3(256), 0, S(stream), M, bias, (int64_t)N,
                       H, W, K / 4, th, tw, y);
    MQ_LAUNCH_CHECK("mq_wino_output_transform");
    return 0;
}
extern "C" int mq_wino_weight_transform(const float *w_krsc, int K, int C, int flip_transpose, float *U, mqStream stream)
{
    MQ_REQUIRE(w_krsc && U && K > 0 && C > 0, "mq_wino_weight_transform: bad arguments");
    hipLaunchKernelGGL(wino_weight_kernel, dim3(ew_grid((int64_t)K * C, 256)), dim3(256), 0, S(stream), w_krsc, K, C, flip_transpose, U);
    MQ_LAUNCH_CHECK("mq_wino_weight_transform");
    return 0;
}
extern "C" int mq_wino_wgrad_inverse(const float *P, int K, int C, float *dw_krsc, mqStream stream)
{
    MQ_REQUIRE(P && dw_krsc && K > 0 && C > 0, "mq_wino_wgrad_inverse: bad arguments");
    hipLaunchKernelGGL(wino_wgrad_inverse_kernel, dim3(ew_grid((int64_t)K * C, 256)), dim3(256), 0, S(stream), P, K, C, dw_krsc);
    MQ_LAUNCH_CHECK("mq_wino_wgrad_inverse");
    return 0;
}
