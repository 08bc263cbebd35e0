// Winograd F(3x3, 3x3) for the refiner's 3x3 / stride-1 / pad-1 convolutions (reference preencoder.py:86-102): the larger
// sibling of winograd.hip's F(2x2, 3x3).  25 multiplications per 9 outputs (2.78 per output) instead of 16 per 4 (4 per
// output) or 81 per 9 direct; the transformed tensors are 25/9 = 2.78x the activation instead of 4x.  Interpolation points
// 0, 1, -1, 2, inf:
//
//   B^T = [2 -1 -2 1 0; 0 -2 -1 1 0; 0 2 -3 1 0; 0 -1 0 1 0; 0 2 -1 -2 1]
//   G   = [1/2 0 0; -1/2 -1/2 -1/2; -1/6 1/6 -1/6; 1/6 1/3 2/3; 0 0 1]
//   A^T = [1 1 1 1 0; 0 1 -1 2 0; 0 1 1 4 1]
//
// Measured error of one layer in f32 (512 channels, numpy): 2.7e-6 of the output rms against 1.5e-6 for F(2x2, 3x3) and
// 1.7e-5 for F(4x4, 3x3) (which stays unused).  The 25 positions are ONE batch-grouped 1x1 convolution (groups = 25) on the
// implicit-GEMM kernels of igemm.hip; this file holds the HBM-bound transforms.  Everything f32, channels-last, 4 channels
// per thread.  Tile (n, ti, tj) covers outputs (3ti .. 3ti+2, 3tj .. 3tj+2) and inputs (3ti-1 .. 3ti+3, 3tj-1 .. 3tj+3).
//
//   forward   V = B^T d B -> M_xi = V_xi U_xi (U = G g G^T) -> Y = A^T M A (+ bias)
//   dgrad     the same with dy as input and the filter rotated by 180 degrees and transposed
//   wgrad     dg = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G
#include "common.h"

namespace mq {

namespace w3 {
constexpr float BT[5][5] = {{2, -1, -2, 1, 0}, {0, -2, -1, 1, 0}, {0, 2, -3, 1, 0}, {0, -1, 0, 1, 0}, {0, 2, -1, -2, 1}};
constexpr float G[5][3] = {{0.5f, 0, 0}, {-0.5f, -0.5f, -0.5f}, {-1.0f / 6, 1.0f / 6, -1.0f / 6}, {1.0f / 6, 1.0f / 3, 2.0f / 3}, {0, 0, 1}};
constexpr float AT[3][5] = {{1, 1, 1, 1, 0}, {0, 1, -1, 2, 0}, {0, 1, 1, 4, 1}};

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// acc += c * v with the constant known at compile time: zero terms vanish, +-1 become adds
__device__ __forceinline__ void axpy(float4 &acc, float c, float4 v)
{
    if (c == 0.f) return;
    if (c == 1.f) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; return; }
    if (c == -1.f) { acc.x -= v.x; acc.y -= v.y; acc.z -= v.z; acc.w -= v.w; return; }
    acc.x += c * v.x; acc.y += c * v.y; acc.z += c * v.z; acc.w += c * v.w;
}

// out (5x5) = B^T d B
__device__ __forceinline__ void input_tf(const float4 (&d)[5][5], float4 (&v)[5][5])
{
    float4 t[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            float4 a = zero4();
#pragma unroll
            for (int k = 0; k < 5; ++k) axpy(a, BT[i][k], d[k][b]);
            t[i][b] = a;
        }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            float4 a = zero4();
#pragma unroll
            for (int k = 0; k < 5; ++k) axpy(a, BT[j][k], t[i][k]);
            v[i][j] = a;
        }
}
}  // namespace w3

__device__ __forceinline__ void w3_tile_of(int64_t i, int C4, int tw, int th, int &c4, int64_t &tile, int &tj, int &ti, int64_t &n)
{
    c4 = (int)(i % C4);
    tile = i / C4;
    tj = (int)(tile % tw);
    const int64_t r = tile / tw;
    ti = (int)(r % th);
    n = r / th;
}

__device__ __forceinline__ void w3_load_window(const float4 *__restrict__ x4, int64_t n, int H, int W, int C4, int c4, int ti, int tj,
                                               float4 (&d)[5][5])
{
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        const int h = 3 * ti - 1 + a;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            const int w = 3 * tj - 1 + b;
            d[a][b] = (h >= 0 && h < H && w >= 0 && w < W) ? x4[((n * H + h) * W + w) * C4 + c4] : w3::zero4();
        }
    }
}

// V[xi][tile][c]
__global__ __launch_bounds__(256) void wino3_input_kernel(const float *__restrict__ x, int64_t N, int H, int W, int C4, int th, int tw,
                                                          float *__restrict__ V)
{
    const int64_t tiles = N * th * tw, total = tiles * C4;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    float4 *V4 = reinterpret_cast<float4 *>(V);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c4, tj, ti;
        int64_t tile, n;
        w3_tile_of(i, C4, tw, th, c4, tile, tj, ti, n);
        float4 d[5][5], v[5][5];
        w3_load_window(x4, n, H, W, C4, c4, ti, tj, d);
        w3::input_tf(d, v);
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) V4[((int64_t)(5 * a + b) * tiles + tile) * C4 + c4] = v[a][b];
    }
}

// One pass over dy for the backward of a layer: Z = A dY A^T of the tile's 3x3 outputs (the middle of the window), V' = B^T dy B
// (the input gradient's transformed operand) and deterministic per-block column sums (see wino_dy_fused_kernel).
__global__ __launch_bounds__(256) void wino3_dy_fused_kernel(const float *__restrict__ dy, int64_t N, int H, int W, int K4, int th, int tw,
                                                             float *__restrict__ Z, float *__restrict__ V, float *__restrict__ partial)
{
    __shared__ float4 s_sum[256];
    const int64_t tiles = N * th * tw, total = tiles * K4;
    const float4 *y4 = reinterpret_cast<const float4 *>(dy);
    float4 *Z4 = reinterpret_cast<float4 *>(Z);
    float4 *V4 = reinterpret_cast<float4 *>(V);
    float4 bsum = w3::zero4();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int k4, tj, ti;
        int64_t tile, n;
        w3_tile_of(i, K4, tw, th, k4, tile, tj, ti, n);
        float4 d[5][5];
        w3_load_window(y4, n, H, W, K4, k4, ti, tj, d);
        // the tile owns output rows / columns 3t .. 3t+2 = window positions 1 .. 3, where they exist (ragged last tiles)
        float4 y[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const bool in = 3 * ti + a < H && 3 * tj + b < W;
                y[a][b] = in ? d[1 + a][1 + b] : w3::zero4();
                w3::axpy(bsum, 1.f, y[a][b]);
            }
        {   // Z = A y A^T  (A = AT^T: 5x3)
            float4 t[5][3];
#pragma unroll
            for (int i5 = 0; i5 < 5; ++i5)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    float4 a = w3::zero4();
#pragma unroll
                    for (int k = 0; k < 3; ++k) w3::axpy(a, w3::AT[k][i5], y[k][b]);
                    t[i5][b] = a;
                }
#pragma unroll
            for (int i5 = 0; i5 < 5; ++i5)
#pragma unroll
                for (int j5 = 0; j5 < 5; ++j5) {
                    float4 a = w3::zero4();
#pragma unroll
                    for (int k = 0; k < 3; ++k) w3::axpy(a, w3::AT[k][j5], t[i5][k]);
                    Z4[((int64_t)(5 * i5 + j5) * tiles + tile) * K4 + k4] = a;
                }
        }
        if (V != nullptr) {
            // a tile's window may reach output positions of the NEXT tile that lie beyond a ragged edge: those are zeros already
            float4 v[5][5];
            w3::input_tf(d, v);
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int b = 0; b < 5; ++b) V4[((int64_t)(5 * a + b) * tiles + tile) * K4 + k4] = v[a][b];
        }
    }
    if (partial == nullptr) return;
    s_sum[threadIdx.x] = bsum;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < K4) {
        float4 acc = s_sum[t];
        for (int j = t + K4; j < 256; j += K4) w3::axpy(acc, 1.f, s_sum[j]);
        const int k4 = (int)(((int64_t)blockIdx.x * 256 + t) % K4);
        reinterpret_cast<float4 *>(partial)[(int64_t)blockIdx.x * K4 + k4] = acc;
    }
}

// y = epilogue(A^T M A + bias);  act 0: none, 2: LeakyReLU(slope) then zero where padded[(m / d1) * d2 + m % d3] (m = output pixel
// index), as mq_conv_fprop_leaky
__global__ __launch_bounds__(256) void wino3_output_kernel(const float *__restrict__ M, const float *__restrict__ bias, int64_t N, int H,
                                                           int W, int K4, int th, int tw, int act, float slope,
                                                           const uint8_t *__restrict__ padded, int64_t d1, int64_t d2, int64_t d3,
                                                           float *__restrict__ y)
{
    const int64_t tiles = N * th * tw, total = tiles * K4;
    const float4 *M4 = reinterpret_cast<const float4 *>(M);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int k4, tj, ti;
        int64_t tile, n;
        w3_tile_of(i, K4, tw, th, k4, tile, tj, ti, n);
        float4 m[5][5];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) m[a][b] = M4[((int64_t)(5 * a + b) * tiles + tile) * K4 + k4];
        float4 s[3][5];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                float4 acc = w3::zero4();
#pragma unroll
                for (int k = 0; k < 5; ++k) w3::axpy(acc, w3::AT[a][k], m[k][b]);
                s[a][b] = acc;
            }
        const float4 bv = bias ? reinterpret_cast<const float4 *>(bias)[k4] : w3::zero4();
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int h = 3 * ti + a;
            if (h >= H) continue;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int w = 3 * tj + b;
                if (w >= W) continue;
                float4 acc = bv;
#pragma unroll
                for (int k = 0; k < 5; ++k) w3::axpy(acc, w3::AT[b][k], s[a][k]);
                const int64_t pix = (n * H + h) * W + w;
                if (act == 2) {
                    acc.x = acc.x > 0.f ? acc.x : acc.x * slope;
                    acc.y = acc.y > 0.f ? acc.y : acc.y * slope;
                    acc.z = acc.z > 0.f ? acc.z : acc.z * slope;
                    acc.w = acc.w > 0.f ? acc.w : acc.w * slope;
                    if (padded && padded[(pix / d1) * d2 + pix % d3]) acc = w3::zero4();
                }
                y4[pix * K4 + k4] = acc;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 5x5 / stride 2 / pad 2 layers (the patch discriminator's widest, discriminators.py:86-173) as a 3x3 / stride-1 convolution
// over the four POLYPHASE components of the input: with r = 2a + rho, s = 2b + sigma
//     y[p][q] = sum_{rho, sigma, a, b} w[2a + rho][2b + sigma] * x[2(p + a - 1) + rho][2(q + b - 1) + sigma]
// i.e. X'[p][q][(rho, sigma, c)] = x[2p + rho][2q + sigma][c] (H/2 x W/2, 4C channels) convolved with the 3x3 filter
// g'[(rho, sigma, c)][a][b] = w[c][2a + rho][2b + sigma] (zero where 2a + rho or 2b + sigma is 5).  Dense that is 36 instead of
// 25 products per output and channel - but it is a 3x3 / stride-1 convolution, so F(3x3, 3x3) applies: 25/9 * 4 = 11.1.
// X' is never materialised: the input transform reads x with the polyphase indexing, the input-gradient output transform
// writes dx in place (with the producer layer's LeakyReLU' / mask gate), the weight transforms gather / scatter the 5x5 taps.
__device__ __forceinline__ void w3_load_window_s2d(const float4 *__restrict__ x4, int64_t n, int H, int W, int C4, int c4, int rho, int sigma,
                                                   int ti, int tj, float4 (&d)[5][5])
{
    const int H2 = H >> 1, W2 = W >> 1;
#pragma unroll
    for (int a = 0; a < 5; ++a) {
        const int h = 3 * ti - 1 + a;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            const int w = 3 * tj - 1 + b;
            d[a][b] = (h >= 0 && h < H2 && w >= 0 && w < W2) ? x4[((n * H + 2 * h + rho) * W + 2 * w + sigma) * C4 + c4] : w3::zero4();
        }
    }
}

// V[xi][tile][c'] with c' = (2 rho + sigma) * C + c;  tiles over the (H/2, W/2) grid
__global__ __launch_bounds__(256) void wino3_input_s2d_kernel(const float *__restrict__ x, int64_t N, int H, int W, int C4, int th, int tw,
                                                              float *__restrict__ V)
{
    const int C4p = 4 * C4;
    const int64_t tiles = N * th * tw, total = tiles * C4p;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    float4 *V4 = reinterpret_cast<float4 *>(V);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c4p, tj, ti;
        int64_t tile, n;
        w3_tile_of(i, C4p, tw, th, c4p, tile, tj, ti, n);
        const int ph = c4p / C4, c4 = c4p - ph * C4;
        float4 d[5][5], v[5][5];
        w3_load_window_s2d(x4, n, H, W, C4, c4, ph >> 1, ph & 1, ti, tj, d);
        w3::input_tf(d, v);
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) V4[((int64_t)(5 * a + b) * tiles + tile) * C4p + c4p] = v[a][b];
    }
}

// dx (N, H, W, C) from M (25, tiles, 4C): A^T M A per polyphase channel, written to its place in the full-resolution image;
// optionally gated: dx *= gate > 0 ? 1 : slope, zero where padded[(m / d1) * d2 + m % d3] (m = input pixel index)
__global__ __launch_bounds__(256) void wino3_output_d2s_kernel(const float *__restrict__ M, int64_t N, int H, int W, int C4, int th, int tw,
                                                               const float *__restrict__ gate, float slope,
                                                               const uint8_t *__restrict__ padded, int64_t d1, int64_t d2, int64_t d3,
                                                               float *__restrict__ dx)
{
    const int C4p = 4 * C4, H2 = H >> 1, W2 = W >> 1;
    const int64_t tiles = N * th * tw, total = tiles * C4p;
    const float4 *M4 = reinterpret_cast<const float4 *>(M);
    float4 *x4 = reinterpret_cast<float4 *>(dx);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c4p, tj, ti;
        int64_t tile, n;
        w3_tile_of(i, C4p, tw, th, c4p, tile, tj, ti, n);
        const int ph = c4p / C4, c4 = c4p - ph * C4, rho = ph >> 1, sigma = ph & 1;
        float4 m[5][5];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) m[a][b] = M4[((int64_t)(5 * a + b) * tiles + tile) * C4p + c4p];
        float4 s[3][5];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                float4 acc = w3::zero4();
#pragma unroll
                for (int k = 0; k < 5; ++k) w3::axpy(acc, w3::AT[a][k], m[k][b]);
                s[a][b] = acc;
            }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int h = 3 * ti + a;
            if (h >= H2) continue;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int w = 3 * tj + b;
                if (w >= W2) continue;
                float4 acc = w3::zero4();
#pragma unroll
                for (int k = 0; k < 5; ++k) w3::axpy(acc, w3::AT[b][k], s[a][k]);
                const int64_t pix = (n * H + 2 * h + rho) * W + 2 * w + sigma;
                if (gate) {
                    const float4 gv = reinterpret_cast<const float4 *>(gate)[pix * C4 + c4];
                    acc.x *= gv.x > 0.f ? 1.f : slope;
                    acc.y *= gv.y > 0.f ? 1.f : slope;
                    acc.z *= gv.z > 0.f ? 1.f : slope;
                    acc.w *= gv.w > 0.f ? 1.f : slope;
                    if (padded && padded[(pix / d1) * d2 + pix % d3]) acc = w3::zero4();
                }
                x4[pix * C4 + c4] = acc;
            }
        }
    }
}

// U = G g' G^T of the polyphase 3x3 filters gathered from the 5x5 taps.  w is KRSC (K, 25, C), channel-blocked.
// flip = 0: U[xi][k][c'] (forward);  flip = 1: U[xi][c'][k] of g' rotated by 180 degrees (input gradient)
__global__ __launch_bounds__(256) void wino3_weight_s2d_kernel(const float *__restrict__ w, int K, int C, int flip, float *__restrict__ U)
{
    const int Cp = 4 * C;
    const int64_t total = (int64_t)K * Cp;
    const int cb = chan_block(C);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cp = (int)(i % Cp), k = (int)(i / Cp);
        const int ph = cp / C, c = cp - ph * C, rho = ph >> 1, sigma = ph & 1;
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int aa = flip ? 2 - a : a, bb = flip ? 2 - b : b;
                const int r = 2 * aa + rho, s = 2 * bb + sigma;
                g[a][b] = (r < 5 && s < 5) ? w[(size_t)k * 25 * C + kidx(r * 5 + s, c, 25, cb)] : 0.f;
            }
        float t[5][3];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int s = 0; s < 3; ++s) t[a][s] = (w3::G[a][0] * g[0][s] + w3::G[a][1] * g[1][s]) + w3::G[a][2] * g[2][s];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                const float u = (w3::G[b][0] * t[a][0] + w3::G[b][1] * t[a][1]) + w3::G[b][2] * t[a][2];
                const int xi = 5 * a + b;
                if (flip) U[((size_t)xi * Cp + cp) * K + k] = u;
                else U[((size_t)xi * K + k) * Cp + cp] = u;
            }
    }
}

// dw (K, 25, C) = the 5x5 taps scattered back from G^T P G of the four polyphase channels;  P[xi][k][c']
__global__ __launch_bounds__(256) void wino3_wgrad_inverse_s2d_kernel(const float *__restrict__ P, int K, int C, float *__restrict__ dw)
{
    const int Cp = 4 * C;
    const int64_t total = (int64_t)K * Cp;
    const int cb = chan_block(C);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cp = (int)(i % Cp), k = (int)(i / Cp);
        const int ph = cp / C, c = cp - ph * C, rho = ph >> 1, sigma = ph & 1;
        float p[5][5];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) p[a][b] = P[((size_t)(5 * a + b) * K + k) * Cp + cp];
        float q[3][5];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                float a = 0.f;
#pragma unroll
                for (int m = 0; m < 5; ++m) a += w3::G[m][r] * p[m][b];
                q[r][b] = a;
            }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int r = 2 * a + rho, s = 2 * b + sigma;
                if (r >= 5 || s >= 5) continue;                    // taps the 5x5 filter does not have (their gradient is discarded)
                float acc = 0.f;
#pragma unroll
                for (int m = 0; m < 5; ++m) acc += w3::G[m][b] * q[a][m];
                dw[(size_t)k * 25 * C + kidx(r * 5 + s, c, 25, cb)] = acc;
            }
    }
}

// U = G g G^T.  w is KRSC (K, 9, C) in the channel-blocked order of common.h.
// flip = 0: U[xi][k][c] (forward);  flip = 1: U[xi][c][k] of the filter rotated by 180 degrees (input gradient)
__global__ __launch_bounds__(256) void wino3_weight_kernel(const float *__restrict__ w, int K, int C, int flip, float *__restrict__ U)
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
        float t[5][3];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int s = 0; s < 3; ++s) t[a][s] = (w3::G[a][0] * g[0][s] + w3::G[a][1] * g[1][s]) + w3::G[a][2] * g[2][s];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                const float u = (w3::G[b][0] * t[a][0] + w3::G[b][1] * t[a][1]) + w3::G[b][2] * t[a][2];
                const int xi = 5 * a + b;
                if (flip) U[((size_t)xi * C + c) * K + k] = u;
                else U[((size_t)xi * K + k) * C + c] = u;
            }
    }
}

// dw = G^T P G per (k, c):  P[xi][k][c] -> KRSC (K, 9, C), channel-blocked
__global__ __launch_bounds__(256) void wino3_wgrad_inverse_kernel(const float *__restrict__ P, int K, int C, float *__restrict__ dw)
{
    const int64_t total = (int64_t)K * C;
    const int cb = chan_block(C);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), k = (int)(i / C);
        float p[5][5];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) p[a][b] = P[((size_t)(5 * a + b) * K + k) * C + c];
        float q[3][5];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                float a = 0.f;
#pragma unroll
                for (int m = 0; m < 5; ++m) a += w3::G[m][r] * p[m][b];
                q[r][b] = a;
            }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                float a = 0.f;
#pragma unroll
                for (int m = 0; m < 5; ++m) a += w3::G[m][s] * q[r][m];
                dw[(size_t)k * 9 * C + kidx(r * 3 + s, c, 9, cb)] = a;
            }
    }
}

static int w3_blocks_for(int64_t work_items, int K4)
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

static int w3_args(const void *a, const void *b, int N, int H, int W, int C, const char *who)
{
    MQ_REQUIRE(a && b && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "%s: bad arguments (channels must be a multiple of 4, got %d)", who, C);
    MQ_REQUIRE(((uintptr_t)a | (uintptr_t)b) % 16 == 0, "%s: tensors must be 16-byte aligned", who);
    return 0;
}

extern "C" int mq_wino3_input_transform(const float *x, int N, int H, int W, int C, float *V, mqStream stream)
{
    if (w3_args(x, V, N, H, W, C, "mq_wino3_input_transform")) return 1;
    const int th = (H + 2) / 3, tw = (W + 2) / 3;
    hipLaunchKernelGGL(wino3_input_kernel, dim3(ew_grid((int64_t)N * th * tw * (C / 4), 256)), dim3(256), 0, S(stream), x, (int64_t)N, H, W,
                       C / 4, th, tw, V);
    MQ_LAUNCH_CHECK("mq_wino3_input_transform");
    return 0;
}
extern "C" int mq_wino3_dy_fused_rows(int N, int H, int W, int K)
{
    if (N <= 0 || H <= 0 || W <= 0 || K <= 0 || K % 4 != 0 || K > 1024) return 0;
    return w3_blocks_for((int64_t)N * ((H + 2) / 3) * ((W + 2) / 3) * (K / 4), K / 4);
}
extern "C" int mq_wino3_dy_fused(const float *dy, int N, int H, int W, int K, float *Z, float *Vd, float *bias_partial, mqStream stream)
{
    if (w3_args(dy, Z, N, H, W, K, "mq_wino3_dy_fused")) return 1;
    MQ_REQUIRE(((uintptr_t)Vd | (uintptr_t)bias_partial) % 16 == 0, "mq_wino3_dy_fused: Vd / bias_partial unaligned");
    const int blocks = mq_wino3_dy_fused_rows(N, H, W, K);
    MQ_REQUIRE(blocks > 0 && ((int64_t)blocks * 256) % (K / 4) == 0, "mq_wino3_dy_fused: shape not supported (K <= 1024)");
    const int th = (H + 2) / 3, tw = (W + 2) / 3;
    hipLaunchKernelGGL(wino3_dy_fused_kernel, dim3(blocks), dim3(256), 0, S(stream), dy, (int64_t)N, H, W, K / 4, th, tw, Z, Vd, bias_partial);
    MQ_LAUNCH_CHECK("mq_wino3_dy_fused");
    return 0;
}
extern "C" int mq_wino3_output_transform(const float *M, const float *bias, int N, int H, int W, int K, int act, float slope,
                                         const uint8_t *padded, int64_t d1, int64_t d2, int64_t d3, float *y, mqStream stream)
{
    if (w3_args(M, y, N, H, W, K, "mq_wino3_output_transform")) return 1;
    MQ_REQUIRE(!bias || (uintptr_t)bias % 16 == 0, "mq_wino3_output_transform: bias must be 16-byte aligned");
    MQ_REQUIRE((act == 0 || act == 2) && (!padded || (act == 2 && d1 > 0 && d3 > 0)), "mq_wino3_output_transform: act 0 or 2; a mask needs act 2");
    const int th = (H + 2) / 3, tw = (W + 2) / 3;
    hipLaunchKernelGGL(wino3_output_kernel, dim3(ew_grid((int64_t)N * th * tw * (K / 4), 256)), dim3(256), 0, S(stream), M, bias, (int64_t)N,
                       H, W, K / 4, th, tw, act, slope, padded, d1, d2, d3, y);
    MQ_LAUNCH_CHECK("mq_wino3_output_transform");
    return 0;
}
static int w3_s2d_args(const void *a, const void *b, int N, int H, int W, int C, const char *who)
{
    if (w3_args(a, b, N, H, W, C, who)) return 1;
    MQ_REQUIRE(H % 2 == 0 && W % 2 == 0, "%s: the polyphase form needs even H and W (got %d x %d)", who, H, W);
    return 0;
}
extern "C" int mq_wino3_input_transform_s2d(const float *x, int N, int H, int W, int C, float *V, mqStream stream)
{
    if (w3_s2d_args(x, V, N, H, W, C, "mq_wino3_input_transform_s2d")) return 1;
    const int th = (H / 2 + 2) / 3, tw = (W / 2 + 2) / 3;
    hipLaunchKernelGGL(wino3_input_s2d_kernel, dim3(ew_grid((int64_t)N * th * tw * C, 256)), dim3(256), 0, S(stream), x, (int64_t)N, H, W, C / 4,
                       th, tw, V);
    MQ_LAUNCH_CHECK("mq_wino3_input_transform_s2d");
    return 0;
}
extern "C" int mq_wino3_output_transform_d2s(const float *M, int N, int H, int W, int C, const float *gate, float slope, const uint8_t *padded_in,
                                             int64_t d1, int64_t d2, int64_t d3, float *dx, mqStream stream)
{
    if (w3_s2d_args(M, dx, N, H, W, C, "mq_wino3_output_transform_d2s")) return 1;
    MQ_REQUIRE(!gate || (uintptr_t)gate % 16 == 0, "mq_wino3_output_transform_d2s: gate must be 16-byte aligned");
    MQ_REQUIRE(!padded_in || (gate && d1 > 0 && d3 > 0), "mq_wino3_output_transform_d2s: a mask needs the gate tensor and an index map");
    const int th = (H / 2 + 2) / 3, tw = (W / 2 + 2) / 3;
    hipLaunchKernelGGL(wino3_output_d2s_kernel, dim3(ew_grid((int64_t)N * th * tw * C, 256)), dim3(256), 0, S(stream), M, (int64_t)N, H, W, C / 4,
                       th, tw, gate, slope, padded_in, d1, d2, d3, dx);
    MQ_LAUNCH_CHECK("mq_wino3_output_transform_d2s");
    return 0;
}
extern "C" int mq_wino3_weight_transform_s2d(const float *w_krsc, int K, int C, int flip_transpose, float *U, mqStream stream)
{
    MQ_REQUIRE(w_krsc && U && K > 0 && C > 0, "mq_wino3_weight_transform_s2d: bad arguments");
    hipLaunchKernelGGL(wino3_weight_s2d_kernel, dim3(ew_grid((int64_t)K * 4 * C, 256)), dim3(256), 0, S(stream), w_krsc, K, C, flip_transpose, U);
    MQ_LAUNCH_CHECK("mq_wino3_weight_transform_s2d");
    return 0;
}
extern "C" int mq_wino3_wgrad_inverse_s2d(const float *P, int K, int C, float *dw_krsc, mqStream stream)
{
    MQ_REQUIRE(P && dw_krsc && K > 0 && C > 0, "mq_wino3_wgrad_inverse_s2d: bad arguments");
    hipLaunchKernelGGL(wino3_wgrad_inverse_s2d_kernel, dim3(ew_grid((int64_t)K * 4 * C, 256)), dim3(256), 0, S(stream), P, K, C, dw_krsc);
    MQ_LAUNCH_CHECK("mq_wino3_wgrad_inverse_s2d");
    return 0;
}
extern "C" int mq_wino3_weight_transform(const float *w_krsc, int K, int C, int flip_transpose, float *U, mqStream stream)
{
    MQ_REQUIRE(w_krsc && U && K > 0 && C > 0, "mq_wino3_weight_transform: bad arguments");
    hipLaunchKernelGGL(wino3_weight_kernel, dim3(ew_grid((int64_t)K * C, 256)), dim3(256), 0, S(stream), w_krsc, K, C, flip_transpose, U);
    MQ_LAUNCH_CHECK("mq_wino3_weight_transform");
    return 0;
}
extern "C" int mq_wino3_wgrad_inverse(const float *P, int K, int C, float *dw_krsc, mqStream stream)
{
    MQ_REQUIRE(P && dw_krsc && K > 0 && C > 0, "mq_wino3_wgrad_inverse: bad arguments");
    hipLaunchKernelGGL(wino3_wgrad_inverse_kernel, dim3(ew_grid((int64_t)K * C, 256)), dim3(256), 0, S(stream), P, K, C, dw_krsc);
    MQ_LAUNCH_CHECK("mq_wino3_wgrad_inverse");
    return 0;
}
