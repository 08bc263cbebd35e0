// Winograd F(mh x mw, 3x3) with tiles of 4 or 5 outputs per axis, for the same layers as winograd3.hip (the refiner's 3x3 /
// stride-1 / pad-1 convolutions, reference preencoder.py:86-102, and - through the polyphase form described there - the patch
// discriminator's 5x5 / stride-2 layers, discriminators.py:86-173).
//
//   F(4x4, 3x3): 36 products per 16 outputs = 2.25 per output     F(4x5, 3x3): 42 per 20 = 2.10      (F(3x3, 3x3): 2.78)
//
// Round 2 rejected F(4x4, 3x3) at 1.7e-5 of the output rms - with the textbook points 0, +-1, +-2.  With the per-axis point
// sets of wino_tables.h (0, 1, -1, 2, -1/2 for 4 outputs; 0, 1, -1, 2, -2, 1/2 for 5) the nested transforms measure, in f32
// against float64 at 512 channels (numpy, same method for all): F(3x3) 1.3e-6 rms / 1.4e-5 max, F(4x4) 1.6e-6 / 1.3e-5,
// F(4x5) 2.4e-6 / 2.3e-5, F(5x5) 3.6e-6 / 3.9e-5 (unused); direct f32 2.2e-7 / 1.1e-6.
// The refiner's images are T frames x 90 bins (T = 512 .. 64): 90 = 18 tiles of 5, T = whole tiles of 4 - F(4x5) has no ragged
// edge there; the polyphase images of the patch discriminator are 40 x 128 and 20 x 64.
//
// Structure as winograd3.hip: the (mh + 2)(mw + 2) positions are ONE batch-grouped 1x1 convolution on the implicit-GEMM
// kernels of igemm.hip; this file holds the HBM-bound transforms, f32, channels-last, 4 channels per thread; tile (n, ti, tj)
// covers outputs (mh ti .. mh ti + mh - 1, mw tj .. mw tj + mw - 1) and the inputs one row / column further out.
//
//   forward   V = Bh^T d Bw -> M_xi = V_xi U_xi (U = Gh g Gw^T) -> Y = Ah^T M Aw (+ bias)
//   dgrad     the same with dy as input and the filter rotated by 180 degrees and transposed
//   wgrad     dg = Gh^T [ sum_tiles (Bh^T d Bw) (.) (Ah dY Aw^T) ] Gw
#include "common.h"
#include "wino_tables.h"

namespace mq {

// Channels per thread ("C4" / "K4" in the kernels below count these channel groups).  Through r3 every transform carried FOUR
// channels (float4) through its tile: 233-256 vector registers, one wave per SIMD.  r4 measured TWO (8-byte accesses, half the
// registers, two to four waves per SIMD, twice the blocks) kernel by kernel: the INPUT transforms gain (winon_input 324 -> 272 us =
// 6.1 TB/s, the polyphase one 385 -> 362), the kernels that store pixels or two streams lose (output 231 -> 242, dy pass 357 -> 388,
// polyphase output 582 -> 664) - so the input transforms take two, the others stay at four.
constexpr int WN_VW = 4, WN_IW = 2;
typedef float wvec __attribute__((ext_vector_type(WN_VW)));
typedef float ivec __attribute__((ext_vector_type(WN_IW)));
template <class V> __device__ __forceinline__ V vzero() { V z = {}; return z; }
__device__ __forceinline__ wvec wzero() { return vzero<wvec>(); }

__device__ __forceinline__ void wn_tile_of(int64_t i, int C4, int tw, int th, int &c4, int64_t &tile, int &tj, int &ti, int64_t &n)
{
    c4 = (int)(i % C4);
    tile = i / C4;
    tj = (int)(tile % tw);
    const int64_t r = tile / tw;
    ti = (int)(r % th);
    n = r / th;
}

// v = Bh^T d Bw, stored position-major: V4[(xi * tiles + tile) * C4 + c4], xi = NPW a + b
template <int MH, int MW, class V>
__device__ __forceinline__ void wn_input_tf_store(const V (&d)[MH + 2][MW + 2], V *__restrict__ V4, int64_t tiles, int64_t tile, int C4,
                                                  int c4)
{
    constexpr int NPH = MH + 2, NPW = MW + 2;
    V t[NPH][NPW];
#pragma unroll
    for (int b = 0; b < NPW; ++b)
#pragma unroll
        for (int i = 0; i < NPH; ++i) {
            V a = vzero<V>();
#pragma unroll
            for (int k = 0; k < NPH; ++k) wt::axpyv(a, wt::bt<MH>(i, k), d[k][b]);
            t[i][b] = a;
        }
#pragma unroll
    for (int i = 0; i < NPH; ++i)
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            V a = vzero<V>();
#pragma unroll
            for (int k = 0; k < NPW; ++k) wt::axpyv(a, wt::bt<MW>(j, k), t[i][k]);
            V4[((int64_t)(NPW * i + j) * tiles + tile) * C4 + c4] = a;
        }
}

template <int MH, int MW, class V>
__device__ __forceinline__ void wn_load_window(const V *__restrict__ x4, int64_t n, int H, int W, int C4, int c4, int ti, int tj,
                                               V (&d)[MH + 2][MW + 2])
{
#pragma unroll
    for (int a = 0; a < MH + 2; ++a) {
        const int h = MH * ti - 1 + a;
#pragma unroll
        for (int b = 0; b < MW + 2; ++b) {
            const int w = MW * tj - 1 + b;
            d[a][b] = (h >= 0 && h < H && w >= 0 && w < W) ? x4[((n * H + h) * W + w) * C4 + c4] : vzero<V>();
        }
    }
}

// V[xi][tile][c]
template <int MH, int MW>
__global__ __launch_bounds__(256) void winon_input_kernel(const float *__restrict__ x, int64_t N, int H, int W, int C4, int th, int tw,
                                                          float *__restrict__ V)
{
    const int64_t tiles = N * th * tw, total = tiles * C4;
    const ivec *x4 = reinterpret_cast<const ivec *>(x);
    ivec *V4 = reinterpret_cast<ivec *>(V);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c4, tj, ti;
        int64_t tile, n;
        wn_tile_of(i, C4, tw, th, c4, tile, tj, ti, n);
        ivec d[MH + 2][MW + 2];
        wn_load_window<MH, MW>(x4, n, H, W, C4, c4, ti, tj, d);
        wn_input_tf_store<MH, MW>(d, V4, tiles, tile, C4, c4);
    }
}

// One pass over dy for the backward of a layer: Z = Ah dY Aw^T of the tile's outputs (the middle of the window), V' = Bh^T dy Bw
// (the input gradient's transformed operand, optional) and deterministic per-block column sums (optional; the launch keeps
// (blocks * 256) % K4 == 0, so a thread owns one channel group for the whole loop).
template <int MH, int MW>
__global__ __launch_bounds__(256) void winon_dy_fused_kernel(const float *__restrict__ dy, int64_t N, int H, int W, int K4, int th, int tw,
                                                             float *__restrict__ Z, float *__restrict__ V, float *__restrict__ partial)
{
    constexpr int NPH = MH + 2, NPW = MW + 2;
    __shared__ wvec s_sum[256];
    const int64_t tiles = N * th * tw, total = tiles * K4;
    const wvec *y4 = reinterpret_cast<const wvec *>(dy);
    wvec *Z4 = reinterpret_cast<wvec *>(Z);
    wvec *V4 = reinterpret_cast<wvec *>(V);
    wvec bsum = wzero();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int k4, tj, ti;
        int64_t tile, n;
        wn_tile_of(i, K4, tw, th, k4, tile, tj, ti, n);
        // the 8 x 8 window of F(6x6) and the intermediate of Z do not fit the register file together: there this kernel reads
        // the tile's own 36 outputs only and V' comes from winon_input_kernel (mq_winon_dy_fused launches both)
        constexpr bool SHARED = NPH * NPW <= 42;
        wvec d[NPH][NPW];
        if constexpr (SHARED) wn_load_window<MH, MW>(y4, n, H, W, K4, k4, ti, tj, d);
        {   // Z = Ah y Aw^T, y = the tile's own outputs where they exist (ragged last tiles): window positions 1 .. m
            wvec t[NPH][MW];
#pragma unroll
            for (int b = 0; b < MW; ++b) {
                wvec y[MH];
#pragma unroll
                for (int a = 0; a < MH; ++a) {
                    const int h = MH * ti + a, w = MW * tj + b;
                    const bool in = h < H && w < W;
                    if constexpr (SHARED) y[a] = in ? d[1 + a][1 + b] : wzero();
                    else y[a] = in ? y4[((n * H + h) * W + w) * K4 + k4] : wzero();
                    wt::axpyv(bsum, 1.f, y[a]);
                }
#pragma unroll
                for (int i5 = 0; i5 < NPH; ++i5) {
                    wvec a = wzero();
#pragma unroll
                    for (int k = 0; k < MH; ++k) wt::axpyv(a, wt::at<MH>(k, i5), y[k]);
                    t[i5][b] = a;
                }
            }
#pragma unroll
            for (int i5 = 0; i5 < NPH; ++i5)
#pragma unroll
                for (int j5 = 0; j5 < NPW; ++j5) {
                    wvec a = wzero();
#pragma unroll
                    for (int k = 0; k < MW; ++k) wt::axpyv(a, wt::at<MW>(k, j5), t[i5][k]);
                    Z4[((int64_t)(NPW * i5 + j5) * tiles + tile) * K4 + k4] = a;
                }
        }
        // a tile's window may reach output positions of the NEXT tile that lie beyond a ragged edge: those are zeros already
        if constexpr (SHARED)
            if (V != nullptr) wn_input_tf_store<MH, MW>(d, V4, tiles, tile, K4, k4);
    }
    if (partial == nullptr) return;
    s_sum[threadIdx.x] = bsum;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < K4) {
        wvec acc = s_sum[t];
        for (int j = t + K4; j < 256; j += K4) wt::axpyv(acc, 1.f, s_sum[j]);
        const int k4 = (int)(((int64_t)blockIdx.x * 256 + t) % K4);
        reinterpret_cast<wvec *>(partial)[(int64_t)blockIdx.x * K4 + k4] = acc;
    }
}

// s = Ah^T m  (rows), from the position-major M
template <int MH, int MW>
__device__ __forceinline__ void wn_output_rows(const wvec *__restrict__ M4, int64_t tiles, int64_t tile, int C4, int c4, wvec (&s)[MH][MW + 2])
{
    constexpr int NPH = MH + 2, NPW = MW + 2;
#pragma unroll
    for (int b = 0; b < NPW; ++b) {
        wvec m[NPH];
#pragma unroll
        for (int a = 0; a < NPH; ++a) m[a] = M4[((int64_t)(NPW * a + b) * tiles + tile) * C4 + c4];
#pragma unroll
        for (int a = 0; a < MH; ++a) {
            wvec acc = wzero();
#pragma unroll
            for (int k = 0; k < NPH; ++k) wt::axpyv(acc, wt::at<MH>(a, k), m[k]);
            s[a][b] = acc;
        }
    }
}

// y = epilogue(Ah^T M Aw + bias);  act 0: none, 2: LeakyReLU(slope) then zero where padded[(m / d1) * d2 + m % d3] (m = output pixel
// index), as mq_conv_fprop_leaky
template <int MH, int MW>
__global__ __launch_bounds__(256) void winon_output_kernel(const float *__restrict__ M, const float *__restrict__ bias, int64_t N, int H,
                                                           int W, int K4, int th, int tw, int act, float slope,
                                                           const uint8_t *__restrict__ padded, int64_t d1, int64_t d2, int64_t d3,
                                                           float *__restrict__ y)
{
    constexpr int NPW = MW + 2;
    const int64_t tiles = N * th * tw, total = tiles * K4;
    const wvec *M4 = reinterpret_cast<const wvec *>(M);
    wvec *y4 = reinterpret_cast<wvec *>(y);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int k4, tj, ti;
        int64_t tile, n;
        wn_tile_of(i, K4, tw, th, k4, tile, tj, ti, n);
        wvec s[MH][NPW];
        wn_output_rows<MH, MW>(M4, tiles, tile, K4, k4, s);
        const wvec bv = bias ? reinterpret_cast<const wvec *>(bias)[k4] : wzero();
#pragma unroll
        for (int a = 0; a < MH; ++a) {
            const int h = MH * ti + a;
            if (h >= H) continue;
#pragma unroll
            for (int b = 0; b < MW; ++b) {
                const int w = MW * tj + b;
                if (w >= W) continue;
                wvec acc = bv;
#pragma unroll
                for (int k = 0; k < NPW; ++k) wt::axpyv(acc, wt::at<MW>(b, k), s[a][k]);
                const int64_t pix = (n * H + h) * W + w;
                if (act == 2) {
#pragma unroll
                    for (int e = 0; e < WN_VW; ++e) acc[e] = acc[e] > 0.f ? acc[e] : acc[e] * slope;
                    if (padded && padded[mask_index(pix, d1, d2, d3)]) acc = wzero();
                }
                y4[pix * K4 + k4] = acc;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------------------------------------
// the polyphase form of 5x5 / stride 2 / pad 2 layers (see winograd3.hip): X'[p][q][(rho, sigma, c)] = x[2p + rho][2q + sigma][c]
template <int MH, int MW>
__global__ __launch_bounds__(256) void winon_input_s2d_kernel(const float *__restrict__ x, int64_t N, int H, int W, int C4, int th, int tw,
                                                              float *__restrict__ V)
{
    const int C4p = 4 * C4, H2 = H >> 1, W2 = W >> 1;
    const int64_t tiles = N * th * tw, total = tiles * C4p;
    const ivec *x4 = reinterpret_cast<const ivec *>(x);
    ivec *V4 = reinterpret_cast<ivec *>(V);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c4p, tj, ti;
        int64_t tile, n;
        wn_tile_of(i, C4p, tw, th, c4p, tile, tj, ti, n);
        const int ph = c4p / C4, c4 = c4p - ph * C4, rho = ph >> 1, sigma = ph & 1;
        ivec d[MH + 2][MW + 2];
#pragma unroll
        for (int a = 0; a < MH + 2; ++a) {
            const int h = MH * ti - 1 + a;
#pragma unroll
            for (int b = 0; b < MW + 2; ++b) {
                const int w = MW * tj - 1 + b;
                d[a][b] = (h >= 0 && h < H2 && w >= 0 && w < W2) ? x4[((n * H + 2 * h + rho) * W + 2 * w + sigma) * C4 + c4] : vzero<ivec>();
            }
        }
        wn_input_tf_store<MH, MW>(d, V4, tiles, tile, C4p, c4p);
    }
}

// dx (N, H, W, C) from M (positions, tiles, 4C): Ah^T M Aw per polyphase channel, written to its place in the full-resolution
// image; optionally gated: dx *= gate > 0 ? 1 : slope, zero where padded[(m / d1) * d2 + m % d3] (m = input pixel index)
template <int MH, int MW>
__global__ __launch_bounds__(256) void winon_output_d2s_kernel(const float *__restrict__ M, int64_t N, int H, int W, int C4, int th, int tw,
                                                               const float *__restrict__ gate, float slope,
                                                               const uint8_t *__restrict__ padded, int64_t d1, int64_t d2, int64_t d3,
                                                               float *__restrict__ dx)
{
    constexpr int NPW = MW + 2;
    const int C4p = 4 * C4, H2 = H >> 1, W2 = W >> 1;
    const int64_t tiles = N * th * tw, total = tiles * C4p;
    const wvec *M4 = reinterpret_cast<const wvec *>(M);
    wvec *x4 = reinterpret_cast<wvec *>(dx);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c4p, tj, ti;
        int64_t tile, n;
        wn_tile_of(i, C4p, tw, th, c4p, tile, tj, ti, n);
        const int ph = c4p / C4, c4 = c4p - ph * C4, rho = ph >> 1, sigma = ph & 1;
        wvec s[MH][NPW];
        wn_output_rows<MH, MW>(M4, tiles, tile, C4p, c4p, s);
#pragma unroll
        for (int a = 0; a < MH; ++a) {
            const int h = MH * ti + a;
            if (h >= H2) continue;
#pragma unroll
            for (int b = 0; b < MW; ++b) {
                const int w = MW * tj + b;
                if (w >= W2) continue;
                wvec acc = wzero();
#pragma unroll
                for (int k = 0; k < NPW; ++k) wt::axpyv(acc, wt::at<MW>(b, k), s[a][k]);
                const int64_t pix = (n * H + 2 * h + rho) * W + 2 * w + sigma;
                if (gate) {
                    const wvec gv = reinterpret_cast<const wvec *>(gate)[pix * C4 + c4];
#pragma unroll
                    for (int e = 0; e < WN_VW; ++e) acc[e] *= gv[e] > 0.f ? 1.f : slope;
                    if (padded && padded[mask_index(pix, d1, d2, d3)]) acc = wzero();
                }
                x4[pix * C4 + c4] = acc;
            }
        }
    }
}

// u = Gh g Gw^T of one 3x3 filter, handed position by position to put(xi, u)
template <int MH, int MW, class Put>
__device__ __forceinline__ void wn_weight_tf(const float (&g)[3][3], Put &&put)
{
    constexpr int NPH = MH + 2, NPW = MW + 2;
    float t[NPH][3];
#pragma unroll
    for (int a = 0; a < NPH; ++a)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r) wt::axpy1(acc, wt::g<MH>(a, r), g[r][s]);
            t[a][s] = acc;
        }
#pragma unroll
    for (int a = 0; a < NPH; ++a)
#pragma unroll
        for (int b = 0; b < NPW; ++b) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) wt::axpy1(acc, wt::g<MW>(b, s), t[a][s]);
            put(NPW * a + b, acc);
        }
}

// q = Gh^T p Gw of one position block p (NPH x NPW), handed tap by tap to put(r, s, value)
template <int MH, int MW, class Put>
__device__ __forceinline__ void wn_wgrad_tf(const float (&p)[MH + 2][MW + 2], Put &&put)
{
    constexpr int NPH = MH + 2, NPW = MW + 2;
    float q[3][NPW];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int b = 0; b < NPW; ++b) {
            float a = 0.f;
#pragma unroll
            for (int m = 0; m < NPH; ++m) wt::axpy1(a, wt::g<MH>(m, r), p[m][b]);
            q[r][b] = a;
        }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float a = 0.f;
#pragma unroll
            for (int m = 0; m < NPW; ++m) wt::axpy1(a, wt::g<MW>(m, s), q[r][m]);
            put(r, s, a);
        }
}

// U = Gh g Gw^T.  w is KRSC (K, 9, C) in the channel-blocked order of common.h.
// flip = 0: U[xi][k][c] (forward);  flip = 1: U[xi][c][k] of the filter rotated by 180 degrees (input gradient)
template <int MH, int MW>
__global__ __launch_bounds__(256) void winon_weight_kernel(const float *__restrict__ w, int K, int C, int flip, float *__restrict__ U)
{
    const int64_t total = (int64_t)K * C;
    const int cb = chan_block(C);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // the index that is contiguous in the OUTPUT runs fastest over the lanes (c for U[xi][k][c], k for the flipped
        // U[xi][c][k]): coalesced stores; the strided reads of the small filter bank come from L2
        const int c = flip ? (int)(i / K) : (int)(i % C), k = flip ? (int)(i % K) : (int)(i / C);
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int rs = flip ? (2 - r) * 3 + (2 - s) : r * 3 + s;
                g[r][s] = w[(size_t)k * 9 * C + kidx(rs, c, 9, cb)];
            }
        wn_weight_tf<MH, MW>(g, [&](int xi, float u) {
            if (flip) U[((size_t)xi * C + c) * K + k] = u;
            else U[((size_t)xi * K + k) * C + c] = u;
        });
    }
}

// dw = Gh^T P Gw per (k, c):  P[xi][k][c] -> KRSC (K, 9, C), channel-blocked
template <int MH, int MW>
__global__ __launch_bounds__(256) void winon_wgrad_inverse_kernel(const float *__restrict__ P, int K, int C, float *__restrict__ dw)
{
    constexpr int NPH = MH + 2, NPW = MW + 2;
    const int64_t total = (int64_t)K * C;
    const int cb = chan_block(C);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), k = (int)(i / C);
        float p[NPH][NPW];
#pragma unroll
        for (int a = 0; a < NPH; ++a)
#pragma unroll
            for (int b = 0; b < NPW; ++b) p[a][b] = P[((size_t)(NPW * a + b) * K + k) * C + c];
        wn_wgrad_tf<MH, MW>(p, [&](int r, int s, float v) { dw[(size_t)k * 9 * C + kidx(r * 3 + s, c, 9, cb)] = v; });
    }
}

// U = Gh g' Gw^T of the polyphase 3x3 filters g'[(rho, sigma, c)][a][b] = w[c][2a + rho][2b + sigma] gathered from the 5x5
// taps (zero where an index would be 5).  w is KRSC (K, 25, C), channel-blocked.
// flip = 0: U[xi][k][c'] (forward);  flip = 1: U[xi][c'][k] of g' rotated by 180 degrees (input gradient)
template <int MH, int MW>
__global__ __launch_bounds__(256) void winon_weight_s2d_kernel(const float *__restrict__ w, int K, int C, int flip, float *__restrict__ U)
{
    const int Cp = 4 * C;
    const int64_t total = (int64_t)K * Cp;
    const int cb = chan_block(C);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cp = flip ? (int)(i / K) : (int)(i % Cp), k = flip ? (int)(i % K) : (int)(i / Cp);     // see winon_weight_kernel
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
        wn_weight_tf<MH, MW>(g, [&](int xi, float u) {
            if (flip) U[((size_t)xi * Cp + cp) * K + k] = u;
            else U[((size_t)xi * K + k) * Cp + cp] = u;
        });
    }
}

// dw (K, 25, C) = the 5x5 taps scattered back from Gh^T P Gw of the four polyphase channels;  P[xi][k][c']
template <int MH, int MW>
__global__ __launch_bounds__(256) void winon_wgrad_inverse_s2d_kernel(const float *__restrict__ P, int K, int C, float *__restrict__ dw)
{
    constexpr int NPH = MH + 2, NPW = MW + 2;
    const int Cp = 4 * C;
    const int64_t total = (int64_t)K * Cp;
    const int cb = chan_block(C);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cp = (int)(i % Cp), k = (int)(i / Cp);
        const int ph = cp / C, c = cp - ph * C, rho = ph >> 1, sigma = ph & 1;
        float p[NPH][NPW];
#pragma unroll
        for (int a = 0; a < NPH; ++a)
#pragma unroll
            for (int b = 0; b < NPW; ++b) p[a][b] = P[((size_t)(NPW * a + b) * K + k) * Cp + cp];
        wn_wgrad_tf<MH, MW>(p, [&](int a, int b, float v) {
            const int r = 2 * a + rho, s = 2 * b + sigma;
            if (r < 5 && s < 5) dw[(size_t)k * 25 * C + kidx(r * 5 + s, c, 25, cb)] = v;     // the other taps do not exist
        });
    }
}

static int wn_blocks_for(int64_t work_items, int K4)
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

static bool wn_tile_ok(int mh, int mw) { return (mh == 3 && mw == 3) || (mh == 4 && mw == 4) || (mh == 4 && mw == 5) || (mh == 6 && mw == 6); }

static int wn_args(const void *a, const void *b, int N, int H, int W, int C, int mh, int mw, const char *who)
{
    MQ_REQUIRE(a && b && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "%s: bad arguments (channels must be a multiple of 4, got %d)", who, C);
    MQ_REQUIRE(wn_tile_ok(mh, mw), "%s: tile %d x %d (built: 3 x 3, 4 x 4, 4 x 5 and 6 x 6)", who, mh, mw);
    MQ_REQUIRE(((uintptr_t)a | (uintptr_t)b) % 16 == 0, "%s: tensors must be 16-byte aligned", who);
    return 0;
}

#define MQ_WN_LAUNCH(k_, grid_, ...)                                                                                \
    do {                                                                                                            \
        if (mw == 3) hipLaunchKernelGGL((k_<3, 3>), grid_, dim3(256), 0, S(stream), __VA_ARGS__);                   \
        else if (mw == 4) hipLaunchKernelGGL((k_<4, 4>), grid_, dim3(256), 0, S(stream), __VA_ARGS__);              \
        else if (mw == 6) hipLaunchKernelGGL((k_<6, 6>), grid_, dim3(256), 0, S(stream), __VA_ARGS__);              \
        else hipLaunchKernelGGL((k_<4, 5>), grid_, dim3(256), 0, S(stream), __VA_ARGS__);                           \
    } while (0)

extern "C" int mq_winon_input_transform(const float *x, int N, int H, int W, int C, float *V, int mh, int mw, mqStream stream)
{
    if (wn_args(x, V, N, H, W, C, mh, mw, "mq_winon_input_transform")) return 1;
    const int th = (H + mh - 1) / mh, tw = (W + mw - 1) / mw;
    const dim3 grid(ew_grid((int64_t)N * th * tw * (C / WN_IW), 256));
    MQ_WN_LAUNCH(winon_input_kernel, grid, x, (int64_t)N, H, W, C / WN_IW, th, tw, V);
    MQ_LAUNCH_CHECK("mq_winon_input_transform");
    return 0;
}
extern "C" int mq_winon_dy_fused_rows(int N, int H, int W, int K, int mh, int mw)
{
    if (N <= 0 || H <= 0 || W <= 0 || K <= 0 || K % 4 != 0 || K > 256 * WN_VW || !wn_tile_ok(mh, mw)) return 0;
    return wn_blocks_for((int64_t)N * ((H + mh - 1) / mh) * ((W + mw - 1) / mw) * (K / WN_VW), K / WN_VW);
}
extern "C" int mq_winon_dy_fused(const float *dy, int N, int H, int W, int K, float *Z, float *Vd, float *bias_partial, int mh, int mw,
                                 mqStream stream)
{
    if (wn_args(dy, Z, N, H, W, K, mh, mw, "mq_winon_dy_fused")) return 1;
    MQ_REQUIRE(((uintptr_t)Vd | (uintptr_t)bias_partial) % 16 == 0, "mq_winon_dy_fused: Vd / bias_partial unaligned");
    const int blocks = mq_winon_dy_fused_rows(N, H, W, K, mh, mw);
    MQ_REQUIRE(blocks > 0 && ((int64_t)blocks * 256) % (K / WN_VW) == 0, "mq_winon_dy_fused: shape not supported (K <= 1024)");
    const int th = (H + mh - 1) / mh, tw = (W + mw - 1) / mw;
    const dim3 grid(blocks);
    MQ_WN_LAUNCH(winon_dy_fused_kernel, grid, dy, (int64_t)N, H, W, K / WN_VW, th, tw, Z, Vd, bias_partial);
    if (mw == 6 && Vd) {
        const dim3 gv(ew_grid((int64_t)N * th * tw * (K / WN_IW), 256));
        hipLaunchKernelGGL((winon_input_kernel<6, 6>), gv, dim3(256), 0, S(stream), dy, (int64_t)N, H, W, K / WN_IW, th, tw, Vd);
    }
    MQ_LAUNCH_CHECK("mq_winon_dy_fused");
    return 0;
}
extern "C" int mq_winon_output_transform(const float *M, const float *bias, int N, int H, int W, int K, int act, float slope,
                                         const uint8_t *padded, int64_t d1, int64_t d2, int64_t d3, float *y, int mh, int mw, mqStream stream)
{
    if (wn_args(M, y, N, H, W, K, mh, mw, "mq_winon_output_transform")) return 1;
    MQ_REQUIRE(!bias || (uintptr_t)bias % 16 == 0, "mq_winon_output_transform: bias must be 16-byte aligned");
    MQ_REQUIRE((act == 0 || act == 2) && (!padded || (act == 2 && d1 > 0 && d3 > 0)), "mq_winon_output_transform: act 0 or 2; a mask needs act 2");
    const int th = (H + mh - 1) / mh, tw = (W + mw - 1) / mw;
    const dim3 grid(ew_grid((int64_t)N * th * tw * (K / WN_VW), 256));
    MQ_WN_LAUNCH(winon_output_kernel, grid, M, bias, (int64_t)N, H, W, K / WN_VW, th, tw, act, slope, padded, d1, d2, d3, y);
    MQ_LAUNCH_CHECK("mq_winon_output_transform");
    return 0;
}
static int wn_s2d_args(const void *a, const void *b, int N, int H, int W, int C, int mh, int mw, const char *who)
{
    if (wn_args(a, b, N, H, W, C, mh, mw, who)) return 1;
    MQ_REQUIRE(H % 2 == 0 && W % 2 == 0, "%s: the polyphase form needs even H and W (got %d x %d)", who, H, W);
    return 0;
}
extern "C" int mq_winon_input_transform_s2d(const float *x, int N, int H, int W, int C, float *V, int mh, int mw, mqStream stream)
{
    if (wn_s2d_args(x, V, N, H, W, C, mh, mw, "mq_winon_input_transform_s2d")) return 1;
    const int th = (H / 2 + mh - 1) / mh, tw = (W / 2 + mw - 1) / mw;
    const dim3 grid(ew_grid((int64_t)N * th * tw * (4 * C / WN_IW), 256));
    MQ_WN_LAUNCH(winon_input_s2d_kernel, grid, x, (int64_t)N, H, W, C / WN_IW, th, tw, V);
    MQ_LAUNCH_CHECK("mq_winon_input_transform_s2d");
    return 0;
}
extern "C" int mq_winon_output_transform_d2s(const float *M, int N, int H, int W, int C, const float *gate, float slope, const uint8_t *padded_in,
                                             int64_t d1, int64_t d2, int64_t d3, float *dx, int mh, int mw, mqStream stream)
{
    if (wn_s2d_args(M, dx, N, H, W, C, mh, mw, "mq_winon_output_transform_d2s")) return 1;
    MQ_REQUIRE(!gate || (uintptr_t)gate % 16 == 0, "mq_winon_output_transform_d2s: gate must be 16-byte aligned");
    MQ_REQUIRE(!padded_in || (gate && d1 > 0 && d3 > 0), "mq_winon_output_transform_d2s: a mask needs the gate tensor and an index map");
    const int th = (H / 2 + mh - 1) / mh, tw = (W / 2 + mw - 1) / mw;
    const dim3 grid(ew_grid((int64_t)N * th * tw * (4 * C / WN_VW), 256));
    MQ_WN_LAUNCH(winon_output_d2s_kernel, grid, M, (int64_t)N, H, W, C / WN_VW, th, tw, gate, slope, padded_in, d1, d2, d3, dx);
    MQ_LAUNCH_CHECK("mq_winon_output_transform_d2s");
    return 0;
}
extern "C" int mq_winon_weight_transform_s2d(const float *w_krsc, int K, int C, int flip_transpose, float *U, int mh, int mw, mqStream stream)
{
    MQ_REQUIRE(w_krsc && U && K > 0 && C > 0 && wn_tile_ok(mh, mw), "mq_winon_weight_transform_s2d: bad arguments");
    const dim3 grid(ew_grid((int64_t)K * 4 * C, 256));
    MQ_WN_LAUNCH(winon_weight_s2d_kernel, grid, w_krsc, K, C, flip_transpose, U);
    MQ_LAUNCH_CHECK("mq_winon_weight_transform_s2d");
    return 0;
}
extern "C" int mq_winon_wgrad_inverse_s2d(const float *P, int K, int C, float *dw_krsc, int mh, int mw, mqStream stream)
{
    MQ_REQUIRE(P && dw_krsc && K > 0 && C > 0 && wn_tile_ok(mh, mw), "mq_winon_wgrad_inverse_s2d: bad arguments");
    const dim3 grid(ew_grid((int64_t)K * 4 * C, 256));
    MQ_WN_LAUNCH(winon_wgrad_inverse_s2d_kernel, grid, P, K, C, dw_krsc);
    MQ_LAUNCH_CHECK("mq_winon_wgrad_inverse_s2d");
    return 0;
}
extern "C" int mq_winon_weight_transform(const float *w_krsc, int K, int C, int flip_transpose, float *U, int mh, int mw, mqStream stream)
{
    MQ_REQUIRE(w_krsc && U && K > 0 && C > 0 && wn_tile_ok(mh, mw), "mq_winon_weight_transform: bad arguments");
    const dim3 grid(ew_grid((int64_t)K * C, 256));
    MQ_WN_LAUNCH(winon_weight_kernel, grid, w_krsc, K, C, flip_transpose, U);
    MQ_LAUNCH_CHECK("mq_winon_weight_transform");
    return 0;
}
extern "C" int mq_winon_wgrad_inverse(const float *P, int K, int C, float *dw_krsc, int mh, int mw, mqStream stream)
{
    MQ_REQUIRE(P && dw_krsc && K > 0 && C > 0 && wn_tile_ok(mh, mw), "mq_winon_wgrad_inverse: bad arguments");
    const dim3 grid(ew_grid((int64_t)K * C, 256));
    MQ_WN_LAUNCH(winon_wgrad_inverse_kernel, grid, P, K, C, dw_krsc);
    MQ_LAUNCH_CHECK("mq_winon_wgrad_inverse");
    return 0;
}
