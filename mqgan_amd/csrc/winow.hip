// One-dimensional Winograd F(m, s) along the image COLUMNS (the time axis of every tensor on this path) for convolutions with
// s = 3, 5 or 7 taps, stride 1 and no dilation in that direction, any taps / stride along the rows, any left padding (the
// right padding follows from the output width), batch groups:
//   * the encoder / decoder blocks' 1-D convolutions (reference attentions.py:95-130: k = 3, 5, 7, "same" or causal padding),
//   * the patch discriminator's 3x3 / stride-(2, 1) layers (discriminators.py:86-173),
//   * the (1 x 5) row convolutions INSIDE the multi-bin discriminator's F(5, 3) form (wino1d.hip): the two one-dimensional
//     transforms nest to F(5 x 4, 3 x 5), 56 products per 20 outputs instead of 300.
// Tiles: F(4, 3) (6 products per 4 outputs), F(4, 5) (8 per 4), F(2, 7) (8 per 2); matrices and their measured f32 error:
// wino_tables.h.  wino1d.hip's scheme with the axes exchanged:
//
//   V_xi = sum_j BT[xi][j] x[.., m t - pl + j, ..]        (n, h, tile t);  xi = 0 .. m + s - 2
//   M_xi = conv_{R x 1}(V_xi, U_xi),  U_xi[r] = sum_s G[xi][s] g[r][s]     one batch-grouped convolution on igemm.hip's kernels:
//                                                                          images (xi, n) of H x tw, groups = (m + s - 1) x groups
//   y[.., m t + a, ..] = sum_xi AT[a][xi] M_xi  (+ bias, LeakyReLU, frame mask)
//   backward: Z = A dY;  dV = dgrad(Z, U);  dx = B dV;  dU = wgrad(V, Z);  dg = G^T dU.
// f32, channels-last, 4 channels per thread; every transform is one pass at the HBM roof.
#include "common.h"
#include "wino_tables.h"

namespace mq {

// V[xi][n][h][t][c]
template <int M, int S>
__global__ __launch_bounds__(256) void winow_input_kernel(const float *__restrict__ x, int64_t N, int H, int W, int C4, int tw, int pl,
                                                          float *__restrict__ V)
{
    constexpr int NP = M + S - 1;
    const int64_t plane = N * H * tw * C4;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    float4 *V4 = reinterpret_cast<float4 *>(V);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t r = i / C4;
        const int t = (int)(r % tw);
        const int64_t nh = r / tw;                                         // n * H + h
        float4 d[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int w = M * t - pl + j;
            d[j] = (w >= 0 && w < W) ? x4[(nh * W + w) * C4 + c4] : wt::zero4();
        }
#pragma unroll
        for (int xi = 0; xi < NP; ++xi) {
            float4 a = wt::zero4();
#pragma unroll
            for (int j = 0; j < NP; ++j) wt::axpy(a, wt::bt2<M, S>(xi, j), d[j]);
            V4[xi * plane + i] = a;
        }
    }
}

// y = epilogue(A^T M + bias).  M[xi][n][p][t][k]; act 0: none, 2: LeakyReLU(slope) then zero where padded[(m / d1) * d2 + m % d3]
// (m = output pixel index), as mq_conv_fprop_leaky
template <int M, int S>
__global__ __launch_bounds__(256) void winow_output_kernel(const float *__restrict__ Mt, const float *__restrict__ bias, int64_t N, int P,
                                                           int Q, int K4, int tw, int groups, int act, float slope,
                                                           const uint8_t *__restrict__ padded, int64_t d1, int64_t d2, int64_t d3,
                                                           float *__restrict__ y)
{
    constexpr int NP = M + S - 1;
    const int64_t plane = N * P * tw * K4;
    const float4 *M4 = reinterpret_cast<const float4 *>(Mt);
    float4 *y4 = reinterpret_cast<float4 *>(y);
    const int64_t per_group = N / groups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (int64_t)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % K4);
        int64_t r = i / K4;
        const int t = (int)(r % tw);
        const int64_t np_ = r / tw;                                        // n * P + p
        const int64_t n = np_ / P;
        float4 m[NP];
#pragma unroll
        for (int xi = 0; xi < NP; ++xi) m[xi] = M4[xi * plane + i];
        const float4 bv = bias ? reinterpret_cast<const float4 *>(bias)[(n / per_group) * K4 + k4] : wt::zero4();
#pragma unroll
        for (int a = 0; a < M; ++a) {
            const int q = M * t + a;
            if (q >= Q) continue;
            float4 v = bv;
#pragma unroll
            for (int xi = 0; xi < NP; ++xi) wt::axpy(v, wt::at2<M, S>(a, xi), m[xi]);
            const int64_t pix = np_ * Q + q;
            if (act == 2) {
                v.x = v.x > 0.f ? v.x : v.x * slope;
                v.y = v.y > 0.f ? v.y : v.y * slope;
                v.z = v.z > 0.f ? v.z : v.z * slope;
                v.w = v.w > 0.f ? v.w : v.w * slope;
                if (padded && padded[mask_index(pix, d1, d2, d3)]) v = wt::zero4();
            }
            y4[pix * K4 + k4] = v;
        }
    }
}

// Z[xi][n][p][t][k] = A dY, plus per-block column sums of dy for the bias gradient.  grid (blocks, groups): block (b, g) walks
// group g's elements; the launch keeps (blocks * 256) % K4 == 0, so a thread owns one channel group for the whole loop.
template <int M, int S>
__global__ __launch_bounds__(256) void winow_dy_kernel(const float *__restrict__ dy, int64_t N, int P, int Q, int K4, int tw, int groups,
                                                       float *__restrict__ Z, float *__restrict__ partial)
{
    constexpr int NP = M + S - 1;
    __shared__ float4 s_sum[256];
    const int64_t per_group = N / groups;
    const int g = blockIdx.y;
    const int64_t plane = N * P * tw * K4;
    const int64_t gsize = per_group * P * tw * K4, gbase = (int64_t)g * gsize;
    const float4 *y4 = reinterpret_cast<const float4 *>(dy);
    float4 *Z4 = reinterpret_cast<float4 *>(Z);
    float4 bsum = wt::zero4();
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < gsize; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = gbase + j;
        const int k4 = (int)(i % K4);
        int64_t r = i / K4;
        const int t = (int)(r % tw);
        const int64_t np_ = r / tw;
        float4 yv[M];
#pragma unroll
        for (int a = 0; a < M; ++a) {
            const int q = M * t + a;
            yv[a] = q < Q ? y4[(np_ * Q + q) * K4 + k4] : wt::zero4();
            bsum = make_float4(bsum.x + yv[a].x, bsum.y + yv[a].y, bsum.z + yv[a].z, bsum.w + yv[a].w);
        }
#pragma unroll
        for (int xi = 0; xi < NP; ++xi) {
            float4 z = wt::zero4();
#pragma unroll
            for (int a = 0; a < M; ++a) wt::axpy(z, wt::at2<M, S>(a, xi), yv[a]);
            Z4[xi * plane + i] = z;
        }
    }
    if (partial == nullptr) return;
    s_sum[threadIdx.x] = bsum;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < K4) {
        float4 acc = s_sum[t];
        for (int j = t + K4; j < 256; j += K4) wt::axpy(acc, 1.f, s_sum[j]);
        const int k4 = (int)(((int64_t)blockIdx.x * 256 + t) % K4);
        reinterpret_cast<float4 *>(partial)[((int64_t)blockIdx.x * gridDim.y + g) * K4 + k4] = acc;      // [block][group][K]
    }
}

// dx = B dV.  Thread (n, h, t, c4) writes input columns m t - pl .. m t - pl + m - 1 (position j = 0 .. m-1 of tile t's
// window); the same column is position j + m d of tile t - d, d = 1 .. ceil((m + s - 1) / m) - 1.  t runs to twx >= tw so that
// the columns behind the last tile's first m are written too; tiles >= tw do not exist.  Optionally gated: dx *= gate > 0 ? 1 :
// slope, zero where padded[(m / d1) * d2 + m % d3] (m = input pixel index).
template <int M, int S>
__global__ __launch_bounds__(256) void winow_dx_kernel(const float *__restrict__ dV, int64_t N, int H, int W, int C4, int tw, int twx, int pl,
                                                       const float *__restrict__ gate, float slope, const uint8_t *__restrict__ padded,
                                                       int64_t d1, int64_t d2, int64_t d3, float *__restrict__ dx)
{
    constexpr int NP = M + S - 1, ND = (NP + M - 1) / M;
    const int64_t plane = N * H * tw * C4;
    const int64_t total = N * H * twx * C4;
    const float4 *V4 = reinterpret_cast<const float4 *>(dV);
    float4 *x4 = reinterpret_cast<float4 *>(dx);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t r = i / C4;
        const int t = (int)(r % twx);
        const int64_t nh = r / twx;
        float4 o[M];
#pragma unroll
        for (int j = 0; j < M; ++j) o[j] = wt::zero4();
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) {
            const int tt = t - dd;
            if (tt < 0 || tt >= tw) continue;
            const int64_t e = (nh * tw + tt) * C4 + c4;
#pragma unroll
            for (int xi = 0; xi < NP; ++xi) {
                bool used = false;
#pragma unroll
                for (int j = 0; j < M; ++j) used = used || (j + M * dd < NP && wt::bt2<M, S>(xi, j + M * dd < NP ? j + M * dd : 0) != 0.f);
                if (!used) continue;
                const float4 v = V4[xi * plane + e];
#pragma unroll
                for (int j = 0; j < M; ++j)
                    if (j + M * dd < NP) wt::axpy(o[j], wt::bt2<M, S>(xi, j + M * dd < NP ? j + M * dd : 0), v);
            }
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const int w = M * t - pl + j;
            if (w < 0 || w >= W) continue;
            const int64_t pix = nh * W + w;
            float4 v = o[j];
            if (gate) {
                const float4 gv = reinterpret_cast<const float4 *>(gate)[pix * C4 + c4];
                v.x *= gv.x > 0.f ? 1.f : slope;
                v.y *= gv.y > 0.f ? 1.f : slope;
                v.z *= gv.z > 0.f ? 1.f : slope;
                v.w *= gv.w > 0.f ? 1.f : slope;
                if (padded && padded[mask_index(pix, d1, d2, d3)]) v = wt::zero4();
            }
            x4[pix * C4 + c4] = v;
        }
    }
}

// U = G g per filter row.  w is KRSC (GK, R * S, C) in the channel-blocked order of common.h (GK = groups * K).
// U_krsc (NP * GK, R, C) for fprop / wgrad; U_crsk (NP * groups * C, R, K), K-blocked, for dgrad (may be null).
template <int M, int S>
__global__ __launch_bounds__(256) void winow_weight_kernel(const float *__restrict__ w, int GK, int K, int C, int R, float *__restrict__ U_krsc,
                                                           float *__restrict__ U_crsk)
{
    constexpr int NP = M + S - 1;
    const int64_t total = (int64_t)GK * R * C;
    const int cbc = chan_block(C), cbk = chan_block(K);
    const int groups = GK / K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int r = (int)((i / C) % R);
        const int gk = (int)(i / ((int64_t)C * R));
        const float *wr = w + (size_t)gk * R * S * C;
        float g[S];
#pragma unroll
        for (int s = 0; s < S; ++s) g[s] = wr[kidx(r * S + s, c, R * S, cbc)];
#pragma unroll
        for (int xi = 0; xi < NP; ++xi) {
            float u = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) wt::axpy1(u, wt::g2<M, S>(xi, s), g[s]);
            U_krsc[((size_t)xi * GK + gk) * R * C + kidx(r, c, R, cbc)] = u;
        }
    }
    if (!U_crsk) return;
    // the CRSK copy in a pass of its own with k running fastest: contiguous stores per 16-channel block instead of 64 lines per
    // wave store; the strided re-read of w comes from L2
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int r = (int)((i / K) % R);
        const int c = (int)((i / ((int64_t)K * R)) % C);
        const int grp = (int)(i / ((int64_t)K * R * C));
        const float *wr = w + ((size_t)grp * K + k) * R * S * C;
        float g[S];
#pragma unroll
        for (int s = 0; s < S; ++s) g[s] = wr[kidx(r * S + s, c, R * S, cbc)];
#pragma unroll
        for (int xi = 0; xi < NP; ++xi) {
            float u = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) wt::axpy1(u, wt::g2<M, S>(xi, s), g[s]);
            U_crsk[(((size_t)xi * groups + grp) * C + c) * R * K + kidx(r, k, R, cbk)] = u;
        }
    }
}

// dw = G^T dU:  dU (NP * GK, R, C) -> dw KRSC (GK, R * S, C), channel-blocked
template <int M, int S>
__global__ __launch_bounds__(256) void winow_wgrad_inverse_kernel(const float *__restrict__ dU, int GK, int C, int R, float *__restrict__ dw)
{
    constexpr int NP = M + S - 1;
    const int64_t total = (int64_t)GK * R * C;
    const int cbc = chan_block(C);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int r = (int)((i / C) % R);
        const int gk = (int)(i / ((int64_t)C * R));
        float p[NP];
#pragma unroll
        for (int xi = 0; xi < NP; ++xi) p[xi] = dU[((size_t)xi * GK + gk) * R * C + kidx(r, c, R, cbc)];
        float *o = dw + (size_t)gk * R * S * C;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float a = 0.f;
#pragma unroll
            for (int xi = 0; xi < NP; ++xi) wt::axpy1(a, wt::g2<M, S>(xi, s), p[xi]);
            o[kidx(r * S + s, c, R * S, cbc)] = a;
        }
    }
}

static int ww_blocks_for(int64_t work_items, int K4)
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

// the tile of an s-tap filter: 4 outputs for 3 and 5 taps, 2 for 7 (eight interpolation points at most: wino_tables.h)
extern "C" int mq_winow_tile(int taps) { return taps == 3 || taps == 5 ? 4 : taps == 7 ? 2 : 0; }

static int ww_args(const void *a, const void *b, int N, int H, int W, int C, int taps, const char *who)
{
    MQ_REQUIRE(a && b && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "%s: bad arguments (channels must be a multiple of 4, got %d)", who, C);
    MQ_REQUIRE(mq_winow_tile(taps) > 0, "%s: %d column taps (3, 5 or 7)", who, taps);
    MQ_REQUIRE(((uintptr_t)a | (uintptr_t)b) % 16 == 0, "%s: tensors must be 16-byte aligned", who);
    return 0;
}

#define MQ_WW_LAUNCH(taps_, k_, grid_, ...)                                                                              \
    do {                                                                                                                 \
        if ((taps_) == 3) hipLaunchKernelGGL((k_<4, 3>), grid_, dim3(256), 0, S(stream), __VA_ARGS__);                   \
        else if ((taps_) == 5) hipLaunchKernelGGL((k_<4, 5>), grid_, dim3(256), 0, S(stream), __VA_ARGS__);              \
        else hipLaunchKernelGGL((k_<2, 7>), grid_, dim3(256), 0, S(stream), __VA_ARGS__);                                \
    } while (0)

// x (N, H, W, C) -> V (np, N, H, tw, C), tw = ceil(Q / m) for the output width Q; pad_left: zero columns in front of the image
extern "C" int mq_winow_input_transform(const float *x, int N, int H, int W, int C, int taps, int Q, int pad_left, float *V, mqStream stream)
{
    if (ww_args(x, V, N, H, W, C, taps, "mq_winow_input_transform")) return 1;
    MQ_REQUIRE(Q > 0 && pad_left >= 0, "mq_winow_input_transform: output width / left padding");
    const int m = mq_winow_tile(taps), tw = (Q + m - 1) / m;
    const dim3 grid(ew_grid((int64_t)N * H * tw * (C / 4), 256));
    MQ_WW_LAUNCH(taps, winow_input_kernel, grid, x, (int64_t)N, H, W, C / 4, tw, pad_left, V);
    MQ_LAUNCH_CHECK("mq_winow_input_transform");
    return 0;
}

extern "C" int mq_winow_output_transform(const float *M, const float *bias, int N, int P, int Q, int K, int taps, int groups, int act,
                                         float slope, const uint8_t *padded, int64_t d1, int64_t d2, int64_t d3, float *y, mqStream stream)
{
    if (ww_args(M, y, N, P, Q, K, taps, "mq_winow_output_transform")) return 1;
    MQ_REQUIRE(groups > 0 && N % groups == 0 && (act == 0 || act == 2), "mq_winow_output_transform: groups must divide N, act 0 or 2");
    MQ_REQUIRE(!bias || (uintptr_t)bias % 16 == 0, "mq_winow_output_transform: bias must be 16-byte aligned");
    MQ_REQUIRE(!padded || (act == 2 && d1 > 0 && d3 > 0), "mq_winow_output_transform: a mask needs act 2 and an index map");
    const int m = mq_winow_tile(taps), tw = (Q + m - 1) / m;
    const dim3 grid(ew_grid((int64_t)N * P * tw * (K / 4), 256));
    MQ_WW_LAUNCH(taps, winow_output_kernel, grid, M, bias, (int64_t)N, P, Q, K / 4, tw, groups, act, slope, padded, d1, d2, d3, y);
    MQ_LAUNCH_CHECK("mq_winow_output_transform");
    return 0;
}

extern "C" int mq_winow_dy_rows(int N, int P, int Q, int K, int taps, int groups)
{
    const int m = mq_winow_tile(taps);
    if (N <= 0 || P <= 0 || Q <= 0 || K <= 0 || K % 4 != 0 || K > 1024 || groups <= 0 || N % groups != 0 || m == 0) return 0;
    return ww_blocks_for((int64_t)(N / groups) * P * ((Q + m - 1) / m) * (K / 4), K / 4);
}

// bias_partial (may be NULL): [mq_winow_dy_rows(...)][groups * K] per-block column sums of dy
extern "C" int mq_winow_dy_transform(const float *dy, int N, int P, int Q, int K, int taps, int groups, float *Z, float *bias_partial,
                                     mqStream stream)
{
    if (ww_args(dy, Z, N, P, Q, K, taps, "mq_winow_dy_transform")) return 1;
    MQ_REQUIRE(groups > 0 && N % groups == 0, "mq_winow_dy_transform: groups must divide N");
    MQ_REQUIRE(!bias_partial || (uintptr_t)bias_partial % 16 == 0, "mq_winow_dy_transform: bias_partial must be 16-byte aligned");
    const int m = mq_winow_tile(taps), tw = (Q + m - 1) / m;
    int blocks = mq_winow_dy_rows(N, P, Q, K, taps, groups);
    if (!bias_partial && blocks == 0) blocks = ew_grid((int64_t)(N / groups) * P * tw * (K / 4), 256);     // K > 1024: no sums asked
    MQ_REQUIRE(blocks > 0, "mq_winow_dy_transform: the bias sums need K <= 1024");
    const dim3 grid(blocks, groups);
    MQ_WW_LAUNCH(taps, winow_dy_kernel, grid, dy, (int64_t)N, P, Q, K / 4, tw, groups, Z, bias_partial);
    MQ_LAUNCH_CHECK("mq_winow_dy_transform");
    return 0;
}

extern "C" int mq_winow_dx_transform(const float *dV, int N, int H, int W, int C, int taps, int Q, int pad_left, const float *gate, float slope,
                                     const uint8_t *padded_in, int64_t d1, int64_t d2, int64_t d3, float *dx, mqStream stream)
{
    if (ww_args(dV, dx, N, H, W, C, taps, "mq_winow_dx_transform")) return 1;
    MQ_REQUIRE(Q > 0 && pad_left >= 0, "mq_winow_dx_transform: output width / left padding");
    MQ_REQUIRE(!gate || (uintptr_t)gate % 16 == 0, "mq_winow_dx_transform: gate must be 16-byte aligned");
    MQ_REQUIRE(!padded_in || (gate && d1 > 0 && d3 > 0), "mq_winow_dx_transform: a mask needs the gate tensor and an index map");
    const int m = mq_winow_tile(taps), tw = (Q + m - 1) / m, twx = (W + pad_left + m - 1) / m;
    const dim3 grid(ew_grid((int64_t)N * H * twx * (C / 4), 256));
    MQ_WW_LAUNCH(taps, winow_dx_kernel, grid, dV, (int64_t)N, H, W, C / 4, tw, twx, pad_left, gate, slope, padded_in, d1, d2, d3, dx);
    MQ_LAUNCH_CHECK("mq_winow_dx_transform");
    return 0;
}

// w_krsc (groups * K, R * taps, C) -> U_krsc (np * groups * K, R, C) and, if not NULL, U_crsk (np * groups * C, R, K)
extern "C" int mq_winow_weight_transform(const float *w_krsc, int groups, int K, int C, int R, int taps, float *U_krsc, float *U_crsk,
                                         mqStream stream)
{
    MQ_REQUIRE(w_krsc && U_krsc && groups > 0 && K > 0 && C > 0 && R > 0 && mq_winow_tile(taps) > 0, "mq_winow_weight_transform: bad arguments");
    const dim3 grid(ew_grid((int64_t)groups * K * R * C, 256));
    MQ_WW_LAUNCH(taps, winow_weight_kernel, grid, w_krsc, groups * K, K, C, R, U_krsc, U_crsk);
    MQ_LAUNCH_CHECK("mq_winow_weight_transform");
    return 0;
}

extern "C" int mq_winow_wgrad_inverse(const float *dU, int groups, int K, int C, int R, int taps, float *dw_krsc, mqStream stream)
{
    MQ_REQUIRE(dU && dw_krsc && groups > 0 && K > 0 && C > 0 && R > 0 && mq_winow_tile(taps) > 0, "mq_winow_wgrad_inverse: bad arguments");
    const dim3 grid(ew_grid((int64_t)groups * K * R * C, 256));
    MQ_WW_LAUNCH(taps, winow_wgrad_inverse_kernel, grid, dU, groups * K, C, R, dw_krsc);
    MQ_LAUNCH_CHECK("mq_winow_wgrad_inverse");
    return 0;
}
