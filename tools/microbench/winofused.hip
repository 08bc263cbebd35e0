// PROTOTYPE (round 5, review item 2c): F(4x5, 3x3) with the transforms INSIDE the GEMM kernel, for layers with K = 64 output
// channels (the refiner's outer level, reference preencoder.py:86-102: 64 -> 64 at 512 x 90).  DESIGN 3.6 argued by arithmetic
// that a fused Winograd kernel cannot pay on the f32 MFMA pipe at >= 256 channels and "would pay 4-7 ms" for K <= 128; this file
// turns that into a measurement on one layer shape (tools/winofused_micro.py).  Not selected by mq_conv_plan.
//
// Why it can exist at all: with v_mfma_f32_16x16x4_f32 a lane of the 16 x 16 accumulator holds 4 tiles x ONE output channel,
// so if a wave keeps ALL 42 transform positions of its (16 tiles x 16 channels) as 42 x 4 accumulator registers, the output
// transform Y = Ah^T M Aw is lane-local - no exchange, no M tensor.  On the input side the A operand of that MFMA is one value per
// lane (tile l & 15, channel 4 s + (l >> 4)), so the lane loads the 6 x 7 window of ITS (tile, channel), forms V = Bh^T d Bw in
// registers (no V tensor) and feeds the 42 positions to 42 MFMAs against U_xi[c][16-channel slice].  Per 4 input channels and wave:
// 42 window loads (L1 / L2: a pixel's 256-byte line serves 16 such steps), ~380 transform FMAs, 42 filter loads (U^T, 688 KB at
// 64 x 64: L2-resident, shared by every block), 42 MFMAs (32 cycles each).  HBM traffic: x once (+ halo from L2), y once.
#include "common.h"
#include "gemm_core.h"
#include "wino_tables.h"

namespace mq {

constexpr int WF_MH = 4, WF_MW = 5, WF_NPH = 6, WF_NPW = 7, WF_NP = 42;

// Ut[xi][c][k] <- U[xi][k][c]  (U as mq_winon_weight_transform writes it for the batch-grouped 1x1 GEMM)
__global__ void winofused_ut_kernel(const float *__restrict__ U, int K, int C, float *__restrict__ Ut)
{
    const int64_t total = (int64_t)WF_NP * K * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int64_t r = i / K;
        const int c = (int)(r % C), xi = (int)(r / C);
        Ut[i] = U[((int64_t)xi * K + k) * C + c];
    }
}

// grid: ceil(tiles / 16) blocks of K / 16 waves; wave w owns output channels [16 w, 16 w + 16) of the block's 16 tiles
__global__ __launch_bounds__(256) void winofused_fprop_kernel(const float *__restrict__ x, const float *__restrict__ Ut,
                                                              const float *__restrict__ bias, int64_t N, int H, int W, int C, int K, int th,
                                                              int tw, float *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 15, kk = lane >> 4;
    const int64_t tiles = N * th * tw;
    // ---- this lane's tile as the A operand sees it: tile m of the block
    const int64_t ta = (int64_t)blockIdx.x * 16 + m;
    const bool tav = ta < tiles;
    const int64_t tq = tav ? ta : 0;
    const int tj = (int)(tq % tw);
    const int ti = (int)((tq / tw) % th);
    const int64_t n = tq / tw / th;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (size_t)N * H * W * C * 4);
    unsigned woff[WF_NPH][WF_NPW];          // byte offset of window pixel (a, b), channel kk; OOB outside the image
#pragma unroll
    for (int a = 0; a < WF_NPH; ++a) {
        const int h = WF_MH * ti - 1 + a;
#pragma unroll
        for (int b = 0; b < WF_NPW; ++b) {
            const int w = WF_MW * tj - 1 + b;
            const bool in = tav && h >= 0 && h < H && w >= 0 && w < W;
            woff[a][b] = in ? (unsigned)((((n * H + h) * W + w) * C + kk) * 4) : OOB;
        }
    }
    const float *ub = Ut + 16 * wave + m;                     // + (xi * C + c) * K
    f32x4v acc[WF_NP];
#pragma unroll
    for (int p = 0; p < WF_NP; ++p) acc[p] = f32x4v{0.f, 0.f, 0.f, 0.f};

    for (int c0 = 0; c0 < C; c0 += 4) {
        // ---- window of (tile m, channel c0 + kk)
        float d[WF_NPH][WF_NPW];
#pragma unroll
        for (int a = 0; a < WF_NPH; ++a)
#pragma unroll
            for (int b = 0; b < WF_NPW; ++b)
                d[a][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                        rx, woff[a][b] == OOB ? (int)OOB : (int)(woff[a][b] + (unsigned)c0 * 4u), 0, 0));
        // ---- filter values of the 42 positions for (channel c0 + kk, output channel 16 wave + m)
        float u[WF_NP];
        const float *up = ub + (size_t)(c0 + kk) * K;
#pragma unroll
        for (int p = 0; p < WF_NP; ++p) u[p] = up[(size_t)p * C * K];
        // ---- V = Bh^T d Bw
        float t[WF_NPH][WF_NPW];
#pragma unroll
        for (int b = 0; b < WF_NPW; ++b)
#pragma unroll
            for (int i = 0; i < WF_NPH; ++i) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < WF_NPH; ++k) wt::axpy1(s, wt::bt<WF_MH>(i, k), d[k][b]);
                t[i][b] = s;
            }
#pragma unroll
        for (int i = 0; i < WF_NPH; ++i)
#pragma unroll
            for (int j = 0; j < WF_NPW; ++j) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < WF_NPW; ++k) wt::axpy1(v, wt::bt<WF_MW>(j, k), t[i][k]);
                acc[WF_NPW * i + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, u[WF_NPW * i + j], acc[WF_NPW * i + j], 0, 0, 0);
            }
    }
    // ---- output transform, lane-local: register j of every position belongs to tile 4 (lane >> 4) + j, channel 16 wave + m
    const int kout = 16 * wave + m;
    const float bv = bias ? bias[kout] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t to = (int64_t)blockIdx.x * 16 + 4 * kk + j;
        if (to >= tiles) continue;
        const int otj = (int)(to % tw);
        const int oti = (int)((to / tw) % th);
        const int64_t on = to / tw / th;
        float s[WF_MH][WF_NPW];
#pragma unroll
        for (int b = 0; b < WF_NPW; ++b)
#pragma unroll
            for (int a = 0; a < WF_MH; ++a) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < WF_NPH; ++k) wt::axpy1(v, wt::at<WF_MH>(a, k), acc[WF_NPW * k + b][j]);
                s[a][b] = v;
            }
#pragma unroll
        for (int a = 0; a < WF_MH; ++a) {
            const int h = WF_MH * oti + a;
            if (h >= H) continue;
#pragma unroll
            for (int b = 0; b < WF_MW; ++b) {
                const int w = WF_MW * otj + b;
                if (w >= W) continue;
                float v = bv;
#pragma unroll
                for (int k = 0; k < WF_NPW; ++k) wt::axpy1(v, wt::at<WF_MW>(b, k), s[a][k]);
                y[((on * H + h) * W + w) * K + kout] = v;
            }
        }
    }
}

}  // namespace mq

using namespace mq;

// y (N, H, W, K) = conv3x3(x (N, H, W, C), w) + bias, stride 1, pad 1, as F(4x5, 3x3) in ONE kernel.  U: the transformed filters of
// mq_winon_weight_transform(w_krsc, K, C, 0, U, 4, 5); Ut: scratch of 42 * C * K floats.  Prototype: K % 16 == 0, K <= 64,
// C % 4 == 0, the tensors below 4 GiB.
extern "C" int mq_winofused_fprop(const float *x, const float *U, const float *bias, int N, int H, int W, int C, int K, float *Ut, float *y,
                                  mqStream stream)
{
    MQ_REQUIRE(x && U && Ut && y && N > 0 && H > 0 && W > 0, "mq_winofused_fprop: bad arguments");
    MQ_REQUIRE(K % 16 == 0 && K <= 64 && C % 4 == 0 && C > 0, "mq_winofused_fprop: prototype takes K %% 16 == 0, K <= 64, C %% 4 == 0 (C=%d, K=%d)", C, K);
    MQ_REQUIRE((size_t)N * H * W * C * 4 < 0xFFFFFFF0ull, "mq_winofused_fprop: x must stay below 4 GiB");
    const int th = (H + WF_MH - 1) / WF_MH, tw = (W + WF_MW - 1) / WF_MW;
    const int64_t tiles = (int64_t)N * th * tw;
    hipLaunchKernelGGL(winofused_ut_kernel, dim3(ew_grid((int64_t)WF_NP * K * C, 256)), dim3(256), 0, S(stream), U, K, C, Ut);
    hipLaunchKernelGGL(winofused_fprop_kernel, dim3((unsigned)cdiv(tiles, 16)), dim3(64 * (K / 16)), 0, S(stream), x, Ut, bias, (int64_t)N, H, W,
                       C, K, th, tw, y);
    MQ_LAUNCH_CHECK("mq_winofused_fprop");
    return 0;
}
