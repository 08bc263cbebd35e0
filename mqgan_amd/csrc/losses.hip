// Loss kernels: MaskedMelLoss (losses.py:146-182), LSGANLoss with LeCam (losses.py:5-116),
// masked_mae (train.py:38-45).  No host synchronisation: the reference's `.item()` branches
// (losses.py:29) are taken on the device.
#include "common.h"

namespace mq {

// ------------------------------------------------------------------------------------------ masked mel loss
// partial[chunk][c] = sum over the chunk's valid frames of (x-y)^2 ; grid (ceil(C/64), nchunks), block (64,4)
__global__ void mel_partial_kernel(const float *__restrict__ x, const float *__restrict__ y, const int32_t *__restrict__ len, int B, int T,
                                   int C, int nchunks, float *__restrict__ part)
{
    __shared__ float ss[4][64];
    const int c = blockIdx.x * 64 + threadIdx.x, ty = threadIdx.y;
    const int64_t rows = (int64_t)B * T;
    const int64_t per = (rows + nchunks - 1) / nchunks;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(rows, r0 + per);
    float s = 0.f;
    if (c < C)
        for (int64_t r = r0 + ty; r < r1; r += 4) {
            const int b = (int)(r / T), t = (int)(r - (int64_t)b * T);
            if (t < len[b]) {
                const float d = x[r * C + c] - y[r * C + c];
                s = fmaf(d, d, s);
            }
        }
    ss[ty][threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && c < C)
        part[(size_t)blockIdx.y * C + c] = (ss[0][threadIdx.x] + ss[1][threadIdx.x]) + (ss[2][threadIdx.x] + ss[3][threadIdx.x]);
}
__device__ __forceinline__ float valid_frames(const int32_t *len, int B, int T)
{
    float v = 0.f;
    for (int b = 0; b < B; ++b) {
        int l = len[b];
        l = l < 0 ? 0 : (l > T ? T : l);
        v += (float)l;
    }
    return v;
}
// single block: out[0] = loss(g1), out[1] = loss(g2)
__global__ void mel_final_kernel(const float *__restrict__ part, const int32_t *__restrict__ len, int B, int T, int C, int nchunks, int g1,
                                 int g2, float *__restrict__ out)
{
    extern __shared__ float cs[];   // C
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nchunks; ++k) s += part[(size_t)k * C + c];
        cs[c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int g = threadIdx.x == 0 ? g1 : g2;
        const int G = C / g;
        const float cnt = valid_frames(len, B, T) * (float)g;
        float acc = 0.f;
        for (int q = 0; q < G; ++q) {
            float s = 0.f;
            for (int j = 0; j < g; ++j) s += cs[q * g + j];
            acc += s / (cnt + 1e-12f);
        }
        out[threadIdx.x] = acc / (float)G;
    }
}
__global__ void mel_bwd_kernel(const float *__restrict__ x, const float *__restrict__ y, const int32_t *__restrict__ len, int B, int T, int C,
                               int g1, int g2, float w1, float w2, const float *__restrict__ gout, float *__restrict__ dx)
{
    const float v = valid_frames(len, B, T);
    const float k1 = w1 * 2.0f / ((float)(C / g1) * (v * (float)g1 + 1e-12f));
    const float k2 = w2 * 2.0f / ((float)(C / g2) * (v * (float)g2 + 1e-12f));
    const float k = (k1 + k2) * gout[0];
    const int64_t total = (int64_t)B * T * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C;
        const int b = (int)(r / T), t = (int)(r - (int64_t)b * T);
        dx[i] = t < len[b] ? k * (x[i] - y[i]) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------ LSGAN + LeCam
// logits are (B, H, W) (single channel); the padded mask is (B, W), 1 = padded (loss mask = NOT padded).
struct LsSums { float a, b, c, d, e; };

__device__ __forceinline__ bool ls_valid(const uint8_t *padded, int64_t i, int HW, int W)
{
    const int64_t b = i / HW;
    const int w = (int)(i % W);
    return padded[b * W + w] == 0;
}

// single block of 1024 threads
__global__ void lsgan_d_fwd_kernel(const float *__restrict__ real, const float *__restrict__ fake, const uint8_t *__restrict__ padded, int B,
                                   int H, int W, float decay, float *__restrict__ state, float *__restrict__ out, float *__restrict__ saved)
{
    __shared__ float red[16];
    __shared__ float sh[4];
    const int64_t n = (int64_t)B * H * W;
    const int HW = H * W;
    float s_r1 = 0.f, s_f0 = 0.f, s_rx = 0.f, s_fx = 0.f, s_v = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024)
        if (ls_valid(padded, i, HW, W)) {
            const float r = real[i], f = fake[i];
            s_r1 = fmaf(r - 1.0f, r - 1.0f, s_r1);
            s_f0 = fmaf(f, f, s_f0);
            s_rx += r;
            s_fx += f;
            s_v += 1.0f;
        }
    s_r1 = block_sum<1024>(s_r1, red);
    s_f0 = block_sum<1024>(s_f0, red);
    s_rx = block_sum<1024>(s_rx, red);
    s_fx = block_sum<1024>(s_fx, red);
    s_v = block_sum<1024>(s_v, red);
    if (threadIdx.x == 0) {
        const float vc = fmaxf(s_v, 1.0f);
        const float rm = s_rx / vc, fm = s_fx / vc;
        float er, ef;
        if (state[2] == 0.f) {          // losses.py:52-55 first call initialises the EMAs
            er = rm;
            ef = fm;
            state[2] = 1.0f;
        } else {                        // :57-58
            er = state[0] * decay + (1.0f - decay) * rm;
            ef = state[1] * decay + (1.0f - decay) * fm;
        }
        state[0] = er;
        state[1] = ef;
        sh[0] = er;
        sh[1] = ef;
    }
    __syncthreads();
    const float er = sh[0], ef = sh[1];
    float l_r = 0.f, l_f = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024)
        if (ls_valid(padded, i, HW, W)) {
            const float a = fmaxf(real[i] - ef, 0.f), b = fmaxf(er - fake[i], 0.f);
            l_r = fmaf(a, a, l_r);
            l_f = fmaf(b, b, l_f);
        }
    l_r = block_sum<1024>(l_r, red);
    l_f = block_sum<1024>(l_f, red);
    if (threadIdx.x == 0) {
        const float vc = fmaxf(s_v, 1.0f);
        const float mse = s_v > 0.f ? 0.5f * (s_r1 / s_v + s_f0 / s_v) : 0.f;   // losses.py:29-32
        out[0] = mse + l_r / vc + l_f / vc;
        saved[0] = s_v;
        saved[1] = er;
        saved[2] = ef;
    }
}
__global__ void lsgan_d_bwd_kernel(const float *__restrict__ real, const float *__restrict__ fake, const uint8_t *__restrict__ padded, int B,
                                   int H, int W, const float *__restrict__ saved, const float *__restrict__ gout, float *__restrict__ dreal,
                                   float *__restrict__ dfake)
{
    const int64_t n = (int64_t)B * H * W;
    const int HW = H * W;
    const float v = saved[0], er = saved[1], ef = saved[2], g = gout[0];
    const float inv_mse = v > 0.f ? 1.0f / v : 0.f, inv_lc = 2.0f / fmaxf(v, 1.0f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float dr = 0.f, df = 0.f;
        if (ls_valid(padded, i, HW, W)) {
            const float r = real[i], f = fake[i];
            dr = g * ((r - 1.0f) * inv_mse + fmaxf(r - ef, 0.f) * inv_lc);
            df = g * (f * inv_mse - fmaxf(er - f, 0.f) * inv_lc);
        }
        dreal[i] = dr;
        dfake[i] = df;
    }
}
__global__ void lsgan_g_fwd_kernel(const float *__restrict__ fake, const uint8_t *__restrict__ padded, int B, int H, int W,
                                   float *__restrict__ out, float *__restrict__ saved)
{
    __shared__ float red[16];
    const int64_t n = (int64_t)B * H * W;
    const int HW = H * W;
    float s = 0.f, v = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024)
        if (ls_valid(padded, i, HW, W)) {
            const float d = fake[i] - 1.0f;
            s = fmaf(d, d, s);
            v += 1.0f;
        }
    s = block_sum<1024>(s, red);
    v = block_sum<1024>(v, red);
    if (threadIdx.x == 0) {
        out[0] = v > 0.f ? s / v : 0.f;
        saved[0] = v;
    }
}
__global__ void lsgan_g_bwd_kernel(const float *__restrict__ fake, const uint8_t *__restrict__ padded, int B, int H, int W,
                                   const float *__restrict__ saved, const float *__restrict__ gout, float *__restrict__ dfake)
{
    const int64_t n = (int64_t)B * H * W;
    const int HW = H * W;
    const float v = saved[0];
    const float k = v > 0.f ? 2.0f * gout[0] / v : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dfake[i] = ls_valid(padded, i, HW, W) ? k * (fake[i] - 1.0f) : 0.f;
}

// ------------------------------------------------------------------------------------------ masked MAE (feature matching)
// pred / target (B, H, W, C), padded (B, W)
__global__ void mae_partial_kernel(const float *__restrict__ p, const float *__restrict__ t, const uint8_t *__restrict__ padded, int B, int H,
                                   int W, int C, float *__restrict__ part)
{
    __shared__ float red[4];
    const int64_t n = (int64_t)B * H * W * C;
    float s = 0.f, v = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i / C;
        const int w = (int)(pix % W);
        const int64_t b = pix / ((int64_t)H * W);
        if (!padded[b * W + w]) {
            s += fabsf(p[i] - t[i]);
            v += 1.0f;
        }
    }
    s = block_sum<256>(s, red);
    v = block_sum<256>(v, red);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2] = s;
        part[blockIdx.x * 2 + 1] = v;
    }
}
__global__ void mae_final_kernel(const float *__restrict__ part, int nb, float *__restrict__ out)
{
    if (threadIdx.x != 0) return;
    float s = 0.f, v = 0.f;
    for (int i = 0; i < nb; ++i) {
        s += part[2 * i];
        v += part[2 * i + 1];
    }
    out[0] = s / (v + 1e-8f);
    out[1] = v;
}
__global__ void mae_bwd_kernel(const float *__restrict__ p, const float *__restrict__ t, const uint8_t *__restrict__ padded, int B, int H,
                               int W, int C, const float *__restrict__ out2, const float *__restrict__ gout, float *__restrict__ dp)
{
    const int64_t n = (int64_t)B * H * W * C;
    const float k = gout[0] / (out2[1] + 1e-8f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / C;
        const int w = (int)(pix % W);
        const int64_t b = pix / ((int64_t)H * W);
        float d = 0.f;
        if (!padded[b * W + w]) {
            const float e = p[i] - t[i];
            d = e > 0.f ? k : (e < 0.f ? -k : 0.f);
        }
        dp[i] = d;
    }
}

}  // namespace mq

using namespace mq;

extern "C" int mq_mel_loss_fwd(const float *x, const float *y, const int32_t *lengths, int B, int T, int C, int g1, int g2, float *out2,
                               float *workspace, mqStream stream)
{
    MQ_REQUIRE(x && y && lengths && out2 && workspace && B > 0 && T > 0 && C > 0, "mq_mel_loss_fwd: bad arguments");
    MQ_REQUIRE(g1 > 0 && g2 > 0 && C % g1 == 0 && C % g2 == 0, "mq_mel_loss_fwd: C=%d must be divisible by the group sizes %d, %d", C, g1, g2);
    MQ_REQUIRE(C <= 4096, "mq_mel_loss_fwd: C too large");
    const int nchunks = 64;
    hipLaunchKernelGGL(mel_partial_kernel, dim3((unsigned)cdiv(C, 64), nchunks), dim3(64, 4), 0, S(stream), x, y, lengths, B, T, C, nchunks,
                       workspace);
    hipLaunchKernelGGL(mel_final_kernel, dim3(1), dim3(256), C * sizeof(float), S(stream), workspace, lengths, B, T, C, nchunks, g1, g2, out2);
    MQ_LAUNCH_CHECK("mq_mel_loss_fwd");
    return 0;
}
extern "C" int mq_mel_loss_bwd(const float *x, const float *y, const int32_t *lengths, int B, int T, int C, int g1, int g2, float w1, float w2,
                               const float *gout, float *dx, mqStream stream)
{
    MQ_REQUIRE(x && y && lengths && gout && dx && B > 0 && T > 0 && C > 0 && g1 > 0 && g2 > 0 && C % g1 == 0 && C % g2 == 0,
               "mq_mel_loss_bwd: bad arguments");
    hipLaunchKernelGGL(mel_bwd_kernel, dim3(ew_grid((int64_t)B * T * C, 256)), dim3(256), 0, S(stream), x, y, lengths, B, T, C, g1, g2, w1, w2,
                       gout, dx);
    MQ_LAUNCH_CHECK("mq_mel_loss_bwd");
    return 0;
}
extern "C" int mq_lsgan_d_fwd(const float *real, const float *fake, const uint8_t *padded, int B, int H, int W, float decay, float *state,
                              float *out, float *saved, mqStream stream)
{
    MQ_REQUIRE(real && fake && padded && state && out && saved && B > 0 && H > 0 && W > 0, "mq_lsgan_d_fwd: bad arguments");
    hipLaunchKernelGGL(lsgan_d_fwd_kernel, dim3(1), dim3(1024), 0, S(stream), real, fake, padded, B, H, W, decay, state, out, saved);
    MQ_LAUNCH_CHECK("mq_lsgan_d_fwd");
    return 0;
}
extern "C" int mq_lsgan_d_bwd(const float *real, const float *fake, const uint8_t *padded, int B, int H, int W, const float *saved,
                              const float *gout, float *dreal, float *dfake, mqStream stream)
{
    MQ_REQUIRE(real && fake && padded && saved && gout && dreal && dfake && B > 0 && H > 0 && W > 0, "mq_lsgan_d_bwd: bad arguments");
    hipLaunchKernelGGL(lsgan_d_bwd_kernel, dim3(ew_grid((int64_t)B * H * W, 256)), dim3(256), 0, S(stream), real, fake, padded, B, H, W, saved,
                       gout, dreal, dfake);
    MQ_LAUNCH_CHECK("mq_lsgan_d_bwd");
    return 0;
}
extern "C" int mq_lsgan_g_fwd(const float *fake, const uint8_t *padded, int B, int H, int W, float *out, float *saved, mqStream stream)
{
    MQ_REQUIRE(fake && padded && out && saved && B > 0 && H > 0 && W > 0, "mq_lsgan_g_fwd: bad arguments");
    hipLaunchKernelGGL(lsgan_g_fwd_kernel, dim3(1), dim3(1024), 0, S(stream), fake, padded, B, H, W, out, saved);
    MQ_LAUNCH_CHECK("mq_lsgan_g_fwd");
    return 0;
}
extern "C" int mq_lsgan_g_bwd(const float *fake, const uint8_t *padded, int B, int H, int W, const float *saved, const float *gout,
                              float *dfake, mqStream stream)
{
    MQ_REQUIRE(fake && padded && saved && gout && dfake && B > 0 && H > 0 && W > 0, "mq_lsgan_g_bwd: bad arguments");
    hipLaunchKernelGGL(lsgan_g_bwd_kernel, dim3(ew_grid((int64_t)B * H * W, 256)), dim3(256), 0, S(stream), fake, padded, B, H, W, saved, gout,
                       dfake);
    MQ_LAUNCH_CHECK("mq_lsgan_g_bwd");
    return 0;
}
extern "C" int mq_masked_mae_fwd(const float *pred, const float *target, const uint8_t *padded, int B, int H, int W, int C, float *out2,
                                 float *workspace, mqStream stream)
{
    MQ_REQUIRE(pred && target && padded && out2 && workspace && B > 0 && H > 0 && W > 0 && C > 0, "mq_masked_mae_fwd: bad arguments");
    int nb = ew_grid((int64_t)B * H * W * C, 256);
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(mae_partial_kernel, dim3(nb), dim3(256), 0, S(stream), pred, target, padded, B, H, W, C, workspace);
    hipLaunchKernelGGL(mae_final_kernel, dim3(1), dim3(64), 0, S(stream), workspace, nb, out2);
    MQ_LAUNCH_CHECK("mq_masked_mae_fwd");
    return 0;
}
extern "C" int mq_masked_mae_bwd(const float *pred, const float *target, const uint8_t *padded, int B, int H, int W, int C, const float *out2,
                                 const float *gout, float *dpred, mqStream stream)
{
    MQ_REQUIRE(pred && target && padded && out2 && gout && dpred && B > 0 && H > 0 && W > 0 && C > 0, "mq_masked_mae_bwd: bad arguments");
    hipLaunchKernelGGL(mae_bwd_kernel, dim3(ew_grid((int64_t)B * H * W * C, 256)), dim3(256), 0, S(stream), pred, target, padded, B, H, W, C,
                       out2, gout, dpred);
    MQ_LAUNCH_CHECK("mq_masked_mae_bwd");
    return 0;
}
