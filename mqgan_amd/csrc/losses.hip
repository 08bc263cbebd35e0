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
// single block: sums[c] = sum over chunks of the partials (fixed order), sums[C] = number of valid frames
__global__ void mel_sums_kernel(const float *__restrict__ part, const int32_t *__restrict__ len, int B, int T, int C, int nchunks,
                                float *__restrict__ sums)
{
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nchunks; ++k) s += part[(size_t)k * C + c];
        sums[c] = s;
    }
    if (threadIdx.x == 0) sums[C] = valid_frames(len, B, T);
}
// out[0] = loss(g1), out[1] = loss(g2) from (possibly all-reduced) sums: mean over groups of sum_g / (frames * g + 1e-12)
__global__ void mel_final_kernel(const float *__restrict__ sums, int C, int g1, int g2, float *__restrict__ out)
{
    if (threadIdx.x < 2) {
        const int g = threadIdx.x == 0 ? g1 : g2;
        const int G = C / g;
        const float cnt = sums[C] * (float)g;
        float acc = 0.f;
        for (int q = 0; q < G; ++q) {
            float s = 0.f;
            for (int j = 0; j < g; ++j) s += sums[q * g + j];
            acc += s / (cnt + 1e-12f);
        }
        out[threadIdx.x] = acc / (float)G;
    }
}
__global__ void mel_bwd_kernel(const float *__restrict__ x, const float *__restrict__ y, const int32_t *__restrict__ len, int B, int T, int C,
                               int g1, int g2, float w1, float w2, const float *__restrict__ sums, const float *__restrict__ gout,
                               float *__restrict__ dx)
{
    const float v = sums[C];                       // valid frames of the (global) batch
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
// Logits are (B, H, W) (single channel).  Masks say which elements are PADDED (1): mode 0 = (B, W) frame mask, constant
// along H (what the discriminators produce); mode 1 = one byte per element (B, H, W) (arbitrary masks, losses.py:26-32);
// a null mask means "everything valid".  Every loss is a ratio of sums over the batch; the kernels produce the sums,
// the caller may all-reduce them over the data-parallel ranks (global normalisers), and the *_final /
// backward kernels work from the reduced values.
__device__ __forceinline__ bool ls_valid(const uint8_t *padded, int mode, int64_t i, int HW, int W)
{
    if (padded == nullptr) return true;
    if (mode == 1) return padded[i] == 0;
    const int64_t b = i / HW;
    const int w = (int)(i % W);
    return padded[b * W + w] == 0;
}

// single block of 1024: sums6 = [S m_r (r-1)^2, S m_f f^2, S m_r r, S m_f f, S m_r, S m_f]
__global__ __launch_bounds__(1024) void lsgan_d_sums_kernel(const float *__restrict__ real, const float *__restrict__ fake,
                                                            const uint8_t *__restrict__ rpad, const uint8_t *__restrict__ fpad, int mode,
                                                            int B, int H, int W, float *__restrict__ sums6)
{
    __shared__ float red[16];
    const int64_t n = (int64_t)B * H * W;
    const int HW = H * W;
    float s_r1 = 0.f, s_f0 = 0.f, s_rx = 0.f, s_fx = 0.f, s_vr = 0.f, s_vf = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        if (ls_valid(rpad, mode, i, HW, W)) {
            const float r = real[i];
            s_r1 = fmaf(r - 1.0f, r - 1.0f, s_r1);
            s_rx += r;
            s_vr += 1.0f;
        }
        if (ls_valid(fpad, mode, i, HW, W)) {
            const float f = fake[i];
            s_f0 = fmaf(f, f, s_f0);
            s_fx += f;
            s_vf += 1.0f;
        }
    }
    s_r1 = block_sum<1024>(s_r1, red);
    s_f0 = block_sum<1024>(s_f0, red);
    s_rx = block_sum<1024>(s_rx, red);
    s_fx = block_sum<1024>(s_fx, red);
    s_vr = block_sum<1024>(s_vr, red);
    s_vf = block_sum<1024>(s_vf, red);
    if (threadIdx.x == 0) {
        sums6[0] = s_r1; sums6[1] = s_f0; sums6[2] = s_rx; sums6[3] = s_fx; sums6[4] = s_vr; sums6[5] = s_vf;
    }
}
// EMA chain of n consecutive discriminator_loss calls (losses.py:36-58): state = [ema_real, ema_fake, initialised, -];
// ema[i] = the EMAs right after call i's update (what call i's LeCam term uses)
__global__ void lsgan_ema_kernel(const float *__restrict__ sums6, int n, float decay, float *__restrict__ state, float *__restrict__ ema)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float er = state[0], ef = state[1];
    bool init = state[2] != 0.f;
    for (int i = 0; i < n; ++i) {
        const float *s = sums6 + 6 * i;
        const float rm = s[2] / fmaxf(s[4], 1.0f), fm = s[3] / fmaxf(s[5], 1.0f);
        if (!init) {                                 // losses.py:52-55: the first call initialises the EMAs
            er = rm;
            ef = fm;
            init = true;
        } else {                                     // :57-58
            er = er * decay + (1.0f - decay) * rm;
            ef = ef * decay + (1.0f - decay) * fm;
        }
        ema[2 * i] = er;
        ema[2 * i + 1] = ef;
    }
    state[0] = er;
    state[1] = ef;
    state[2] = init ? 1.0f : 0.f;
}
// lsums2 = [S m_r relu(r - ema_f)^2, S m_f relu(ema_r - f)^2]
__global__ __launch_bounds__(1024) void lsgan_d_lecam_kernel(const float *__restrict__ real, const float *__restrict__ fake,
                                                             const uint8_t *__restrict__ rpad, const uint8_t *__restrict__ fpad, int mode,
                                                             int B, int H, int W, const float *__restrict__ ema2, float *__restrict__ lsums2)
{
    __shared__ float red[16];
    const int64_t n = (int64_t)B * H * W;
    const int HW = H * W;
    const float er = ema2[0], ef = ema2[1];
    float l_r = 0.f, l_f = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        if (ls_valid(rpad, mode, i, HW, W)) {
            const float a = fmaxf(real[i] - ef, 0.f);
            l_r = fmaf(a, a, l_r);
        }
        if (ls_valid(fpad, mode, i, HW, W)) {
            const float b = fmaxf(er - fake[i], 0.f);
            l_f = fmaf(b, b, l_f);
        }
    }
    l_r = block_sum<1024>(l_r, red);
    l_f = block_sum<1024>(l_f, red);
    if (threadIdx.x == 0) {
        lsums2[0] = l_r;
        lsums2[1] = l_f;
    }
}
__global__ void lsgan_d_final_kernel(const float *__restrict__ s, const float *__restrict__ l, float *__restrict__ out)
{
    if (threadIdx.x != 0) return;
    const float real_loss = s[4] > 0.f ? s[0] / s[4] : 0.f;          // losses.py:29-32
    const float fake_loss = s[5] > 0.f ? s[1] / s[5] : 0.f;
    out[0] = 0.5f * (real_loss + fake_loss) + l[0] / fmaxf(s[4], 1.0f) + l[1] / fmaxf(s[5], 1.0f);
}
__global__ void lsgan_d_bwd_kernel(const float *__restrict__ real, const float *__restrict__ fake, const uint8_t *__restrict__ rpad,
                                   const uint8_t *__restrict__ fpad, int mode, int B, int H, int W, const float *__restrict__ s,
                                   const float *__restrict__ ema2, const float *__restrict__ gout, float *__restrict__ dreal,
                                   float *__restrict__ dfake)
{
    const int64_t n = (int64_t)B * H * W;
    const int HW = H * W;
    const float vr = s[4], vf = s[5], er = ema2[0], ef = ema2[1], g = gout[0];
    const float mr = vr > 0.f ? 1.0f / vr : 0.f, mf = vf > 0.f ? 1.0f / vf : 0.f;
    const float lr = 2.0f / fmaxf(vr, 1.0f), lf = 2.0f / fmaxf(vf, 1.0f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float dr = 0.f, df = 0.f;
        if (ls_valid(rpad, mode, i, HW, W)) {
            const float r = real[i];
            dr = g * ((r - 1.0f) * mr + fmaxf(r - ef, 0.f) * lr);
        }
        if (ls_valid(fpad, mode, i, HW, W)) {
            const float f = fake[i];
            df = g * (f * mf - fmaxf(er - f, 0.f) * lf);
        }
        dreal[i] = dr;
        dfake[i] = df;
    }
}
// generator loss: sums2 = [S m (f-1)^2, S m]
__global__ __launch_bounds__(1024) void lsgan_g_sums_kernel(const float *__restrict__ fake, const uint8_t *__restrict__ pad, int mode, int B,
                                                            int H, int W, float *__restrict__ sums2)
{
    __shared__ float red[16];
    const int64_t n = (int64_t)B * H * W;
    const int HW = H * W;
    float s = 0.f, v = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024)
        if (ls_valid(pad, mode, i, HW, W)) {
            const float d = fake[i] - 1.0f;
            s = fmaf(d, d, s);
            v += 1.0f;
        }
    s = block_sum<1024>(s, red);
    v = block_sum<1024>(v, red);
    if (threadIdx.x == 0) {
        sums2[0] = s;
        sums2[1] = v;
    }
}
// out = v > 0 ? s / v : 0   (losses.py:29-32);  eps > 0: out = s / (v + eps)   (train.py:45)
__global__ void ratio_final_kernel(const float *__restrict__ sums2, float eps, float *__restrict__ out)
{
    if (threadIdx.x != 0) return;
    const float s = sums2[0], v = sums2[1];
    out[0] = eps > 0.f ? s / (v + eps) : (v > 0.f ? s / v : 0.f);
}
__global__ void lsgan_g_bwd_kernel(const float *__restrict__ fake, const uint8_t *__restrict__ pad, int mode, int B, int H, int W,
                                   const float *__restrict__ sums2, const float *__restrict__ gout, float *__restrict__ dfake)
{
    const int64_t n = (int64_t)B * H * W;
    const int HW = H * W;
    const float v = sums2[1];
    const float k = v > 0.f ? 2.0f * gout[0] / v : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dfake[i] = ls_valid(pad, mode, i, HW, W) ? k * (fake[i] - 1.0f) : 0.f;
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
__global__ void mae_sums_kernel(const float *__restrict__ part, int nb, float *__restrict__ sums2)
{
    if (threadIdx.x != 0) return;
    float s = 0.f, v = 0.f;
    for (int i = 0; i < nb; ++i) {
        s += part[2 * i];
        v += part[2 * i + 1];
    }
    sums2[0] = s;
    sums2[1] = v;
}
__global__ void mae_bwd_kernel(const float *__restrict__ p, const float *__restrict__ t, const uint8_t *__restrict__ padded, int B, int H,
                               int W, int C, const float *__restrict__ sums2, const float *__restrict__ gout, float leaky_slope,
                               float *__restrict__ dp)
{
    // leaky_slope != 1: p = leaky(h) of a layer that takes its gradient with respect to h (mq_conv_dgrad_gated): d *= leaky'(p)
    const int64_t n = (int64_t)B * H * W * C;
    const float k = gout[0] / (sums2[1] + 1e-8f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / C;
        const int w = (int)(pix % W);
        const int64_t b = pix / ((int64_t)H * W);
        float d = 0.f;
        if (!padded[b * W + w]) {
            const float pv = p[i];
            const float e = pv - t[i];
            d = e > 0.f ? k : (e < 0.f ? -k : 0.f);
            d = pv > 0.f ? d : leaky_slope * d;
        }
        dp[i] = d;
    }
}

}  // namespace mq

using namespace mq;

extern "C" int mq_mel_loss_sums(const float *x, const float *y, const int32_t *lengths, int B, int T, int C, float *sums, float *workspace,
                                mqStream stream)
{
    MQ_REQUIRE(x && y && lengths && sums && workspace && B > 0 && T > 0 && C > 0, "mq_mel_loss_sums: bad arguments");
    MQ_REQUIRE(C <= 4096, "mq_mel_loss_sums: C too large");
    const int nchunks = 64;
    hipLaunchKernelGGL(mel_partial_kernel, dim3((unsigned)cdiv(C, 64), nchunks), dim3(64, 4), 0, S(stream), x, y, lengths, B, T, C, nchunks,
                       workspace);
    hipLaunchKernelGGL(mel_sums_kernel, dim3(1), dim3(256), 0, S(stream), workspace, lengths, B, T, C, nchunks, sums);
    MQ_LAUNCH_CHECK("mq_mel_loss_sums");
    return 0;
}
extern "C" int mq_mel_loss_final(const float *sums, int C, int g1, int g2, float *out2, mqStream stream)
{
    MQ_REQUIRE(sums && out2 && C > 0, "mq_mel_loss_final: bad arguments");
    MQ_REQUIRE(g1 > 0 && g2 > 0 && C % g1 == 0 && C % g2 == 0, "mq_mel_loss_final: C=%d must be divisible by the group sizes %d, %d", C, g1, g2);
    hipLaunchKernelGGL(mel_final_kernel, dim3(1), dim3(64), 0, S(stream), sums, C, g1, g2, out2);
    MQ_LAUNCH_CHECK("mq_mel_loss_final");
    return 0;
}
extern "C" int mq_mel_loss_bwd(const float *x, const float *y, const int32_t *lengths, int B, int T, int C, int g1, int g2, float w1, float w2,
                               const float *sums, const float *gout, float *dx, mqStream stream)
{
    MQ_REQUIRE(x && y && lengths && sums && gout && dx && B > 0 && T > 0 && C > 0 && g1 > 0 && g2 > 0 && C % g1 == 0 && C % g2 == 0,
               "mq_mel_loss_bwd: bad arguments");
    hipLaunchKernelGGL(mel_bwd_kernel, dim3(ew_grid((int64_t)B * T * C, 256)), dim3(256), 0, S(stream), x, y, lengths, B, T, C, g1, g2, w1, w2,
                       sums, gout, dx);
    MQ_LAUNCH_CHECK("mq_mel_loss_bwd");
    return 0;
}

static int ls_args(const void *a, int mode, int B, int H, int W, const char *who)
{
    MQ_REQUIRE(a && B > 0 && H > 0 && W > 0, "%s: bad arguments", who);
    MQ_REQUIRE(mode == 0 || mode == 1, "%s: mask mode must be 0 (frame mask (B, W)) or 1 (per element (B, H, W))", who);
    return 0;
}
extern "C" int mq_lsgan_d_sums(const float *real, const float *fake, const uint8_t *real_padded, const uint8_t *fake_padded, int mask_mode,
                               int B, int H, int W, float *sums6, mqStream stream)
{
    if (ls_args(real, mask_mode, B, H, W, "mq_lsgan_d_sums")) return 1;
    MQ_REQUIRE(fake && sums6, "mq_lsgan_d_sums: null pointer");
    hipLaunchKernelGGL(lsgan_d_sums_kernel, dim3(1), dim3(1024), 0, S(stream), real, fake, real_padded, fake_padded, mask_mode, B, H, W, sums6);
    MQ_LAUNCH_CHECK("mq_lsgan_d_sums");
    return 0;
}
extern "C" int mq_lsgan_ema(const float *sums6, int n, float decay, float *state, float *ema_out, mqStream stream)
{
    MQ_REQUIRE(sums6 && state && ema_out && n > 0, "mq_lsgan_ema: bad arguments");
    hipLaunchKernelGGL(lsgan_ema_kernel, dim3(1), dim3(64), 0, S(stream), sums6, n, decay, state, ema_out);
    MQ_LAUNCH_CHECK("mq_lsgan_ema");
    return 0;
}
extern "C" int mq_lsgan_d_lecam(const float *real, const float *fake, const uint8_t *real_padded, const uint8_t *fake_padded, int mask_mode,
                                int B, int H, int W, const float *ema2, float *lsums2, mqStream stream)
{
    if (ls_args(real, mask_mode, B, H, W, "mq_lsgan_d_lecam")) return 1;
    MQ_REQUIRE(fake && ema2 && lsums2, "mq_lsgan_d_lecam: null pointer");
    hipLaunchKernelGGL(lsgan_d_lecam_kernel, dim3(1), dim3(1024), 0, S(stream), real, fake, real_padded, fake_padded, mask_mode, B, H, W, ema2,
                       lsums2);
    MQ_LAUNCH_CHECK("mq_lsgan_d_lecam");
    return 0;
}
extern "C" int mq_lsgan_d_final(const float *sums6, const float *lsums2, float *out, mqStream stream)
{
    MQ_REQUIRE(sums6 && lsums2 && out, "mq_lsgan_d_final: null pointer");
    hipLaunchKernelGGL(lsgan_d_final_kernel, dim3(1), dim3(64), 0, S(stream), sums6, lsums2, out);
    MQ_LAUNCH_CHECK("mq_lsgan_d_final");
    return 0;
}
extern "C" int mq_lsgan_d_bwd(const float *real, const float *fake, const uint8_t *real_padded, const uint8_t *fake_padded, int mask_mode, int B,
                              int H, int W, const float *sums6, const float *ema2, const float *gout, float *dreal, float *dfake,
                              mqStream stream)
{
    if (ls_args(real, mask_mode, B, H, W, "mq_lsgan_d_bwd")) return 1;
    MQ_REQUIRE(fake && sums6 && ema2 && gout && dreal && dfake, "mq_lsgan_d_bwd: null pointer");
    hipLaunchKernelGGL(lsgan_d_bwd_kernel, dim3(ew_grid((int64_t)B * H * W, 256)), dim3(256), 0, S(stream), real, fake, real_padded, fake_padded,
                       mask_mode, B, H, W, sums6, ema2, gout, dreal, dfake);
    MQ_LAUNCH_CHECK("mq_lsgan_d_bwd");
    return 0;
}
extern "C" int mq_lsgan_g_sums(const float *fake, const uint8_t *padded, int mask_mode, int B, int H, int W, float *sums2, mqStream stream)
{
    if (ls_args(fake, mask_mode, B, H, W, "mq_lsgan_g_sums")) return 1;
    MQ_REQUIRE(sums2, "mq_lsgan_g_sums: null pointer");
    hipLaunchKernelGGL(lsgan_g_sums_kernel, dim3(1), dim3(1024), 0, S(stream), fake, padded, mask_mode, B, H, W, sums2);
    MQ_LAUNCH_CHECK("mq_lsgan_g_sums");
    return 0;
}
extern "C" int mq_ratio_final(const float *sums2, float eps, float *out, mqStream stream)
{
    MQ_REQUIRE(sums2 && out && eps >= 0.f, "mq_ratio_final: bad arguments");
    hipLaunchKernelGGL(ratio_final_kernel, dim3(1), dim3(64), 0, S(stream), sums2, eps, out);
    MQ_LAUNCH_CHECK("mq_ratio_final");
    return 0;
}
extern "C" int mq_lsgan_g_bwd(const float *fake, const uint8_t *padded, int mask_mode, int B, int H, int W, const float *sums2,
                              const float *gout, float *dfake, mqStream stream)
{
    if (ls_args(fake, mask_mode, B, H, W, "mq_lsgan_g_bwd")) return 1;
    MQ_REQUIRE(sums2 && gout && dfake, "mq_lsgan_g_bwd: null pointer");
    hipLaunchKernelGGL(lsgan_g_bwd_kernel, dim3(ew_grid((int64_t)B * H * W, 256)), dim3(256), 0, S(stream), fake, padded, mask_mode, B, H, W,
                       sums2, gout, dfake);
    MQ_LAUNCH_CHECK("mq_lsgan_g_bwd");
    return 0;
}
extern "C" int mq_masked_mae_sums(const float *pred, const float *target, const uint8_t *padded, int B, int H, int W, int C, float *sums2,
                                  float *workspace, mqStream stream)
{
    MQ_REQUIRE(pred && target && padded && sums2 && workspace && B > 0 && H > 0 && W > 0 && C > 0, "mq_masked_mae_sums: bad arguments");
    int nb = ew_grid((int64_t)B * H * W * C, 256);
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(mae_partial_kernel, dim3(nb), dim3(256), 0, S(stream), pred, target, padded, B, H, W, C, workspace);
    hipLaunchKernelGGL(mae_sums_kernel, dim3(1), dim3(64), 0, S(stream), workspace, nb, sums2);
    MQ_LAUNCH_CHECK("mq_masked_mae_sums");
    return 0;
}
extern "C" int mq_masked_mae_bwd(const float *pred, const float *target, const uint8_t *padded, int B, int H, int W, int C, const float *sums2,
                                 const float *gout, float leaky_slope, float *dpred, mqStream stream)
{
    MQ_REQUIRE(pred && target && padded && sums2 && gout && dpred && B > 0 && H > 0 && W > 0 && C > 0, "mq_masked_mae_bwd: bad arguments");
    hipLaunchKernelGGL(mae_bwd_kernel, dim3(ew_grid((int64_t)B * H * W * C, 256)), dim3(256), 0, S(stream), pred, target, padded, B, H, W, C,
                       sums2, gout, leaky_slope, dpred);
    MQ_LAUNCH_CHECK("mq_masked_mae_bwd");
    return 0;
}
