// CBAM1D (attentions.py:217-273, 310-365, 393-419) and the masked SE block of the patch
// discriminator (discriminators.py:40-67), channels-last.  All bandwidth-bound reductions.
//
// Reference quirk reproduced: the helper masked_fill_ is not in place, so the
// channel max-pool runs over padded frames too, SAM's features / logits / map are unmasked and the
// CBAM output is unmasked.  Only the average pool is masked.
#include "common.h"

namespace mq {

// grid (ceil(C/64), B), block (64, 4).  x (B, T, C).
__global__ void cam_pool_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask, int B, int T, int C,
                                float *__restrict__ maxv, int32_t *__restrict__ argmax_t, float *__restrict__ avgv)
{
    __shared__ float smax[4][64], ssum[4][64];
    __shared__ int sarg[4][64], scnt[4][64];
    const int c = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y, ty = threadIdx.y;
    float mx = -INFINITY, sum = 0.f;
    int am = 0, cnt = 0;
    if (c < C)
        for (int t = ty; t < T; t += 4) {
            const float v = x[((size_t)b * T + t) * C + c];
            if (v > mx) { mx = v; am = t; }
            if (!mask[(size_t)b * T + t]) { sum += v; ++cnt; }
        }
    smax[ty][threadIdx.x] = mx; sarg[ty][threadIdx.x] = am; ssum[ty][threadIdx.x] = sum; scnt[ty][threadIdx.x] = cnt;
    __syncthreads();
    if (ty == 0 && c < C) {
        for (int k = 1; k < 4; ++k) {
            const float o = smax[k][threadIdx.x];
            const int oa = sarg[k][threadIdx.x];
            if (o > mx || (o == mx && oa < am)) { mx = o; am = oa; }
            sum += ssum[k][threadIdx.x];
            cnt += scnt[k][threadIdx.x];
        }
        maxv[(size_t)b * C + c] = mx;
        argmax_t[(size_t)b * C + c] = am;
        avgv[(size_t)b * C + c] = sum / (float)(cnt > 0 ? cnt : 1);
    }
}

// one wave per (b, t) row: max / mean over c of gate*x
__global__ void sam_pool_kernel(const float *__restrict__ x, const float *__restrict__ gate, int B, int T, int C,
                                float *__restrict__ feat, int32_t *__restrict__ argmax_c)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= (int64_t)B * T) return;
    const int b = (int)(row / T);
    const float *xr = x + row * C, *gr = gate + (size_t)b * C;
    float mx = -INFINITY, sum = 0.f;
    int am = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float v = xr[c] * gr[c];
        if (v > mx) { mx = v; am = c; }
        sum += v;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_xor(mx, off, 64);
        const int oa = __shfl_xor(am, off, 64);
        if (o > mx || (o == mx && oa < am)) { mx = o; am = oa; }
    }
    sum = wave_sum(sum);
    if (lane == 0) {
        feat[row * 2 + 0] = mx;
        feat[row * 2 + 1] = sum / (float)C;
        argmax_c[row] = am;
    }
}

// s[b,t] = sigmoid(conv_ks(feat)), zero padding ks/2 each side.  w7 is Conv1d(2,1,ks).weight[0] = (2, ks)
__global__ void sam_gate_kernel(const float *__restrict__ feat, const float *__restrict__ w7, int ks, int B, int T,
                                float *__restrict__ s)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const int b = i / T, t = i - b * T, half = ks / 2;
    float a = 0.f;
    for (int j = 0; j < ks; ++j) {
        const int tt = t + j - half;
        if (tt >= 0 && tt < T) {
            a = fmaf(w7[j], feat[((size_t)b * T + tt) * 2 + 0], a);
            a = fmaf(w7[ks + j], feat[((size_t)b * T + tt) * 2 + 1], a);
        }
    }
    s[i] = 1.0f / (1.0f + __expf(-a));
}
__global__ void sam_apply_kernel(const float *__restrict__ x, const float *__restrict__ gate, const float *__restrict__ s, int64_t rows,
                                 int T, int C, float *__restrict__ out)
{
    const int64_t total = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C;
        const int c = (int)(i - r * C);
        const int b = (int)(r / T);
        const float xv = x[i];
        out[i] = fmaf(s[r] * gate[(size_t)b * C + c], xv, xv);
    }
}

// bwd a: dlogit[b,t] = (sum_c dout*gate*x) * s(1-s)        (one wave per row)
__global__ void cbam_bwd_ds_kernel(const float *__restrict__ dout, const float *__restrict__ x, const float *__restrict__ gate,
                                   const float *__restrict__ s, int B, int T, int C, float *__restrict__ dlogit)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= (int64_t)B * T) return;
    const int b = (int)(row / T);
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a = fmaf(dout[row * C + c] * gate[(size_t)b * C + c], x[row * C + c], a);
    a = wave_sum(a);
    if (lane == 0) {
        const float sv = s[row];
        dlogit[row] = a * sv * (1.0f - sv);
    }
}
// bwd b: dfeat = conv^T(dlogit);  dw7 partials per block
__global__ void cbam_bwd_conv_kernel(const float *__restrict__ dlogit, const float *__restrict__ feat, const float *__restrict__ w7,
                                     int ks, int B, int T, float *__restrict__ dfeat, float *__restrict__ part)
{
    __shared__ float red[4];
    const int half = ks / 2;
    float gw[2][16];
#pragma unroll
    for (int j = 0; j < 16; ++j) gw[0][j] = gw[1][j] = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B * T; i += gridDim.x * 256) {
        const int b = i / T, t = i - b * T;
        const float dl = dlogit[i];
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < ks) {
                // logit[t] uses feat[t + j - half]  =>  dfeat[t] += w[j] * dlogit[t - j + half]
                const int ts = t - j + half;
                if (ts >= 0 && ts < T) {
                    const float dls = dlogit[(size_t)b * T + ts];
                    d0 = fmaf(w7[j], dls, d0);
                    d1 = fmaf(w7[ks + j], dls, d1);
                }
                const int tf = t + j - half;
                if (tf >= 0 && tf < T) {
                    gw[0][j] = fmaf(dl, feat[((size_t)b * T + tf) * 2 + 0], gw[0][j]);
                    gw[1][j] = fmaf(dl, feat[((size_t)b * T + tf) * 2 + 1], gw[1][j]);
                }
            }
        }
        dfeat[(size_t)i * 2 + 0] = d0;
        dfeat[(size_t)i * 2 + 1] = d1;
    }
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < ks) {
                const float v = block_sum<256>(gw[ch][j], red);
                if (threadIdx.x == 0) part[(size_t)blockIdx.x * 32 + ch * 16 + j] = v;
            }
        }
}
__global__ void cbam_bwd_w7_kernel(const float *__restrict__ part, int nblocks, int ks, float *__restrict__ dw7)
{
    const int i = threadIdx.x;   // 0..31
    if (i >= 32) return;
    const int ch = i / 16, j = i % 16;
    if (j >= ks) return;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * 32 + i];
    dw7[ch * ks + j] = s;
}
// bwd c: dx = dout + dxc*gate ; dgate[b,c] = sum_t dxc * x ;  dxc = dout*s + dfeat_mean/C + [c == argmax] dfeat_max
// grid (ceil(C/64), B), block (64, 4)
__global__ void cbam_bwd_x_kernel(const float *__restrict__ dout, const float *__restrict__ x, const float *__restrict__ gate,
                                  const float *__restrict__ s, const float *__restrict__ dfeat, const int32_t *__restrict__ argmax_c,
                                  int B, int T, int C, float *__restrict__ dx, float *__restrict__ dgate)
{
    __shared__ float sg[4][64];
    const int c = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y, ty = threadIdx.y;
    float acc = 0.f;
    if (c < C) {
        const float gv = gate[(size_t)b * C + c];
        const float invC = 1.0f / (float)C;
        for (int t = ty; t < T; t += 4) {
            const size_t r = (size_t)b * T + t;
            const float dov = dout[r * C + c], xv = x[r * C + c];
            float dxc = dov * s[r] + dfeat[r * 2 + 1] * invC;
            if (argmax_c[r] == c) dxc += dfeat[r * 2 + 0];
            acc = fmaf(dxc, xv, acc);
            dx[r * C + c] = fmaf(dxc, gv, dov);
        }
    }
    sg[ty][threadIdx.x] = acc;
    __syncthreads();
    if (ty == 0 && c < C) dgate[(size_t)b * C + c] = (sg[0][threadIdx.x] + sg[1][threadIdx.x]) + (sg[2][threadIdx.x] + sg[3][threadIdx.x]);
}
// dx[b, argmax_t[b,c], c] += dmax[b,c] ; dx[b,t,c] += davg[b,c]/count_b on valid t
__global__ void cam_pool_bwd_kernel(const float *__restrict__ dmax, const float *__restrict__ davg, const int32_t *__restrict__ argmax_t,
                                    const uint8_t *__restrict__ mask, int B, int T, int C, float *__restrict__ dx)
{
    __shared__ int scnt;
    const int c = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y, ty = threadIdx.y;
    if (threadIdx.x == 0 && ty == 0) {
        int n = 0;
        for (int t = 0; t < T; ++t) n += mask[(size_t)b * T + t] ? 0 : 1;
        scnt = n > 0 ? n : 1;
    }
    __syncthreads();
    if (c >= C) return;
    const float da = davg[(size_t)b * C + c] / (float)scnt;
    const float dm = dmax[(size_t)b * C + c];
    const int am = argmax_t[(size_t)b * C + c];
    for (int t = ty; t < T; t += 4) {
        const size_t o = ((size_t)b * T + t) * C + c;
        float v = dx[o];
        if (!mask[(size_t)b * T + t]) v += da;
        if (t == am) v += dm;
        dx[o] = v;
    }
}

// ------------------------------------------------------------------------------------------ masked SE
// x (B, H, W, C); padded mask (B, W).  grid (ceil(C/64), B), block (64, 4)
__global__ void se_squeeze_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask, int B, int H, int W, int C,
                                  float *__restrict__ sq)
{
    __shared__ float ss[4][64];
    __shared__ int scnt[4][64];
    const int c = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y, ty = threadIdx.y;
    float sum = 0.f;
    int cnt = 0;
    if (c < C)
        for (int p = ty; p < H * W; p += 4) {
            const int w = p % W;
            if (!mask[(size_t)b * W + w]) {
                sum += x[((size_t)b * H * W + p) * C + c];
                ++cnt;
            }
        }
    ss[ty][threadIdx.x] = sum;
    scnt[ty][threadIdx.x] = cnt;
    __syncthreads();
    if (ty == 0 && c < C) {
        const float s = (ss[0][threadIdx.x] + ss[1][threadIdx.x]) + (ss[2][threadIdx.x] + ss[3][threadIdx.x]);
        const int n = scnt[0][threadIdx.x] + scnt[1][threadIdx.x] + scnt[2][threadIdx.x] + scnt[3][threadIdx.x];
        sq[(size_t)b * C + c] = s / (float)(n > 0 ? n : 1);
    }
}
__global__ void se_scale_fwd_kernel(const float *__restrict__ x, const float *__restrict__ e, int64_t B, int64_t HW, int C,
                                    float *__restrict__ y)
{
    const int64_t total = B * HW * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t b = i / (HW * C);
        y[i] = x[i] * e[b * C + c];
    }
}
// dx = dy*e ; dexcite[b,c] = sum_p dy*x.   grid (ceil(C/64), B), block (64, 4)
__global__ void se_scale_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ e, int B, int HW,
                                    int C, float *__restrict__ dx, float *__restrict__ de)
{
    __shared__ float ss[4][64];
    const int c = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y, ty = threadIdx.y;
    float acc = 0.f;
    if (c < C) {
        const float ev = e[(size_t)b * C + c];
        for (int p = ty; p < HW; p += 4) {
            const size_t o = ((size_t)b * HW + p) * C + c;
            const float g = dy[o];
            acc = fmaf(g, x[o], acc);
            dx[o] = g * ev;
        }
    }
    ss[ty][threadIdx.x] = acc;
    __syncthreads();
    if (ty == 0 && c < C) de[(size_t)b * C + c] = (ss[0][threadIdx.x] + ss[1][threadIdx.x]) + (ss[2][threadIdx.x] + ss[3][threadIdx.x]);
}
__global__ void se_squeeze_bwd_kernel(const float *__restrict__ dsq, const uint8_t *__restrict__ mask, int B, int H, int W, int C,
                                      float *__restrict__ dx)
{
    __shared__ int scnt;
    const int c = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y, ty = threadIdx.y;
    if (threadIdx.x == 0 && ty == 0) {
        int n = 0;
        for (int w = 0; w < W; ++w) n += mask[(size_t)b * W + w] ? 0 : 1;
        n *= H;
        scnt = n > 0 ? n : 1;
    }
    __syncthreads();
    if (c >= C) return;
    const float d = dsq[(size_t)b * C + c] / (float)scnt;
    for (int p = ty; p < H * W; p += 4) {
        const int w = p % W;
        if (!mask[(size_t)b * W + w]) dx[((size_t)b * H * W + p) * C + c] += d;
    }
}

// Vector forms (C % 4 == 0): block = 16 channel quads x 16 row groups, float4 per lane, 16 rows of every channel in flight per
// block instead of 4 scalar ones (r3: these passes ran at a third of the HBM rate, 0.5 ms per step at config 2).  Summation order
// per channel: row groups in parallel, combined in a fixed order - deterministic.
__global__ __launch_bounds__(256) void se_squeeze_v4_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask, int B, int H, int W,
                                                            int C, float *__restrict__ sq)
{
    __shared__ float4 ss[16][16];
    __shared__ int scnt[16];
    const int c4 = blockIdx.x * 16 + threadIdx.x, b = blockIdx.y, ty = threadIdx.y, C4 = C >> 2;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;
    for (int p = ty; p < H * W; p += 16) {
        const int w = p % W;
        if (!mask[(size_t)b * W + w]) {
            ++cnt;
            if (c4 < C4) {
                const float4 v = x4[((size_t)b * H * W + p) * C4 + c4];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
        }
    }
    ss[ty][threadIdx.x] = sum;
    if (threadIdx.x == 0) scnt[ty] = cnt;
    __syncthreads();
    if (ty == 0 && c4 < C4) {
        float4 t = ss[0][threadIdx.x];
        int n = scnt[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) {
            const float4 v = ss[i][threadIdx.x];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            n += scnt[i];
        }
        const float nf = (float)(n > 0 ? n : 1);
        reinterpret_cast<float4 *>(sq)[(size_t)b * C4 + c4] = make_float4(t.x / nf, t.y / nf, t.z / nf, t.w / nf);
    }
}
__global__ __launch_bounds__(256) void se_squeeze_bwd_v4_kernel(const float *__restrict__ dsq, const uint8_t *__restrict__ mask, int B, int H,
                                                                int W, int C, float *__restrict__ dx)
{
    __shared__ int scnt;
    const int c4 = blockIdx.x * 16 + threadIdx.x, b = blockIdx.y, ty = threadIdx.y, C4 = C >> 2;
    if (threadIdx.x == 0 && ty == 0) {
        int n = 0;
        for (int w = 0; w < W; ++w) n += mask[(size_t)b * W + w] ? 0 : 1;
        n *= H;
        scnt = n > 0 ? n : 1;
    }
    __syncthreads();
    if (c4 >= C4) return;
    const float nf = (float)scnt;
    float4 d = reinterpret_cast<const float4 *>(dsq)[(size_t)b * C4 + c4];
    d = make_float4(d.x / nf, d.y / nf, d.z / nf, d.w / nf);
    float4 *dx4 = reinterpret_cast<float4 *>(dx);
    for (int p = ty; p < H * W; p += 16) {
        const int w = p % W;
        if (!mask[(size_t)b * W + w]) {
            float4 v = dx4[((size_t)b * H * W + p) * C4 + c4];
            v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
            dx4[((size_t)b * H * W + p) * C4 + c4] = v;
        }
    }
}

}  // namespace mq

using namespace mq;

#define MQ_BC_OK(B, C) ((B) > 0 && (B) <= 65535 && (C) > 0)

extern "C" int mq_cam_pool(const float *x, const uint8_t *mask, int B, int T, int C, float *maxv, int32_t *argmax_t, float *avgv,
                           mqStream stream)
{
    MQ_REQUIRE(x && mask && maxv && argmax_t && avgv && MQ_BC_OK(B, C) && T > 0, "mq_cam_pool: bad arguments");
    hipLaunchKernelGGL(cam_pool_kernel, dim3((unsigned)cdiv(C, 64), B), dim3(64, 4), 0, S(stream), x, mask, B, T, C, maxv, argmax_t, avgv);
    MQ_LAUNCH_CHECK("mq_cam_pool");
    return 0;
}
extern "C" int mq_sam_pool(const float *x, const float *gate, int B, int T, int C, float *feat, int32_t *argmax_c, mqStream stream)
{
    MQ_REQUIRE(x && gate && feat && argmax_c && B > 0 && T > 0 && C > 0, "mq_sam_pool: bad arguments");
    hipLaunchKernelGGL(sam_pool_kernel, dim3((unsigned)cdiv((int64_t)B * T * 64, 256)), dim3(256), 0, S(stream), x, gate, B, T, C, feat,
                       argmax_c);
    MQ_LAUNCH_CHECK("mq_sam_pool");
    return 0;
}
extern "C" int mq_sam_apply(const float *x, const float *gate, const float *feat, const float *w7, int ks, int B, int T, int C,
                            float *s_save, float *out, mqStream stream)
{
    MQ_REQUIRE(x && gate && feat && w7 && s_save && out && B > 0 && T > 0 && C > 0, "mq_sam_apply: bad arguments");
    MQ_REQUIRE(ks >= 1 && ks <= 15 && (ks & 1), "mq_sam_apply: kernel size must be odd and <= 15 (got %d)", ks);
    hipLaunchKernelGGL(sam_gate_kernel, dim3((unsigned)cdiv((int64_t)B * T, 256)), dim3(256), 0, S(stream), feat, w7, ks, B, T, s_save);
    hipLaunchKernelGGL(sam_apply_kernel, dim3(ew_grid((int64_t)B * T * C, 256)), dim3(256), 0, S(stream), x, gate, s_save,
                       (int64_t)B * T, T, C, out);
    MQ_LAUNCH_CHECK("mq_sam_apply");
    return 0;
}
extern "C" size_t mq_cbam_bwd_workspace(int B, int T, int C, int ks)
{
    (void)C; (void)ks;
    if (B <= 0 || T <= 0) return 0;
    return ((size_t)B * T * 3 + 64 * 32) * sizeof(float);
}
extern "C" int mq_cbam_bwd(const float *dout, const float *x, const float *gate, const float *feat, const int32_t *argmax_c,
                           const float *s_save, const float *w7, int ks, int B, int T, int C, float *dx, float *dgate, float *dw7,
                           void *workspace, mqStream stream)
{
    MQ_REQUIRE(dout && x && gate && feat && argmax_c && s_save && w7 && dx && dgate && dw7 && workspace, "mq_cbam_bwd: null pointer");
    MQ_REQUIRE(MQ_BC_OK(B, C) && T > 0 && ks >= 1 && ks <= 15 && (ks & 1), "mq_cbam_bwd: bad shape");
    float *ws = reinterpret_cast<float *>(workspace);
    float *dlogit = ws, *dfeat = ws + (size_t)B * T, *part = ws + (size_t)B * T * 3;
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(cbam_bwd_ds_kernel, dim3((unsigned)cdiv((int64_t)B * T * 64, 256)), dim3(256), 0, st, dout, x, gate, s_save, B, T,
                       C, dlogit);
    int nb = (int)cdiv((int64_t)B * T, 256);
    if (nb > 64) nb = 64;
    hipLaunchKernelGGL(cbam_bwd_conv_kernel, dim3(nb), dim3(256), 0, st, dlogit, feat, w7, ks, B, T, dfeat, part);
    hipLaunchKernelGGL(cbam_bwd_w7_kernel, dim3(1), dim3(32), 0, st, part, nb, ks, dw7);
    hipLaunchKernelGGL(cbam_bwd_x_kernel, dim3((unsigned)cdiv(C, 64), B), dim3(64, 4), 0, st, dout, x, gate, s_save, dfeat, argmax_c, B,
                       T, C, dx, dgate);
    MQ_LAUNCH_CHECK("mq_cbam_bwd");
    return 0;
}
extern "C" int mq_cam_pool_bwd(const float *dmax, const float *davg, const int32_t *argmax_t, const uint8_t *mask, int B, int T, int C,
                               float *dx, mqStream stream)
{
    MQ_REQUIRE(dmax && davg && argmax_t && mask && dx && MQ_BC_OK(B, C) && T > 0, "mq_cam_pool_bwd: bad arguments");
    hipLaunchKernelGGL(cam_pool_bwd_kernel, dim3((unsigned)cdiv(C, 64), B), dim3(64, 4), 0, S(stream), dmax, davg, argmax_t, mask, B, T, C,
                       dx);
    MQ_LAUNCH_CHECK("mq_cam_pool_bwd");
    return 0;
}
extern "C" int mq_se_squeeze(const float *x, const uint8_t *mask, int B, int H, int W, int C, float *squeeze, mqStream stream)
{
    MQ_REQUIRE(x && mask && squeeze && MQ_BC_OK(B, C) && H > 0 && W > 0, "mq_se_squeeze: bad arguments");
    if (C % 4 == 0 && ((uintptr_t)x | (uintptr_t)squeeze) % 16 == 0)
        hipLaunchKernelGGL(se_squeeze_v4_kernel, dim3((unsigned)cdiv(C / 4, 16), B), dim3(16, 16), 0, S(stream), x, mask, B, H, W, C, squeeze);
    else
        hipLaunchKernelGGL(se_squeeze_kernel, dim3((unsigned)cdiv(C, 64), B), dim3(64, 4), 0, S(stream), x, mask, B, H, W, C, squeeze);
    MQ_LAUNCH_CHECK("mq_se_squeeze");
    return 0;
}
extern "C" int mq_se_scale_fwd(const float *x, const float *excite, int B, int HW, int C, float *y, mqStream stream)
{
    MQ_REQUIRE(x && excite && y && B > 0 && HW > 0 && C > 0, "mq_se_scale_fwd: bad arguments");
    hipLaunchKernelGGL(se_scale_fwd_kernel, dim3(ew_grid((int64_t)B * HW * C, 256)), dim3(256), 0, S(stream), x, excite, (int64_t)B,
                       (int64_t)HW, C, y);
    MQ_LAUNCH_CHECK("mq_se_scale_fwd");
    return 0;
}
extern "C" int mq_se_scale_bwd(const float *dy, const float *x, const float *excite, int B, int HW, int C, float *dx, float *dexcite,
                               mqStream stream)
{
    MQ_REQUIRE(dy && x && excite && dx && dexcite && MQ_BC_OK(B, C) && HW > 0, "mq_se_scale_bwd: bad arguments");
    hipLaunchKernelGGL(se_scale_bwd_kernel, dim3((unsigned)cdiv(C, 64), B), dim3(64, 4), 0, S(stream), dy, x, excite, B, HW, C, dx, dexcite);
    MQ_LAUNCH_CHECK("mq_se_scale_bwd");
    return 0;
}
extern "C" int mq_se_squeeze_bwd(const float *dsq, const uint8_t *mask, int B, int H, int W, int C, float *dx_accum, mqStream stream)
{
    MQ_REQUIRE(dsq && mask && dx_accum && MQ_BC_OK(B, C) && H > 0 && W > 0, "mq_se_squeeze_bwd: bad arguments");
    if (C % 4 == 0 && ((uintptr_t)dsq | (uintptr_t)dx_accum) % 16 == 0)
        hipLaunchKernelGGL(se_squeeze_bwd_v4_kernel, dim3((unsigned)cdiv(C / 4, 16), B), dim3(16, 16), 0, S(stream), dsq, mask, B, H, W, C,
                           dx_accum);
    else
        hipLaunchKernelGGL(se_squeeze_bwd_kernel, dim3((unsigned)cdiv(C, 64), B), dim3(64, 4), 0, S(stream), dsq, mask, B, H, W, C, dx_accum);
    MQ_LAUNCH_CHECK("mq_se_squeeze_bwd");
    return 0;
}
