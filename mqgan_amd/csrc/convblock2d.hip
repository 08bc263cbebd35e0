// Fused ConvBlock2D (reference preencoder.py:277-301 as used at :322 / :351).
//
// The reference inflates a (B, C0, T) map to (B, C0, C0, T) (17 GB fp32 at B=32, C0=T=512) and runs
// ~10 elementwise passes over it.  Here each pixel (b, t, c) keeps its C0-wide hidden vector in
// registers: HBM traffic is one read of x and one write of y (8 B / pixel), the work is C0
// tanh-class evaluations per pixel on the VALU / transcendental pipes.
//
//   v      = bdw + sum_{i,j} wdw[i][j] * x[b, t + j - 2, c + i - 2]          (i: channel axis, j: time)
//   v      = 0 on padded frames;  h_k = w1[k] * v + b1[k], also 0 on padded frames
//   y      = b2 + sum_k w2[k] * drop_k * aptx(h_k)                            aptx(h) = (1 + tanh h) * 0.5 * h
#include "common.h"

namespace mq {

// s / 2 = sigmoid(2 h) for s = 1 + tanh h (common.h: aptx_s = 2 * this): the kernels below take the exact factors 2 and 0.5 of
// aptx(h) = s * 0.5 * h and of its derivative out by hand - same roundings in the same order, fewer multiplications
__device__ __forceinline__ float aptx_half(float h) { return __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * h)); }
// The same for two pixels at once, on the packed f32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two results per
// issue slot; exp2 and rcp stay one per value).  __expf(x) is exp2(x * log2 e); -2 h * log2 e = h * (-2 log2 e) exactly (a power
// of two moves through the rounding), so the values are those of aptx_half bit for bit.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 aptx_half2(f32x2 h)
{
    const f32x2 t = h * splat2(-2.0f * 1.44269504088896340736f);
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + splat2(1.0f);
    return f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
__device__ __forceinline__ float and_mask(float v, uint32_t m) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & m); }

// Dropout of the hidden vector.  r3 drew the keep decisions from Philox4x32-7 in the forward AND again in the backward kernel: 28
// quarter-rate integer multiplications per 8 hidden units - as many issue slots as the arithmetic of those units (exp + rcp
// included).  Now each thread runs one cheap stream per pixel (xorshift32 + a Weyl sequence, seeded by a hash of (seed, pixel);
// 16 bits per hidden unit, threshold as in common.h: drop iff u16 < thresh16), the forward kernel leaves the decisions as ONE BIT
// per (pixel, hidden unit) in HBM ([unit / 32][pixel] words: coalesced, C0 / 8 bytes per pixel against the C0 exp + rcp it
// stands for) and the backward kernel reads them back instead of drawing again.
struct DropStream {
    uint32_t s, w;
    __device__ __forceinline__ void seed(uint64_t seed, int64_t pixel)
    {
        // both words of the seed enter both halves of the state through the two keys of common.h (r4: a seed's high word only
        // moved the Weyl term, i.e. translated every word of the stream by a constant)
        const SeedKeys key = seed_keys(seed);
        const uint32_t lo = (uint32_t)pixel, hi = (uint32_t)((uint64_t)pixel >> 32);
        s = mix32(mix32(key.k0 ^ (lo * 0x9E3779B1u) ^ mix32(hi + 0x85ebca6bu)) + key.k1);
        s = s ? s : 0x6d2b79f5u;                                   // xorshift has no zero state
        w = mix32((key.k1 * 0x9E3779B9u) ^ lo ^ 0xc2b2ae35u);
    }
    __device__ __forceinline__ uint32_t next()                     // 32 random bits: one bit plane of 32 decisions
    {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        w += 0x9E3779B9u;
        return s + w;
    }
};

constexpr int CB_TT = 4;     // time steps per block tile
constexpr int CB_TM = 16;    // time steps per thread in the MLP backward: amortises the three per-unit wave reductions (8: 2.41 ms, 16: 2.03 ms)
constexpr int CB_TC = 64;    // channels per block tile (fwd)

// Forward.  Block tile CB_FT time steps x CB_TC channels; thread = channel c, CB_FP consecutive time steps, so the three MLP
// weights of a hidden unit are read from LDS once for CB_FP pixels and the per-pixel chains are independent (packed f32 math).
constexpr int CB_FT = 32, CB_FP = 8;
template <uint32_t TC>
__global__ __launch_bounds__(256) void cb2d_fwd_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask,
                                                       const float *__restrict__ wdw, const float *__restrict__ bdw,
                                                       const float *__restrict__ w1, const float *__restrict__ b1,
                                                       const float *__restrict__ w2, const float *__restrict__ b2, int B, int T,
                                                       int C0, uint32_t thresh16, float scale, uint64_t seed,
                                                       float *__restrict__ y, float *__restrict__ v_save,
                                                       uint32_t *__restrict__ keep_bits)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *sw1 = sm, *sb1 = sm + C0, *sw2 = sm + 2 * C0;
    float *xt = sm + 3 * C0;                       // [(CB_FT+4)][(CB_TC+4)]
    __shared__ float swd[25];
    const int tid = threadIdx.x;
    const int b = blockIdx.z, t0 = blockIdx.y * CB_FT, c0 = blockIdx.x * CB_TC;
    for (int i = tid; i < C0; i += 256) {
        sw1[i] = w1[i];
        sb1[i] = b1[i];
        sw2[i] = w2[i];
    }
    if (tid < 25) swd[tid] = wdw[tid];
    for (int i = tid; i < (CB_FT + 4) * (CB_TC + 4); i += 256) {
        const int tt = i / (CB_TC + 4), cc = i - tt * (CB_TC + 4);
        const int t = t0 + tt - 2, c = c0 + cc - 2;
        xt[i] = (t >= 0 && t < T && c >= 0 && c < C0) ? x[((size_t)b * T + t) * C0 + c] : 0.f;
    }
    __syncthreads();
    const int tq = tid >> 6, tc = tid & 63;
    const int c = c0 + tc;
    if (c >= C0) return;
    const float bias2 = b2[0], bias_dw = bdw[0];
    float v[CB_FP], acc[CB_FP];
    bool inb[CB_FP], on[CB_FP];
    int64_t pix[CB_FP];
#pragma unroll
    for (int p = 0; p < CB_FP; ++p) {
        const int tt = tq * CB_FP + p, t = t0 + tt;
        inb[p] = t < T;
        pix[p] = inb[p] ? ((int64_t)b * T + t) * C0 + c : 0;
        on[p] = inb[p] && mask[(size_t)b * T + t] == 0;
        float vv = bias_dw;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) vv = fmaf(swd[i * 5 + j], xt[(tt + j) * (CB_TC + 4) + tc + i], vv);
        v[p] = on[p] ? vv : 0.f;
        if (v_save && inb[p]) v_save[pix[p]] = v[p];
        acc[p] = bias2;
    }
    f32x2 v2[CB_FP / 2], acc2[CB_FP / 2];           // pixels 2 q, 2 q + 1 side by side
#pragma unroll
    for (int q = 0; q < CB_FP / 2; ++q) {
        v2[q] = f32x2{v[2 * q], v[2 * q + 1]};
        acc2[q] = f32x2{acc[2 * q], acc[2 * q + 1]};
    }
    if (thresh16 == 0) {                            // eval mode: no dropout, no random numbers
        for (int k = 0; k < C0; ++k) {
            const f32x2 wk1 = splat2(sw1[k]), bk1 = splat2(sb1[k]), wk2 = splat2(sw2[k]);
#pragma unroll
            for (int q = 0; q < CB_FP / 2; ++q) {
                const f32x2 h = fma2(wk1, v2[q], bk1);
                acc2[q] = fma2(wk2, aptx_half2(h) * h, acc2[q]);    // = aptx_s(h) * 0.5f * h bit for bit (the factors 2 and 0.5 are exact)
            }
        }
    } else {
        const int64_t npix = (int64_t)B * T * C0;
        DropStream rng[CB_FP];
#pragma unroll
        for (int p = 0; p < CB_FP; ++p) rng[p].seed(seed, pix[p]);
        for (int k32 = 0; k32 < C0; k32 += 32) {
            // The 32 decisions of a pixel as ONE word, bit-sliced: the stream's next 16 words are the 16 bit planes (LSB first) of 32
            // uniform 16-bit numbers u_e, and "u_e >= thresh16" (= keep) falls out of a ripple comparison against the threshold's
            // bits, one bitwise operation per plane for all 32 units: going up from the LSB, a threshold bit of 1 keeps "greater or
            // equal so far" only where the plane has a 1, a threshold bit of 0 sets it wherever the plane has a 1.  Same rule (drop
            // iff u16 < thresh16, probability thresh16 / 65536), same number of random words as two 16-bit numbers per word, but
            // half an instruction per decision where extracting, comparing and collecting the bits took six (r4).
            uint32_t keep[CB_FP];
#pragma unroll
            for (int p = 0; p < CB_FP; ++p) keep[p] = ~0u;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                // TC: the threshold as a compile-time constant (the reference's hard-coded p = 0.1): one operation per plane
                const uint32_t tmask = (thresh16 >> j) & 1u ? ~0u : 0u;                 // wave-uniform
#pragma unroll
                for (int p = 0; p < CB_FP; ++p) {
                    const uint32_t r = rng[p].next();
                    if constexpr (TC != 0) keep[p] = ((TC >> j) & 1u) ? (keep[p] & r) : (keep[p] | r);
                    else keep[p] = (keep[p] & (r | ~tmask)) | (r & ~tmask);             // tmask ? keep & r : keep | r
                }
            }
#pragma unroll 4
            for (int e = 0; e < 32; ++e) {
                const int k = k32 + e;
                if (k >= C0) break;
                const f32x2 wk1 = splat2(sw1[k]), bk1 = splat2(sb1[k]), wk2s = splat2(sw2[k] * scale);
#pragma unroll
                for (int q = 0; q < CB_FP / 2; ++q) {
                    const f32x2 h = fma2(wk1, v2[q], bk1);
                    const f32x2 a = aptx_half2(h) * h;
                    // bit e of the pixel's word as a mask of all ones (v_bfe_i32), and-ed onto the activation
                    const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)keep[2 * q], e, 1);
                    const uint32_t m1 = (uint32_t)__builtin_amdgcn_sbfe((int)keep[2 * q + 1], e, 1);
                    acc2[q] = fma2(wk2s, f32x2{and_mask(a.x, m0), and_mask(a.y, m1)}, acc2[q]);
                }
            }
#pragma unroll
            for (int p = 0; p < CB_FP; ++p)
                if (inb[p]) keep_bits[(int64_t)(k32 >> 5) * npix + pix[p]] = keep[p];
        }
    }
#pragma unroll
    for (int p = 0; p < CB_FP; ++p)
        if (inb[p]) y[pix[p]] = on[p] ? (p & 1 ? acc2[p >> 1].y : acc2[p >> 1].x) : bias2;
}

// Backward pass 1: per-pixel MLP backward.  Thread = channel c, CB_TM consecutive time steps.
// Produces dv (masked) and per-block partials of dw1, db1, dw2 (3*C0) and db2.
__global__ __launch_bounds__(256) void cb2d_bwd_mlp_kernel(const float *__restrict__ v_save, const float *__restrict__ dy,
                                                           const uint8_t *__restrict__ mask, const float *__restrict__ w1,
                                                           const float *__restrict__ b1, const float *__restrict__ w2, int B, int T,
                                                           int C0, uint32_t thresh16, float scale,
                                                           const uint32_t *__restrict__ keep_bits, float *__restrict__ dv,
                                                           float *__restrict__ part, int ntiles_c, int ntiles_t)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *sw1 = sm, *sb1 = sm + C0, *sw2 = sm + 2 * C0;
    float *wacc = sm + 3 * C0;                     // [4 waves][3][C0]
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < C0; i += 256) {
        sw1[i] = w1[i];
        sb1[i] = b1[i];
        sw2[i] = w2[i];
    }
    for (int i = tid; i < 12 * C0; i += 256) wacc[i] = 0.f;
    __syncthreads();
    float *my = wacc + wave * 3 * C0;
    const int64_t npix = (int64_t)B * T * C0;
    float db2 = 0.f;
    const int64_t ntiles = (int64_t)B * ntiles_t * ntiles_c;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int ct = (int)(tile % ntiles_c);
        const int64_t r = tile / ntiles_c;
        const int tt_ = (int)(r % ntiles_t), b = (int)(r / ntiles_t);
        const int c = ct * 256 + tid;
        const int t0 = tt_ * CB_TM;
        float v[CB_TM], g[CB_TM], dva[CB_TM];
        bool on[CB_TM];
        int64_t pix[CB_TM];
#pragma unroll
        for (int p = 0; p < CB_TM; ++p) {
            const int t = t0 + p;
            const bool inb = (c < C0) && (t < T);
            pix[p] = inb ? ((int64_t)b * T + t) * C0 + c : 0;
            g[p] = inb ? dy[pix[p]] : 0.f;
            db2 += g[p];
            on[p] = inb && (mask[(size_t)b * T + t] == 0);
            v[p] = on[p] ? v_save[pix[p]] : 0.f;
            dva[p] = 0.f;
        }
        float gs[CB_TM];                                   // the output gradient times the dropout scale (1 without dropout), 0 off the mask
#pragma unroll
        for (int p = 0; p < CB_TM; ++p) gs[p] = on[p] ? g[p] * scale : 0.f;
        for (int k32 = 0; k32 < C0; k32 += 32) {
            uint32_t bits[CB_TM];
#pragma unroll
            for (int p = 0; p < CB_TM; ++p) bits[p] = (thresh16 && on[p]) ? keep_bits[(int64_t)(k32 >> 5) * npix + pix[p]] : ~0u;
#pragma unroll 4
            for (int e = 0; e < 32; ++e) {
                const int k = k32 + e;
                if (k >= C0) break;
                // two pixels per packed instruction (see aptx_half2); the sums over the thread's pixels run as an even and an odd chain
                const f32x2 wk1 = splat2(sw1[k]), bk1 = splat2(sb1[k]), wk2 = splat2(sw2[k]);
                f32x2 s_w2v = splat2(0.f), s_w1v = splat2(0.f), s_b1v = splat2(0.f);
#pragma unroll
                for (int q = 0; q < CB_TM / 2; ++q) {
                    const f32x2 vq = f32x2{v[2 * q], v[2 * q + 1]};
                    const f32x2 h = fma2(wk1, vq, bk1);
                    const f32x2 r = aptx_half2(h);                             // s / 2, s = 1 + tanh h
                    // g * scale where the unit was kept, 0 where it was dropped: bit e of the pixel's word as a mask of all ones
                    const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)bits[2 * q], e, 1);
                    const uint32_t m1 = (uint32_t)__builtin_amdgcn_sbfe((int)bits[2 * q + 1], e, 1);
                    const f32x2 gk = f32x2{and_mask(gs[2 * q], m0), and_mask(gs[2 * q + 1], m1)};
                    // a = s * 0.5 * h and dh = gk * wk2 * 0.5 * (s + h * s * (2 - s)) with the exact factors of two taken out:
                    // the same roundings in the same order, five multiplications fewer
                    const f32x2 a = r * h;
                    const f32x2 dh = (gk * wk2) * fma2(splat2(2.0f), a * (splat2(1.0f) - r), r);
                    s_w2v = fma2(gk, a, s_w2v);
                    s_w1v = fma2(dh, vq, s_w1v);
                    s_b1v += dh;
                    const f32x2 d2 = fma2(dh, wk1, f32x2{dva[2 * q], dva[2 * q + 1]});
                    dva[2 * q] = d2.x;
                    dva[2 * q + 1] = d2.y;
                }
                float s_w2 = s_w2v.x + s_w2v.y, s_w1 = s_w1v.x + s_w1v.y, s_b1 = s_b1v.x + s_b1v.y;
                s_w2 = wave_sum(s_w2);
                s_w1 = wave_sum(s_w1);
                s_b1 = wave_sum(s_b1);
                if (lane == 0) {
                    my[k] += s_w1;
                    my[C0 + k] += s_b1;
                    my[2 * C0 + k] += s_w2;
                }
            }
        }
#pragma unroll
        for (int p = 0; p < CB_TM; ++p)
            if (c < C0 && t0 + p < T) dv[pix[p]] = dva[p];
    }
    db2 = block_sum<256>(db2, red);
    __syncthreads();
    float *out = part + (size_t)blockIdx.x * (3 * C0 + 1);
    for (int i = tid; i < 3 * C0; i += 256) out[i] = (wacc[i] + wacc[3 * C0 + i]) + (wacc[6 * C0 + i] + wacc[9 * C0 + i]);
    if (tid == 0) out[3 * C0] = db2;
}

// Backward pass 2: transpose of the 5x5 stencil (dx) + its 25 weight gradients and bias gradient.
__global__ __launch_bounds__(256) void cb2d_bwd_stencil_kernel(const float *__restrict__ x, const float *__restrict__ dv,
                                                               const float *__restrict__ wdw, int B, int T, int C0,
                                                               float *__restrict__ dx, float *__restrict__ part, int ntiles_c,
                                                               int ntiles_t)
{
    __shared__ float dvt[(CB_TT + 4) * (CB_TC + 4)];
    __shared__ float xt[(CB_TT + 4) * (CB_TC + 4)];
    __shared__ float swd[25];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    if (tid < 25) swd[tid] = wdw[tid];
    float gw[26];
#pragma unroll
    for (int i = 0; i < 26; ++i) gw[i] = 0.f;
    const int tt = tid >> 6, tc = tid & 63;
    const int64_t ntiles = (int64_t)B * ntiles_t * ntiles_c;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int ct = (int)(tile % ntiles_c);
        const int64_t r = tile / ntiles_c;
        const int tt_ = (int)(r % ntiles_t), b = (int)(r / ntiles_t);
        const int t0 = tt_ * CB_TT, c0 = ct * CB_TC;
        __syncthreads();
        for (int i = tid; i < (CB_TT + 4) * (CB_TC + 4); i += 256) {
            const int a = i / (CB_TC + 4), cc = i - a * (CB_TC + 4);
            const int t = t0 + a - 2, c = c0 + cc - 2;
            const bool inb = t >= 0 && t < T && c >= 0 && c < C0;
            const size_t o = ((size_t)b * T + (inb ? t : 0)) * C0 + (inb ? c : 0);
            dvt[i] = inb ? dv[o] : 0.f;
            xt[i] = inb ? x[o] : 0.f;
        }
        __syncthreads();
        const int t = t0 + tt, c = c0 + tc;
        if (t < T && c < C0) {
            // dx[t][c] = sum_{i,j} w[i][j] * dv[t - (j-2)][c - (i-2)]
            float a = 0.f;
            const float d0 = dvt[(tt + 2) * (CB_TC + 4) + tc + 2];
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    a = fmaf(swd[i * 5 + j], dvt[(tt + 4 - j) * (CB_TC + 4) + tc + 4 - i], a);
                    gw[i * 5 + j] = fmaf(d0, xt[(tt + j) * (CB_TC + 4) + tc + i], gw[i * 5 + j]);
                }
            gw[25] += d0;
            dx[((size_t)b * T + t) * C0 + c] = a;
        }
    }
#pragma unroll
    for (int i = 0; i < 26; ++i) {
        const float s = block_sum<256>(gw[i], red);
        if (tid == 0) part[(size_t)blockIdx.x * 26 + i] = s;
    }
}

// sums `nparts` rows of `width` floats: block = 64 columns x 16 row groups (1024 threads), thread (c, g) adds rows g, g + 16, ...
// in that order, the 16 group sums are combined in a fixed order - deterministic, and 16x the loads in flight of one thread per
// column (r3: 0.24 ms per call, latency-bound on 1024 dependent loads)
__global__ __launch_bounds__(1024) void cb2d_reduce_kernel(const float *__restrict__ part, int nparts, int width, float *__restrict__ out)
{
    __shared__ float red[16][64];
    const int c = blockIdx.x * 64 + threadIdx.x, g = threadIdx.y;
    float s = 0.f;
    if (c < width)
        for (int p = g; p < nparts; p += 16) s += part[(size_t)p * width + c];
    red[g][threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && c < width) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][threadIdx.x];
        out[c] = t;
    }
}

struct CbPlan {
    int ntc_mlp, ntt, ntt_mlp, grid_mlp, ntc_st, grid_st;
    size_t off_dv, off_p1, off_p2, off_r1, off_r2, total;
};
static CbPlan cb_plan(int B, int T, int C0)
{
    CbPlan p;
    p.ntc_mlp = (int)cdiv(C0, 256);
    p.ntt = (int)cdiv(T, CB_TT);
    p.ntt_mlp = (int)cdiv(T, CB_TM);
    p.ntc_st = (int)cdiv(C0, CB_TC);
    int64_t t1 = (int64_t)B * p.ntt_mlp * p.ntc_mlp, t2 = (int64_t)B * p.ntt * p.ntc_st;
    p.grid_mlp = (int)(t1 < 1024 ? t1 : 1024);
    p.grid_st = (int)(t2 < 1024 ? t2 : 1024);
    size_t o = 0;
    p.off_dv = o; o += (size_t)B * T * C0;
    p.off_p1 = o; o += (size_t)p.grid_mlp * (3 * C0 + 1);
    p.off_p2 = o; o += (size_t)p.grid_st * 26;
    p.off_r1 = o; o += (size_t)3 * C0 + 1;
    p.off_r2 = o; o += 32;
    p.total = o * sizeof(float);
    return p;
}

__global__ void cb2d_scatter_kernel(const float *__restrict__ r1, const float *__restrict__ r2, int C0, float *dw1, float *db1,
                                    float *dw2, float *db2, float *dwdw, float *dbdw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C0) {
        dw1[i] = r1[i];
        db1[i] = r1[C0 + i];
        dw2[i] = r1[2 * C0 + i];
    }
    if (i == 0) {
        db2[0] = r1[3 * C0];
        dbdw[0] = r2[25];
    }
    if (i < 25) dwdw[i] = r2[i];
}

}  // namespace mq

using namespace mq;

static int cb_check(const mqCb2dWeights *w, int B, int T, int C0, float p, const char *who)
{
    MQ_REQUIRE(w && w->wdw && w->bdw && w->w1 && w->b1 && w->w2 && w->b2, "%s: null weight pointer", who);
    MQ_REQUIRE(B > 0 && T > 0 && C0 > 0 && B <= 65535 && cdiv(T, CB_TT) <= 65535, "%s: bad shape", who);
    MQ_REQUIRE(C0 <= 1024, "%s: C0=%d exceeds the LDS budget (1024)", who, C0);
    MQ_REQUIRE(p >= 0.f && p < 1.f, "%s: dropout p out of range", who);
    return 0;
}

extern "C" size_t mq_convblock2d_keep_bytes(int B, int T, int C0)
{
    if (B <= 0 || T <= 0 || C0 <= 0) return 0;
    return (size_t)cdiv(C0, 32) * (size_t)B * T * C0 * sizeof(uint32_t);
}

extern "C" int mq_convblock2d_fwd(const mqCb2dWeights *w, const float *x, const uint8_t *mask, int B, int T, int C0, float p_drop,
                                  uint64_t seed, float *y, float *v_save, uint32_t *keep_bits, mqStream stream)
{
    if (cb_check(w, B, T, C0, p_drop, "mq_convblock2d_fwd")) return 1;
    MQ_REQUIRE(x && mask && y, "mq_convblock2d_fwd: null pointer");
    MQ_REQUIRE(p_drop == 0.f || keep_bits, "mq_convblock2d_fwd: dropout needs the keep-bit buffer (mq_convblock2d_keep_bytes)");
    const size_t shm = (size_t)(3 * C0 + (CB_FT + 4) * (CB_TC + 4)) * sizeof(float);
    dim3 grid((unsigned)cdiv(C0, CB_TC), (unsigned)cdiv(T, CB_FT), B);
    const uint32_t th = p_drop > 0.f ? drop_thresh16(p_drop) : 0u;
    constexpr uint32_t TH_REF = 6554u;                   // drop_thresh16(0.1f): the reference's hard-coded rate gets its own instantiation
    if (th == TH_REF)
        hipLaunchKernelGGL((cb2d_fwd_kernel<TH_REF>), grid, dim3(256), shm, S(stream), x, mask, w->wdw, w->bdw, w->w1, w->b1, w->w2, w->b2,
                           B, T, C0, th, drop_scale(p_drop), seed, y, v_save, keep_bits);
    else
        hipLaunchKernelGGL((cb2d_fwd_kernel<0u>), grid, dim3(256), shm, S(stream), x, mask, w->wdw, w->bdw, w->w1, w->b1, w->w2, w->b2, B,
                           T, C0, th, p_drop > 0.f ? drop_scale(p_drop) : 1.0f, seed, y, v_save, keep_bits);
    MQ_LAUNCH_CHECK("mq_convblock2d_fwd");
    return 0;
}

extern "C" size_t mq_convblock2d_bwd_workspace(int B, int T, int C0)
{
    if (B <= 0 || T <= 0 || C0 <= 0) return 0;
    return cb_plan(B, T, C0).total;
}

extern "C" int mq_convblock2d_bwd(const mqCb2dWeights *w, const float *x, const float *v_save, const float *dy, const uint8_t *mask,
                                  int B, int T, int C0, float p_drop, const uint32_t *keep_bits, float *dx, float *dwdw, float *dbdw,
                                  float *dw1, float *db1, float *dw2, float *db2, void *workspace, mqStream stream)
{
    if (cb_check(w, B, T, C0, p_drop, "mq_convblock2d_bwd")) return 1;
    MQ_REQUIRE(p_drop == 0.f || keep_bits, "mq_convblock2d_bwd: dropout needs the keep bits the forward call wrote");
    MQ_REQUIRE(x && v_save && dy && mask && dx && dwdw && dbdw && dw1 && db1 && dw2 && db2 && workspace,
               "mq_convblock2d_bwd: null pointer");
    const CbPlan p = cb_plan(B, T, C0);
    float *ws = reinterpret_cast<float *>(workspace);
    float *dv = ws + p.off_dv, *p1 = ws + p.off_p1, *p2 = ws + p.off_p2, *r1 = ws + p.off_r1, *r2 = ws + p.off_r2;
    const uint32_t th = p_drop > 0.f ? drop_thresh16(p_drop) : 0u;
    const float sc = p_drop > 0.f ? drop_scale(p_drop) : 1.0f;
    hipStream_t st = S(stream);
    const size_t shm = (size_t)(15 * C0) * sizeof(float);
    hipLaunchKernelGGL(cb2d_bwd_mlp_kernel, dim3(p.grid_mlp), dim3(256), shm, st, v_save, dy, mask, w->w1, w->b1, w->w2, B, T, C0, th,
                       sc, keep_bits, dv, p1, p.ntc_mlp, p.ntt_mlp);
    hipLaunchKernelGGL(cb2d_bwd_stencil_kernel, dim3(p.grid_st), dim3(256), 0, st, x, dv, w->wdw, B, T, C0, dx, p2, p.ntc_st, p.ntt);
    hipLaunchKernelGGL(cb2d_reduce_kernel, dim3((unsigned)cdiv(3 * C0 + 1, 64)), dim3(64, 16), 0, st, p1, p.grid_mlp, 3 * C0 + 1, r1);
    hipLaunchKernelGGL(cb2d_reduce_kernel, dim3(1), dim3(64, 16), 0, st, p2, p.grid_st, 26, r2);
    hipLaunchKernelGGL(cb2d_scatter_kernel, dim3((unsigned)cdiv(C0 > 32 ? C0 : 32, 256)), dim3(256), 0, st, r1, r2, C0, dw1, db1, dw2,
                       db2, dwdw, dbdw);
    MQ_LAUNCH_CHECK("mq_convblock2d_bwd");
    return 0;
}
