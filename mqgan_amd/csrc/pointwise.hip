// Elementwise / data-movement kernels: fused mask+activation+dropout(+residual), frame masks,
// UNet pool / upsample+concat, band split, small helpers.  All HBM-bound: float4 per lane where the
// channel count allows, grid-stride over at most 2048 blocks.
#include "common.h"

namespace mq {

struct PwParams {
    int64_t rows;
    int C;
    int act, mask_pre, mask_post;
    int64_t d1, d2, d3;
    uint32_t thresh16;
    float scale;
    uint64_t seed;
};

__device__ __forceinline__ float act_fwd(int act, float t, float beta, float gamma)
{
    if (act == 1) return aptx_s(beta * t) * gamma * t;
    if (act == 2) return t > 0.f ? t : 0.2f * t;
    return t;
}
// returns f'(t); accumulates d/dbeta, d/dgamma (per unit upstream gradient) into gb, gg
__device__ __forceinline__ float act_bwd(int act, float t, float beta, float gamma, float g, float &gb, float &gg)
{
    if (act == 1) {
        const float s = aptx_s(beta * t);
        const float ss = s * (2.0f - s);          // = 1 - tanh^2
        gb += g * gamma * t * t * ss;
        gg += g * s * t;
        return gamma * (s + t * beta * ss);
    }
    if (act == 2) return t > 0.f ? 1.0f : 0.2f;
    return 1.0f;
}

// 8 consecutive elements <-> registers; T = float (two 16-byte accesses) or bf16 stored as unsigned short (one)
typedef unsigned short bf16_t;
__device__ __forceinline__ void ld8(const float *p, int64_t e0, float v[8])
{
    const float4 a = *reinterpret_cast<const float4 *>(p + e0), b = *reinterpret_cast<const float4 *>(p + e0 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const bf16_t *p, int64_t e0, float v[8])
{
    const uint4 a = *reinterpret_cast<const uint4 *>(p + e0);
    const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __builtin_bit_cast(float, w[i] << 16);
        v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ void st8(float *p, int64_t e0, const float v[8])
{
    *reinterpret_cast<float4 *>(p + e0) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4 *>(p + e0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void st8(bf16_t *p, int64_t e0, const float v[8])
{
    typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
    typedef float f32x8v __attribute__((ext_vector_type(8)));
    const f32x8v f = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    *reinterpret_cast<bf16x8v *>(p + e0) = __builtin_convertvector(f, bf16x8v);       // RNE, NaN-safe
}
__device__ __forceinline__ float ld1(const float *p, int64_t e) { return p[e]; }
__device__ __forceinline__ float ld1(const bf16_t *p, int64_t e) { return __builtin_bit_cast(float, (unsigned)p[e] << 16); }
__device__ __forceinline__ void st1(float *p, int64_t e, float v) { p[e] = v; }
__device__ __forceinline__ void st1(bf16_t *p, int64_t e, float v) { p[e] = __builtin_bit_cast(unsigned short, (__bf16)v); }

// one thread = one group of 8 consecutive flat elements (= one Philox call).  T: storage type of the activations.
template <bool VEC8, typename T>
__global__ void pointwise_fwd_kernel(PwParams p, const T *__restrict__ h, const T *__restrict__ rpre,
                                     const T *__restrict__ rpost, const uint8_t *__restrict__ mask,
                                     const float *__restrict__ beta_p, const float *__restrict__ gamma_p, T *__restrict__ out)
{
    const int64_t total = p.rows * p.C;
    const int64_t ngroups = (total + 7) >> 3;
    const float beta = beta_p ? beta_p[0] : 1.0f, gamma = gamma_p ? gamma_p[0] : 0.5f;
    for (int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gidx < ngroups; gidx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e0 = gidx << 3;
        float t[8], a[8];
        bool pad[8];
        if (VEC8) {
            ld8(h, e0, t);
            if (rpre) {
                float r[8];
                ld8(rpre, e0, r);
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] += r[i];
            }
            bool pd = false;
            if (mask) {
                const int64_t m = e0 / p.C;
                pd = mask[mask_index(m, p.d1, p.d2, p.d3)] != 0;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) pad[i] = pd;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t e = e0 + i;
                t[i] = 0.f;
                pad[i] = false;
                if (e < total) {
                    t[i] = ld1(h, e) + (rpre ? ld1(rpre, e) : 0.f);
                    if (mask) {
                        const int64_t m = e / p.C;
                        pad[i] = mask[mask_index(m, p.d1, p.d2, p.d3)] != 0;
                    }
                }
            }
        }
        const bool use_drop = p.thresh16 != 0;
        Drop8 rng;
        if (use_drop) rng = Drop8(p.seed, (uint64_t)gidx);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float tv = (p.mask_pre && pad[i]) ? 0.f : t[i];
            float av = act_fwd(p.act, tv, beta, gamma);
            if (use_drop) av = rng.keep(i, p.thresh16) ? av * p.scale : 0.f;
            a[i] = av;
        }
        if (VEC8) {
            if (rpost) {
                float r[8];
                ld8(rpost, e0, r);
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] += r[i];
            }
            if (p.mask_post && pad[0]) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = 0.f;
            }
            st8(out, e0, a);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t e = e0 + i;
                if (e < total) {
                    float av = a[i] + (rpost ? ld1(rpost, e) : 0.f);
                    if (p.mask_post && pad[i]) av = 0.f;
                    st1(out, e, av);
                }
            }
        }
    }
}

template <bool VEC8, typename T>
__global__ void pointwise_bwd_kernel(PwParams p, const T *__restrict__ dout, const T *__restrict__ h,
                                     const T *__restrict__ rpre, const uint8_t *__restrict__ mask,
                                     const float *__restrict__ beta_p, const float *__restrict__ gamma_p, T *__restrict__ dt,
                                     T *__restrict__ dpost, float *__restrict__ part)
{
    __shared__ float red[4];
    const int64_t total = p.rows * p.C;
    const int64_t ngroups = (total + 7) >> 3;
    const float beta = beta_p ? beta_p[0] : 1.0f, gamma = gamma_p ? gamma_p[0] : 0.5f;
    float gb = 0.f, gg = 0.f;
    for (int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gidx < ngroups; gidx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e0 = gidx << 3;
        float t[8], g[8];
        bool pad[8], live[8];
        if (VEC8) {
            ld8(h, e0, t);
            if (rpre) {
                float r[8];
                ld8(rpre, e0, r);
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] += r[i];
            }
            ld8(dout, e0, g);
            bool pd = false;
            if (mask) {
                const int64_t m = e0 / p.C;
                pd = mask[mask_index(m, p.d1, p.d2, p.d3)] != 0;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { pad[i] = pd; live[i] = true; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t e = e0 + i;
                t[i] = 0.f; g[i] = 0.f; pad[i] = false;
                live[i] = e < total;
                if (live[i]) {
                    t[i] = ld1(h, e) + (rpre ? ld1(rpre, e) : 0.f);
                    g[i] = ld1(dout, e);
                    if (mask) {
                        const int64_t m = e / p.C;
                        pad[i] = mask[mask_index(m, p.d1, p.d2, p.d3)] != 0;
                    }
                }
            }
        }
        const bool use_drop = p.thresh16 != 0;
        Drop8 rng;
        if (use_drop) rng = Drop8(p.seed, (uint64_t)gidx);
        float po[8], dv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float gv = (p.mask_post && pad[i]) ? 0.f : g[i];
            po[i] = gv;
            if (use_drop) gv = rng.keep(i, p.thresh16) ? gv * p.scale : 0.f;
            const bool mp = p.mask_pre && pad[i];
            const float tv = mp ? 0.f : t[i];
            float b_ = 0.f, g_ = 0.f;
            const float d = act_bwd(p.act, tv, beta, gamma, gv, b_, g_);
            if (live[i]) { gb += b_; gg += g_; }
            dv[i] = mp ? 0.f : gv * d;
        }
        if (VEC8) {
            st8(dt, e0, dv);
            if (dpost) st8(dpost, e0, po);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (live[i]) {
                    st1(dt, e0 + i, dv[i]);
                    if (dpost) st1(dpost, e0 + i, po[i]);
                }
        }
    }
    if (part) {
        gb = block_sum<256>(gb, red);
        gg = block_sum<256>(gg, red);
        if (threadIdx.x == 0) {
            part[blockIdx.x] = gb;
            part[gridDim.x + blockIdx.x] = gg;
        }
    }
}

__global__ void pw_final_kernel(const float *__restrict__ part, int nblocks, float *__restrict__ dbeta, float *__restrict__ dgamma)
{
    __shared__ float red[4];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        a += part[i];
        b += part[nblocks + i];
    }
    a = block_sum<256>(a, red);
    b = block_sum<256>(b, red);
    if (threadIdx.x == 0) {
        if (dbeta) dbeta[0] = a;
        if (dgamma) dgamma[0] = b;
    }
}

// ------------------------------------------------------------------------------------------ masks
__global__ void sequence_mask_kernel(const int32_t *__restrict__ len, int B, int T, uint8_t *__restrict__ mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const int b = i / T, t = i - b * T;
    mask[i] = t >= len[b] ? 1 : 0;
}
__global__ void mask_pool_kernel(const uint8_t *__restrict__ in, int B, int Tin, int Tout, int stride, uint8_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Tout) return;
    const int b = i / Tout, t = i - b * Tout;
    uint8_t m = 0;
    for (int j = 0; j < stride; ++j) {
        const int s = t * stride + j;
        if (s < Tin) m |= in[b * Tin + s];
    }
    out[i] = m;
}
__global__ void mask_upsample2_kernel(const uint8_t *__restrict__ in, int B, int Tin, uint8_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Tin * 2) return;
    const int b = i / (2 * Tin), t = i - b * 2 * Tin;
    out[i] = in[b * Tin + (t >> 1)];
}

// ------------------------------------------------------------------------------------------ UNet data movement
// x (N, T, F, C) viewed as (N, T, FC)
__global__ void avgpool_t2_fwd_kernel(const float *__restrict__ x, int64_t N, int T, int64_t FC, float *__restrict__ y)
{
    const int To = T / 2;
    const int64_t total = N * To * FC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t fc = i % FC, r = i / FC;
        const int64_t t = r % To, n = r / To;
        const float *p = x + ((n * T + 2 * t) * FC + fc);
        y[i] = (p[0] + p[FC]) * 0.5f;
    }
}
__global__ void avgpool_t2_bwd_kernel(const float *__restrict__ dy, int64_t N, int T, int64_t FC, float *__restrict__ dx)
{
    const int To = T / 2;
    const int64_t total = N * T * FC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t fc = i % FC, r = i / FC;
        const int64_t t = r % T, n = r / T;
        dx[i] = 0.5f * dy[(n * To + (t >> 1)) * FC + fc];
    }
}
// out (N, 2T, F, C1+C2)
__global__ void upcat_fwd_kernel(const float *__restrict__ x, const float *__restrict__ skip, int64_t N, int T, int F, int C1,
                                 int C2, float *__restrict__ out)
{
    const int Co = C1 + C2;
    const int64_t total = N * 2 * T * F * Co;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Co);
        const int64_t r = i / Co;           // (n, t2, f)
        const int64_t f = r % F, r2 = r / F;
        const int64_t t2 = r2 % (2 * T), n = r2 / (2 * T);
        out[i] = c < C1 ? x[((n * T + (t2 >> 1)) * F + f) * C1 + c] : skip[r * C2 + (c - C1)];
    }
}
__global__ void upcat_bwd_x_kernel(const float *__restrict__ dout, int64_t N, int T, int F, int C1, int C2, float *__restrict__ dx)
{
    const int Co = C1 + C2;
    const int64_t total = N * T * F * C1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C1);
        const int64_t r = i / C1;
        const int64_t f = r % F, r2 = r / F;
        const int64_t t = r2 % T, n = r2 / T;
        const int64_t o = ((n * 2 * T + 2 * t) * F + f) * Co + c;
        dx[i] = dout[o] + dout[o + (int64_t)F * Co];
    }
}
__global__ void upcat_bwd_skip_kernel(const float *__restrict__ dout, int64_t rows, int C1, int C2, float *__restrict__ dskip)
{
    const int Co = C1 + C2;
    const int64_t total = rows * C2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C2);
        const int64_t r = i / C2;
        dskip[i] = dout[r * Co + C1 + c];
    }
}
__global__ void concat2_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t rows, int C1, int C2,
                               float *__restrict__ y)
{
    const int Co = C1 + C2;
    const int64_t total = rows * Co;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Co);
        const int64_t r = i / Co;
        y[i] = c < C1 ? a[r * C1 + c] : b[r * C2 + (c - C1)];
    }
}
__global__ void axpby_kernel(const float *__restrict__ a, const float *__restrict__ b, float alpha, float beta, int64_t n,
                             float *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}

// (B, T, F) <-> (n_bins*B, F/n_bins, T) through a 32x32 LDS tile.  FWD: x -> out ; !FWD: dout -> dx.
template <bool FWD>
__global__ void band_split_kernel(const float *__restrict__ src, int B, int T, int F, int n_bins, float *__restrict__ dst)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
    const int bs = F / n_bins;
    const int tx = threadIdx.x, ty = threadIdx.y;   // (32, 8)
    if (FWD) {
        for (int j = ty; j < 32; j += 8) {
            const int t = t0 + j, f = f0 + tx;
            tile[j][tx] = (t < T && f < F) ? src[((size_t)b * T + t) * F + f] : 0.f;
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int f = f0 + j, t = t0 + tx;
            if (f < F && t < T) {
                const size_t row = ((size_t)(f / bs) * B + b) * bs + (f % bs);
                dst[row * T + t] = tile[tx][j];
            }
        }
    } else {
        for (int j = ty; j < 32; j += 8) {
            const int f = f0 + j, t = t0 + tx;
            float v = 0.f;
            if (f < F && t < T) {
                const size_t row = ((size_t)(f / bs) * B + b) * bs + (f % bs);
                v = src[row * T + t];
            }
            tile[j][tx] = v;
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int t = t0 + j, f = f0 + tx;
            if (t < T && f < F) dst[((size_t)b * T + t) * F + f] = tile[tx][j];
        }
    }
}

__global__ void sigmoid_add_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t n, float *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float z = a[i] + (b ? b[i] : 0.f);
        out[i] = 1.0f / (1.0f + __expf(-z));
    }
}
__global__ void sigmoid_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ out, int64_t n, float *__restrict__ da)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        da[i] = dout[i] * out[i] * (1.0f - out[i]);
}
__global__ void relu_fwd_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = fmaxf(x[i], 0.f);
}
__global__ void relu_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, int64_t n, float *__restrict__ dx)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

static int make_pw(const mqPointwiseDesc *d, PwParams &p, const char *who)
{
    MQ_REQUIRE(d && d->rows > 0 && d->C > 0, "%s: bad shape", who);
    MQ_REQUIRE(d->act >= 0 && d->act <= 2, "%s: unknown activation %d", who, d->act);
    MQ_REQUIRE(d->p_drop >= 0.f && d->p_drop < 1.f, "%s: dropout p out of range", who);
    p.rows = d->rows; p.C = d->C; p.act = d->act; p.mask_pre = d->mask_pre; p.mask_post = d->mask_post;
    p.d1 = d->d1 > 0 ? d->d1 : 1; p.d2 = d->d2; p.d3 = d->d3 > 0 ? d->d3 : 1;
    p.thresh16 = d->p_drop > 0.f ? drop_thresh16(d->p_drop) : 0u;
    p.scale = d->p_drop > 0.f ? drop_scale(d->p_drop) : 1.0f;
    p.seed = d->seed;
    return 0;
}
static bool al16(const void *q) { return q == nullptr || ((uintptr_t)q % 16) == 0; }


// ------------------------------------------------------------------------------------------ the vector path, specialised (r5)
// The two kernels above decide activation and dropout per element at run time and find the frame-mask byte of a group with a
// 64-bit division plus the two of mask_index: ~100 issue slots per element on paper, against 4 bytes per element in bf16 storage -
// the bf16 launches ran at 4.2-4.8 TB/s where the same code on f32 tensors (8 bytes per element) reached 5.7.  Here the activation
// and the dropout are template parameters and the three divisions are multiply-high divisions by invariants the host prepares
// (Granlund & Montgomery 1994, exact for every 32-bit dividend); shapes beyond 32 bits keep the generic kernels.  Same arithmetic
// in the same order: results are bit-identical to the generic kernels (tests/test_gpu_pointwise.py).
struct FastDiv {
    uint32_t d, mul, sh1, sh2;
};
static FastDiv make_fastdiv(uint32_t d)
{
    FastDiv f;
    f.d = d;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;                                   // ceil(log2 d)
    f.mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &f)
{
    const uint32_t t = __umulhi(f.mul, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}
struct PwFast {
    FastDiv c8, d1, d3;        // groups per row, and mask_index's two divisors
    uint32_t d2;
};
// true when every index of the launch fits the 32-bit divisions
static bool make_pw_fast(const PwParams &p, PwFast &f)
{
    const int64_t groups = p.rows * (int64_t)p.C / 8;
    if (p.C % 8 != 0 || groups >= 0xFFFFFFFFll || p.rows >= 0xFFFFFFFFll || p.d1 >= 0xFFFFFFFFll || p.d3 >= 0xFFFFFFFFll || p.d2 < 0 ||
        p.d2 >= 0xFFFFFFFFll)
        return false;
    f.c8 = make_fastdiv((uint32_t)(p.C / 8));
    f.d1 = make_fastdiv((uint32_t)p.d1);
    f.d3 = make_fastdiv((uint32_t)p.d3);
    f.d2 = (uint32_t)p.d2;
    return true;
}
__device__ __forceinline__ bool pw_padded(const uint8_t *__restrict__ mask, uint32_t group, const PwFast &f)
{
    const uint32_t m = fdiv(group, f.c8);
    const uint32_t q1 = fdiv(m, f.d1), r3 = m - fdiv(m, f.d3) * f.d3.d;
    return mask[(uint64_t)q1 * f.d2 + r3] != 0;
}

template <typename T, int ACT, bool DROP>
__global__ __launch_bounds__(256) void pointwise_fwd_v8_kernel(PwParams p, PwFast f, const T *__restrict__ h, const T *__restrict__ rpre,
                                                               const T *__restrict__ rpost, const uint8_t *__restrict__ mask,
                                                               const float *__restrict__ beta_p, const float *__restrict__ gamma_p,
                                                               T *__restrict__ out)
{
    const int64_t ngroups = p.rows * p.C >> 3;
    const float beta = beta_p ? beta_p[0] : 1.0f, gamma = gamma_p ? gamma_p[0] : 0.5f;
    for (int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gidx < ngroups; gidx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e0 = gidx << 3;
        float t[8], a[8];
        ld8(h, e0, t);
        if (rpre) {
            float r[8];
            ld8(rpre, e0, r);
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] += r[i];
        }
        const bool pd = mask ? pw_padded(mask, (uint32_t)gidx, f) : false;
        Drop8 rng;
        if constexpr (DROP) rng = Drop8(p.seed, (uint64_t)gidx);
        const bool zero_in = p.mask_pre && pd;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float tv = zero_in ? 0.f : t[i];
            float av = act_fwd(ACT, tv, beta, gamma);
            if constexpr (DROP) av = rng.keep(i, p.thresh16) ? av * p.scale : 0.f;
            a[i] = av;
        }
        if (rpost) {
            float r[8];
            ld8(rpost, e0, r);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] += r[i];
        }
        if (p.mask_post && pd) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = 0.f;
        }
        st8(out, e0, a);
    }
}

template <typename T, int ACT, bool DROP>
__global__ __launch_bounds__(256) void pointwise_bwd_v8_kernel(PwParams p, PwFast f, const T *__restrict__ dout, const T *__restrict__ h,
                                                               const T *__restrict__ rpre, const uint8_t *__restrict__ mask,
                                                               const float *__restrict__ beta_p, const float *__restrict__ gamma_p,
                                                               T *__restrict__ dt, T *__restrict__ dpost, float *__restrict__ part)
{
    __shared__ float red[4];
    const int64_t ngroups = p.rows * p.C >> 3;
    const float beta = beta_p ? beta_p[0] : 1.0f, gamma = gamma_p ? gamma_p[0] : 0.5f;
    float gb = 0.f, gg = 0.f;
    for (int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gidx < ngroups; gidx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e0 = gidx << 3;
        float t[8], g[8];
        ld8(h, e0, t);
        if (rpre) {
            float r[8];
            ld8(rpre, e0, r);
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] += r[i];
        }
        ld8(dout, e0, g);
        const bool pd = mask ? pw_padded(mask, (uint32_t)gidx, f) : false;
        Drop8 rng;
        if constexpr (DROP) rng = Drop8(p.seed, (uint64_t)gidx);
        const bool zero_out = p.mask_post && pd, mp = p.mask_pre && pd;
        float po[8], dv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float gv = zero_out ? 0.f : g[i];
            po[i] = gv;
            if constexpr (DROP) gv = rng.keep(i, p.thresh16) ? gv * p.scale : 0.f;
            const float tv = mp ? 0.f : t[i];
            float b_ = 0.f, g_ = 0.f;
            const float d = act_bwd(ACT, tv, beta, gamma, gv, b_, g_);
            gb += b_; gg += g_;
            dv[i] = mp ? 0.f : gv * d;
        }
        st8(dt, e0, dv);
        if (dpost) st8(dpost, e0, po);
    }
    if (part) {
        gb = block_sum<256>(gb, red);
        gg = block_sum<256>(gg, red);
        if (threadIdx.x == 0) {
            part[blockIdx.x] = gb;
            part[gridDim.x + blockIdx.x] = gg;
        }
    }
}

// f32 -> bf16 (round to nearest even, NaN stays NaN): the operand copies the bf16-storage conv kernels read
__global__ void cast_bf16_kernel(const float *__restrict__ in, unsigned short *__restrict__ out, int64_t n8, int64_t n)
{
    typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
    typedef float f32x8v __attribute__((ext_vector_type(8)));
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4 *>(in)[2 * i], b = reinterpret_cast<const float4 *>(in)[2 * i + 1];
        const f32x8v f = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        reinterpret_cast<bf16x8v *>(out)[i] = __builtin_convertvector(f, bf16x8v);
    }
    if (blockIdx.x == 0)
        for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) {
            const __bf16 h = (__bf16)in[i];
            out[i] = __builtin_bit_cast(unsigned short, h);
        }
}

// bf16 -> f32
__global__ void cast_f32_kernel(const bf16_t *__restrict__ in, float *__restrict__ out, int64_t n8, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float v[8];
        ld8(in, i * 8, v);
        st8(out, i * 8, v);
    }
    if (blockIdx.x == 0)
        for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) out[i] = ld1(in, i);
}

// ---- UNet data movement, 8 channels per thread (C % 8 == 0); T = bf16 storage or f32
// x (N, T, F*C) -> y (N, T/2, F*C)
template <typename T>
__global__ void avgpool_t2_fwd_h_kernel(const T *__restrict__ x, int64_t N, int Tt, int64_t FC8, const uint8_t *__restrict__ mask,
                                        T *__restrict__ y)
{
    // mask (N, Tt / 2), nonzero = padded frame of the OUTPUT: written as zeros (the consumer block's masked_fill, fused)
    const int To = Tt / 2;
    const int64_t total = N * To * FC8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t fc = i % FC8, r = i / FC8;
        const int64_t t = r % To, n = r / To;
        float a[8], b[8];
        if (mask && mask[r]) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = 0.f;
        } else {
            ld8(x, ((n * Tt + 2 * t) * FC8 + fc) * 8, a);
            ld8(x, ((n * Tt + 2 * t + 1) * FC8 + fc) * 8, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = (a[k] + b[k]) * 0.5f;
        }
        st8(y, i * 8, a);
    }
}
template <typename T>
__global__ void avgpool_t2_bwd_h_kernel(const T *__restrict__ dy, int64_t N, int Tt, int64_t FC8, const uint8_t *__restrict__ mask,
                                        T *__restrict__ dx)
{
    const int To = Tt / 2;
    const int64_t total = N * To * FC8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t fc = i % FC8, r = i / FC8;
        const int64_t t = r % To, n = r / To;
        float a[8];
        if (mask && mask[r]) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = 0.f;
        } else {
            ld8(dy, i * 8, a);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] *= 0.5f;
        st8(dx, ((n * Tt + 2 * t) * FC8 + fc) * 8, a);
        st8(dx, ((n * Tt + 2 * t + 1) * FC8 + fc) * 8, a);
    }
}
// out (N, 2T, F, C1 + C2) = cat(upsample2(x (N,T,F,C1)), skip (N,2T,F,C2)); one thread = 8 channels of one output pixel
template <typename T>
__global__ void upcat_fwd_h_kernel(const T *__restrict__ x, const T *__restrict__ skip, int64_t N, int Tt, int F, int C1,
                                   int C2, const uint8_t *__restrict__ mask, T *__restrict__ out)
{
    // mask (N, 2 Tt), nonzero = padded frame of the OUTPUT: written as zeros
    const int Co8 = (C1 + C2) / 8, C18 = C1 / 8;
    const int64_t total = N * 2 * Tt * F * Co8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % Co8);
        const int64_t r = i / Co8;           // (n, t2, f)
        const int64_t f = r % F, r2 = r / F;
        const int64_t t2 = r2 % (2 * Tt), n = r2 / (2 * Tt);
        float v[8];
        if (mask && mask[r2]) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
        } else if (c8 < C18) {
            ld8(x, ((n * Tt + (t2 >> 1)) * F + f) * C1 + c8 * 8, v);
        } else {
            ld8(skip, r * C2 + (c8 - C18) * 8, v);
        }
        st8(out, i * 8, v);
    }
}
// dx (N,T,F,C1) = sum of the two upsampled rows; dskip (N,2T,F,C2) = copy
template <typename T>
__global__ void upcat_bwd_h_kernel(const T *__restrict__ dout, int64_t N, int Tt, int F, int C1, int C2,
                                   const uint8_t *__restrict__ mask, T *__restrict__ dx, T *__restrict__ dskip)
{
    const int Co = C1 + C2, Co8 = Co / 8, C18 = C1 / 8;
    const int64_t total = N * Tt * F * Co8;          // one thread: 8 channels of an INPUT-resolution pixel (both output rows)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % Co8);
        const int64_t r = i / Co8;           // (n, t, f)
        const int64_t f = r % F, r2 = r / F;
        const int64_t t = r2 % Tt, n = r2 / Tt;
        const int64_t o0 = ((n * 2 * Tt + 2 * t) * F + f), o1 = o0 + F;
        const bool p0 = mask && mask[n * 2 * Tt + 2 * t], p1 = mask && mask[n * 2 * Tt + 2 * t + 1];
        float a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = b[k] = 0.f;
        if (c8 < C18) {
            if (!p0) ld8(dout, o0 * Co + c8 * 8, a);
            if (!p1) ld8(dout, o1 * Co + c8 * 8, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += b[k];
            st8(dx, r * C1 + c8 * 8, a);
        } else {
            const int cs = (c8 - C18) * 8;
            if (!p0) ld8(dout, o0 * Co + C1 + cs, a);
            if (!p1) ld8(dout, o1 * Co + C1 + cs, b);
            st8(dskip, o0 * C2 + cs, a);
            st8(dskip, o1 * C2 + cs, b);
        }
    }
}

}  // namespace mq

using namespace mq;

// the specialised vector kernels (act x dropout) when the launch fits 32-bit indices
#define MQ_PW_V8(KERNEL, T, grid, ...)                                                                                        \
    do {                                                                                                                      \
        const bool drop_ = p.thresh16 != 0;                                                                                   \
        if (p.act == 1) { if (drop_) hipLaunchKernelGGL((KERNEL<T, 1, true>), dim3(grid), dim3(256), 0, S(stream), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<T, 1, false>), dim3(grid), dim3(256), 0, S(stream), __VA_ARGS__); } \
        else if (p.act == 2) { if (drop_) hipLaunchKernelGGL((KERNEL<T, 2, true>), dim3(grid), dim3(256), 0, S(stream), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<T, 2, false>), dim3(grid), dim3(256), 0, S(stream), __VA_ARGS__); } \
        else { if (drop_) hipLaunchKernelGGL((KERNEL<T, 0, true>), dim3(grid), dim3(256), 0, S(stream), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<T, 0, false>), dim3(grid), dim3(256), 0, S(stream), __VA_ARGS__); } \
    } while (0)
static bool pw_generic() { return getenv("MQGAN_PW_GENERIC") != nullptr; }   // read per call: A/B runs and the identity test

extern "C" int mq_pointwise_fwd(const mqPointwiseDesc *d, const float *h, const float *r_pre, const float *r_post,
                                const uint8_t *mask, const float *beta, const float *gamma, float *out, mqStream stream)
{
    PwParams p;
    if (make_pw(d, p, "mq_pointwise_fwd")) return 1;
    MQ_REQUIRE(h && out, "mq_pointwise_fwd: null pointer");
    MQ_REQUIRE(mask || !(p.mask_pre || p.mask_post), "mq_pointwise_fwd: mask flags set but mask is null");
    const int64_t groups = (p.rows * p.C + 7) / 8;
    const bool vec = (p.C % 8 == 0) && al16(h) && al16(r_pre) && al16(r_post) && al16(out);
    const int grid = ew_grid(groups, 256);
    PwFast f;
    if (vec && !pw_generic() && make_pw_fast(p, f))
        MQ_PW_V8(pointwise_fwd_v8_kernel, float, grid, p, f, h, r_pre, r_post, mask, beta, gamma, out);
    else if (vec)
        hipLaunchKernelGGL((pointwise_fwd_kernel<true, float>), dim3(grid), dim3(256), 0, S(stream), p, h, r_pre, r_post, mask, beta,
                           gamma, out);
    else
        hipLaunchKernelGGL((pointwise_fwd_kernel<false, float>), dim3(grid), dim3(256), 0, S(stream), p, h, r_pre, r_post, mask, beta,
                           gamma, out);
    MQ_LAUNCH_CHECK("mq_pointwise_fwd");
    return 0;
}

extern "C" int mq_pointwise_bwd(const mqPointwiseDesc *d, const float *dout, const float *h, const float *r_pre,
                                const uint8_t *mask, const float *beta, const float *gamma, float *dt, float *dpost, float *dbeta,
                                float *dgamma, float *workspace, mqStream stream)
{
    PwParams p;
    if (make_pw(d, p, "mq_pointwise_bwd")) return 1;
    MQ_REQUIRE(dout && h && dt, "mq_pointwise_bwd: null pointer");
    MQ_REQUIRE(mask || !(p.mask_pre || p.mask_post), "mq_pointwise_bwd: mask flags set but mask is null");
    const bool want_bg = dbeta || dgamma;
    MQ_REQUIRE(!want_bg || workspace, "mq_pointwise_bwd: dbeta/dgamma need a workspace");
    const int64_t groups = (p.rows * p.C + 7) / 8;
    const bool vec = (p.C % 8 == 0) && al16(h) && al16(r_pre) && al16(dout) && al16(dt) && al16(dpost);
    int grid = ew_grid(groups, 256);
    if (grid > 1024) grid = 1024;
    float *part = want_bg ? workspace : nullptr;
    PwFast f;
    if (vec && !pw_generic() && make_pw_fast(p, f))
        MQ_PW_V8(pointwise_bwd_v8_kernel, float, grid, p, f, dout, h, r_pre, mask, beta, gamma, dt, dpost, part);
    else if (vec)
        hipLaunchKernelGGL((pointwise_bwd_kernel<true, float>), dim3(grid), dim3(256), 0, S(stream), p, dout, h, r_pre, mask, beta,
                           gamma, dt, dpost, part);
    else
        hipLaunchKernelGGL((pointwise_bwd_kernel<false, float>), dim3(grid), dim3(256), 0, S(stream), p, dout, h, r_pre, mask, beta,
                           gamma, dt, dpost, part);
    if (want_bg) hipLaunchKernelGGL(pw_final_kernel, dim3(1), dim3(256), 0, S(stream), part, grid, dbeta, dgamma);
    MQ_LAUNCH_CHECK("mq_pointwise_bwd");
    return 0;
}

extern "C" int mq_sequence_mask(const int32_t *lengths, int B, int T, uint8_t *mask, mqStream stream)
{
    MQ_REQUIRE(lengths && mask && B > 0 && T > 0, "mq_sequence_mask: bad arguments");
    hipLaunchKernelGGL(sequence_mask_kernel, dim3((unsigned)cdiv((int64_t)B * T, 256)), dim3(256), 0, S(stream), lengths, B, T, mask);
    MQ_LAUNCH_CHECK("mq_sequence_mask");
    return 0;
}
extern "C" int mq_mask_pool(const uint8_t *in, int B, int Tin, int Tout, int stride, uint8_t *out, mqStream stream)
{
    MQ_REQUIRE(in && out && B > 0 && Tin > 0 && Tout > 0 && stride > 0, "mq_mask_pool: bad arguments");
    MQ_REQUIRE((int64_t)(Tout - 1) * stride < Tin, "mq_mask_pool: Tout=%d too large for Tin=%d stride=%d", Tout, Tin, stride);
    hipLaunchKernelGGL(mask_pool_kernel, dim3((unsigned)cdiv((int64_t)B * Tout, 256)), dim3(256), 0, S(stream), in, B, Tin, Tout,
                       stride, out);
    MQ_LAUNCH_CHECK("mq_mask_pool");
    return 0;
}
extern "C" int mq_mask_upsample2(const uint8_t *in, int B, int Tin, uint8_t *out, mqStream stream)
{
    MQ_REQUIRE(in && out && B > 0 && Tin > 0, "mq_mask_upsample2: bad arguments");
    hipLaunchKernelGGL(mask_upsample2_kernel, dim3((unsigned)cdiv((int64_t)B * Tin * 2, 256)), dim3(256), 0, S(stream), in, B, Tin,
                       out);
    MQ_LAUNCH_CHECK("mq_mask_upsample2");
    return 0;
}

extern "C" int mq_avgpool_t2_fwd(const float *x, int N, int T, int F, int C, const uint8_t *mask, float *y, mqStream stream)
{
    MQ_REQUIRE(x && y && N > 0 && T > 0 && T % 2 == 0 && F > 0 && C > 0, "mq_avgpool_t2_fwd: bad arguments (T must be even)");
    MQ_REQUIRE(!mask || (C % 8 == 0 && al16(x) && al16(y)), "mq_avgpool_t2_fwd: the fused output mask needs C %% 8 == 0 (C=%d)", C);
    const int64_t FC = (int64_t)F * C;
    if (C % 8 == 0 && al16(x) && al16(y))
        hipLaunchKernelGGL((avgpool_t2_fwd_h_kernel<float>), dim3(ew_grid((int64_t)N * (T / 2) * (FC / 8), 256)), dim3(256), 0, S(stream), x,
                           (int64_t)N, T, FC / 8, mask, y);
    else
        hipLaunchKernelGGL(avgpool_t2_fwd_kernel, dim3(ew_grid((int64_t)N * (T / 2) * FC, 256)), dim3(256), 0, S(stream), x, (int64_t)N,
                           T, FC, y);
    MQ_LAUNCH_CHECK("mq_avgpool_t2_fwd");
    return 0;
}
extern "C" int mq_avgpool_t2_bwd(const float *dy, int N, int T, int F, int C, const uint8_t *mask, float *dx, mqStream stream)
{
    MQ_REQUIRE(dy && dx && N > 0 && T > 0 && T % 2 == 0 && F > 0 && C > 0, "mq_avgpool_t2_bwd: bad arguments (T must be even)");
    MQ_REQUIRE(!mask || (C % 8 == 0 && al16(dy) && al16(dx)), "mq_avgpool_t2_bwd: the fused output mask needs C %% 8 == 0 (C=%d)", C);
    const int64_t FC = (int64_t)F * C;
    if (C % 8 == 0 && al16(dy) && al16(dx))
        hipLaunchKernelGGL((avgpool_t2_bwd_h_kernel<float>), dim3(ew_grid((int64_t)N * (T / 2) * (FC / 8), 256)), dim3(256), 0, S(stream), dy,
                           (int64_t)N, T, FC / 8, mask, dx);
    else
        hipLaunchKernelGGL(avgpool_t2_bwd_kernel, dim3(ew_grid((int64_t)N * T * FC, 256)), dim3(256), 0, S(stream), dy, (int64_t)N, T, FC,
                           dx);
    MQ_LAUNCH_CHECK("mq_avgpool_t2_bwd");
    return 0;
}
extern "C" int mq_upcat_fwd(const float *x, const float *skip, int N, int T, int F, int C1, int C2, const uint8_t *mask, float *out,
                            mqStream stream)
{
    MQ_REQUIRE(x && skip && out && N > 0 && T > 0 && F > 0 && C1 > 0 && C2 > 0, "mq_upcat_fwd: bad arguments");
    MQ_REQUIRE(!mask || (C1 % 8 == 0 && C2 % 8 == 0 && al16(x) && al16(skip) && al16(out)),
               "mq_upcat_fwd: the fused output mask needs C1, C2 multiples of 8 (%d, %d)", C1, C2);
    if (C1 % 8 == 0 && C2 % 8 == 0 && al16(x) && al16(skip) && al16(out))
        hipLaunchKernelGGL((upcat_fwd_h_kernel<float>), dim3(ew_grid((int64_t)N * 2 * T * F * ((C1 + C2) / 8), 256)), dim3(256), 0, S(stream), x,
                           skip, (int64_t)N, T, F, C1, C2, mask, out);
    else
        hipLaunchKernelGGL(upcat_fwd_kernel, dim3(ew_grid((int64_t)N * 2 * T * F * (C1 + C2), 256)), dim3(256), 0, S(stream), x, skip,
                           (int64_t)N, T, F, C1, C2, out);
    MQ_LAUNCH_CHECK("mq_upcat_fwd");
    return 0;
}
extern "C" int mq_upcat_bwd(const float *dout, int N, int T, int F, int C1, int C2, const uint8_t *mask, float *dx, float *dskip,
                            mqStream stream)
{
    MQ_REQUIRE(dout && dx && dskip && N > 0 && T > 0 && F > 0 && C1 > 0 && C2 > 0, "mq_upcat_bwd: bad arguments");
    MQ_REQUIRE(!mask || (C1 % 8 == 0 && C2 % 8 == 0 && al16(dout) && al16(dx) && al16(dskip)),
               "mq_upcat_bwd: the fused output mask needs C1, C2 multiples of 8 (%d, %d)", C1, C2);
    if (C1 % 8 == 0 && C2 % 8 == 0 && al16(dout) && al16(dx) && al16(dskip)) {
        hipLaunchKernelGGL((upcat_bwd_h_kernel<float>), dim3(ew_grid((int64_t)N * T * F * ((C1 + C2) / 8), 256)), dim3(256), 0, S(stream), dout,
                           (int64_t)N, T, F, C1, C2, mask, dx, dskip);
    } else {
        hipLaunchKernelGGL(upcat_bwd_x_kernel, dim3(ew_grid((int64_t)N * T * F * C1, 256)), dim3(256), 0, S(stream), dout, (int64_t)N, T,
                           F, C1, C2, dx);
        hipLaunchKernelGGL(upcat_bwd_skip_kernel, dim3(ew_grid((int64_t)N * 2 * T * F * C2, 256)), dim3(256), 0, S(stream), dout,
                           (int64_t)N * 2 * T * F, C1, C2, dskip);
    }
    MQ_LAUNCH_CHECK("mq_upcat_bwd");
    return 0;
}
extern "C" int mq_concat2(const float *a, const float *b, int64_t rows, int C1, int C2, float *y, mqStream stream)
{
    MQ_REQUIRE(a && b && y && rows > 0 && C1 > 0 && C2 > 0, "mq_concat2: bad arguments");
    hipLaunchKernelGGL(concat2_kernel, dim3(ew_grid(rows * (C1 + C2), 256)), dim3(256), 0, S(stream), a, b, rows, C1, C2, y);
    MQ_LAUNCH_CHECK("mq_concat2");
    return 0;
}
extern "C" int mq_axpby(const float *a, const float *b, float alpha, float beta_, int64_t n, float *out, mqStream stream)
{
    MQ_REQUIRE(a && out && n > 0, "mq_axpby: bad arguments");
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), a, b, alpha, beta_, n, out);
    MQ_LAUNCH_CHECK("mq_axpby");
    return 0;
}
extern "C" int mq_band_split_fwd(const float *x, int B, int T, int F, int n_bins, float *out, mqStream stream)
{
    MQ_REQUIRE(x && out && B > 0 && T > 0 && F > 0 && n_bins > 0 && F % n_bins == 0, "mq_band_split_fwd: bad arguments");
    MQ_REQUIRE(B <= 65535, "mq_band_split_fwd: B too large");
    hipLaunchKernelGGL((band_split_kernel<true>), dim3((unsigned)cdiv(T, 32), (unsigned)cdiv(F, 32), B), dim3(32, 8), 0, S(stream), x,
                       B, T, F, n_bins, out);
    MQ_LAUNCH_CHECK("mq_band_split_fwd");
    return 0;
}
extern "C" int mq_band_split_bwd(const float *dout, int B, int T, int F, int n_bins, float *dx, mqStream stream)
{
    MQ_REQUIRE(dout && dx && B > 0 && T > 0 && F > 0 && n_bins > 0 && F % n_bins == 0, "mq_band_split_bwd: bad arguments");
    MQ_REQUIRE(B <= 65535, "mq_band_split_bwd: B too large");
    hipLaunchKernelGGL((band_split_kernel<false>), dim3((unsigned)cdiv(T, 32), (unsigned)cdiv(F, 32), B), dim3(32, 8), 0, S(stream),
                       dout, B, T, F, n_bins, dx);
    MQ_LAUNCH_CHECK("mq_band_split_bwd");
    return 0;
}
extern "C" int mq_sigmoid_add(const float *a, const float *b, int64_t n, float *out, mqStream stream)
{
    MQ_REQUIRE(a && out && n > 0, "mq_sigmoid_add: bad arguments");
    hipLaunchKernelGGL(sigmoid_add_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), a, b, n, out);
    MQ_LAUNCH_CHECK("mq_sigmoid_add");
    return 0;
}
extern "C" int mq_sigmoid_add_bwd(const float *dout, const float *out, int64_t n, float *da, mqStream stream)
{
    MQ_REQUIRE(dout && out && da && n > 0, "mq_sigmoid_add_bwd: bad arguments");
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), dout, out, n, da);
    MQ_LAUNCH_CHECK("mq_sigmoid_add_bwd");
    return 0;
}
extern "C" int mq_relu_fwd(const float *x, int64_t n, float *y, mqStream stream)
{
    MQ_REQUIRE(x && y && n > 0, "mq_relu_fwd: bad arguments");
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), x, n, y);
    MQ_LAUNCH_CHECK("mq_relu_fwd");
    return 0;
}
extern "C" int mq_relu_bwd(const float *dy, const float *y, int64_t n, float *dx, mqStream stream)
{
    MQ_REQUIRE(dy && y && dx && n > 0, "mq_relu_bwd: bad arguments");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), dy, y, n, dx);
    MQ_LAUNCH_CHECK("mq_relu_bwd");
    return 0;
}

extern "C" int mq_cast_bf16(const float *in, uint16_t *out, int64_t n, mqStream stream)
{
    if (n == 0) return 0;
    MQ_REQUIRE(in && out && n > 0, "mq_cast_bf16: bad arguments");
    MQ_REQUIRE((uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0, "mq_cast_bf16: pointers must be 16-byte aligned");
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(ew_grid(n8 > 0 ? n8 : 1, 256)), dim3(256), 0, S(stream), in, out, n8, n);
    MQ_LAUNCH_CHECK("mq_cast_bf16");
    return 0;
}

extern "C" int mq_cast_f32(const uint16_t *in, float *out, int64_t n, mqStream stream)
{
    if (n == 0) return 0;
    MQ_REQUIRE(in && out && n > 0, "mq_cast_f32: bad arguments");
    MQ_REQUIRE((uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0, "mq_cast_f32: pointers must be 16-byte aligned");
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(cast_f32_kernel, dim3(ew_grid(n8 > 0 ? n8 : 1, 256)), dim3(256), 0, S(stream), in, out, n8, n);
    MQ_LAUNCH_CHECK("mq_cast_f32");
    return 0;
}

extern "C" int mq_pointwise_fwd_h(const mqPointwiseDesc *d, const uint16_t *h, const uint16_t *r_pre, const uint16_t *r_post,
                                  const uint8_t *mask, const float *beta, const float *gamma, uint16_t *out, mqStream stream)
{
    PwParams p;
    if (make_pw(d, p, "mq_pointwise_fwd_h")) return 1;
    MQ_REQUIRE(h && out, "mq_pointwise_fwd_h: null pointer");
    MQ_REQUIRE(mask || !(p.mask_pre || p.mask_post), "mq_pointwise_fwd_h: mask flags set but mask is null");
    MQ_REQUIRE(p.C % 8 == 0 && al16(h) && al16(r_pre) && al16(r_post) && al16(out),
               "mq_pointwise_fwd_h: bf16 tensors need C %% 8 == 0 and 16-byte alignment (C=%d)", p.C);
    const int64_t groups = p.rows * p.C / 8;
    const int grid = ew_grid(groups, 256);
    PwFast f;
    if (!pw_generic() && make_pw_fast(p, f))
        MQ_PW_V8(pointwise_fwd_v8_kernel, bf16_t, grid, p, f, h, r_pre, r_post, mask, beta, gamma, out);
    else
        hipLaunchKernelGGL((pointwise_fwd_kernel<true, bf16_t>), dim3(grid), dim3(256), 0, S(stream), p, h, r_pre, r_post, mask, beta,
                           gamma, out);
    MQ_LAUNCH_CHECK("mq_pointwise_fwd_h");
    return 0;
}

extern "C" int mq_pointwise_bwd_h(const mqPointwiseDesc *d, const uint16_t *dout, const uint16_t *h, const uint16_t *r_pre,
                                  const uint8_t *mask, const float *beta, const float *gamma, uint16_t *dt, uint16_t *dpost,
                                  float *dbeta, float *dgamma, float *workspace, mqStream stream)
{
    PwParams p;
    if (make_pw(d, p, "mq_pointwise_bwd_h")) return 1;
    MQ_REQUIRE(dout && h && dt, "mq_pointwise_bwd_h: null pointer");
    MQ_REQUIRE(mask || !(p.mask_pre || p.mask_post), "mq_pointwise_bwd_h: mask flags set but mask is null");
    const bool want_bg = dbeta || dgamma;
    MQ_REQUIRE(!want_bg || workspace, "mq_pointwise_bwd_h: dbeta/dgamma need a workspace");
    MQ_REQUIRE(p.C % 8 == 0 && al16(h) && al16(r_pre) && al16(dout) && al16(dt) && al16(dpost),
               "mq_pointwise_bwd_h: bf16 tensors need C %% 8 == 0 and 16-byte alignment (C=%d)", p.C);
    const int64_t groups = p.rows * p.C / 8;
    int grid = ew_grid(groups, 256);
    if (grid > 1024) grid = 1024;
    float *part = want_bg ? workspace : nullptr;
    PwFast f;
    if (!pw_generic() && make_pw_fast(p, f))
        MQ_PW_V8(pointwise_bwd_v8_kernel, bf16_t, grid, p, f, dout, h, r_pre, mask, beta, gamma, dt, dpost, part);
    else
        hipLaunchKernelGGL((pointwise_bwd_kernel<true, bf16_t>), dim3(grid), dim3(256), 0, S(stream), p, dout, h, r_pre, mask, beta,
                           gamma, dt, dpost, part);
    if (want_bg) hipLaunchKernelGGL(pw_final_kernel, dim3(1), dim3(256), 0, S(stream), part, grid, dbeta, dgamma);
    MQ_LAUNCH_CHECK("mq_pointwise_bwd_h");
    return 0;
}

extern "C" int mq_avgpool_t2_fwd_h(const uint16_t *x, int N, int T, int F, int C, const uint8_t *mask, uint16_t *y, mqStream stream)
{
    MQ_REQUIRE(x && y && N > 0 && T > 0 && T % 2 == 0 && F > 0 && C > 0 && C % 8 == 0 && al16(x) && al16(y),
               "mq_avgpool_t2_fwd_h: bad arguments (T even, C %% 8 == 0, 16-byte alignment)");
    const int64_t FC8 = (int64_t)F * C / 8;
    hipLaunchKernelGGL((avgpool_t2_fwd_h_kernel<bf16_t>), dim3(ew_grid((int64_t)N * (T / 2) * FC8, 256)), dim3(256), 0, S(stream), x, (int64_t)N,
                       T, FC8, mask, y);
    MQ_LAUNCH_CHECK("mq_avgpool_t2_fwd_h");
    return 0;
}
extern "C" int mq_avgpool_t2_bwd_h(const uint16_t *dy, int N, int T, int F, int C, const uint8_t *mask, uint16_t *dx, mqStream stream)
{
    MQ_REQUIRE(dy && dx && N > 0 && T > 0 && T % 2 == 0 && F > 0 && C > 0 && C % 8 == 0 && al16(dy) && al16(dx),
               "mq_avgpool_t2_bwd_h: bad arguments (T even, C %% 8 == 0, 16-byte alignment)");
    const int64_t FC8 = (int64_t)F * C / 8;
    hipLaunchKernelGGL((avgpool_t2_bwd_h_kernel<bf16_t>), dim3(ew_grid((int64_t)N * (T / 2) * FC8, 256)), dim3(256), 0, S(stream), dy, (int64_t)N,
                       T, FC8, mask, dx);
    MQ_LAUNCH_CHECK("mq_avgpool_t2_bwd_h");
    return 0;
}
extern "C" int mq_upcat_fwd_h(const uint16_t *x, const uint16_t *skip, int N, int T, int F, int C1, int C2, const uint8_t *mask,
                              uint16_t *out, mqStream stream)
{
    MQ_REQUIRE(x && skip && out && N > 0 && T > 0 && F > 0 && C1 > 0 && C2 > 0 && C1 % 8 == 0 && C2 % 8 == 0 && al16(x) && al16(skip) && al16(out),
               "mq_upcat_fwd_h: bad arguments (C1, C2 multiples of 8, 16-byte alignment)");
    hipLaunchKernelGGL((upcat_fwd_h_kernel<bf16_t>), dim3(ew_grid((int64_t)N * 2 * T * F * ((C1 + C2) / 8), 256)), dim3(256), 0, S(stream), x, skip,
                       (int64_t)N, T, F, C1, C2, mask, out);
    MQ_LAUNCH_CHECK("mq_upcat_fwd_h");
    return 0;
}
extern "C" int mq_upcat_bwd_h(const uint16_t *dout, int N, int T, int F, int C1, int C2, const uint8_t *mask, uint16_t *dx,
                              uint16_t *dskip, mqStream stream)
{
    MQ_REQUIRE(dout && dx && dskip && N > 0 && T > 0 && F > 0 && C1 > 0 && C2 > 0 && C1 % 8 == 0 && C2 % 8 == 0 && al16(dout) && al16(dx) && al16(dskip),
               "mq_upcat_bwd_h: bad arguments (C1, C2 multiples of 8, 16-byte alignment)");
    hipLaunchKernelGGL((upcat_bwd_h_kernel<bf16_t>), dim3(ew_grid((int64_t)N * T * F * ((C1 + C2) / 8), 256)), dim3(256), 0, S(stream), dout,
                       (int64_t)N, T, F, C1, C2, mask, dx, dskip);
    MQ_LAUNCH_CHECK("mq_upcat_bwd_h");
    return 0;
}
