// Implicit-GEMM convolution (fprop / dgrad / wgrad) on f32 MFMA for gfx950, NHWC.
// See include/mqgan_hip.h (mq_conv_*) for the interface and gemm_core.h for the tile engine.
#include "gemm_core.h"

namespace mq {

struct ConvGeom {
    // GEMM view:  rows m = (n, oh, ow) of the OUTPUT of this pass, cols = Ncols, reduction kk = (r, s, ic)
    int OH, OW;            // spatial dims the rows enumerate
    int IH, IW, IC;        // dims of the gathered activation tensor
    int R, S;
    int sh, sw, ph, pw, dh, dw;
    int Mtot;              // rows per group
    int Ncols, Kdim;
    int nMt, nNt;
};

// ------------------------------------------------------------------------------------------ loaders
// A operand: gathered activations.  DGRAD = false: ih = oh*sh - ph + r*dh.  DGRAD = true: the gathered
// tensor is dy (N,P,Q,K) and ih = (oh + ph - r*dh) / sh when divisible.
template <bool DGRAD, bool VEC>
struct ConvALoader {
    const float *__restrict__ src;   // group base of the gathered tensor
    ConvGeom g;
    int n_[2], bh_[2], bw_[2];
    bool rv_[2];
    int kq;
    // per K-step decode (VEC): one (r, s, c) for the thread's 4 consecutive k
    int cr, cs, cc;
    bool kv;
    int kbase;

    __device__ __forceinline__ void init(const float *s, const ConvGeom &gg, int m0)
    {
        src = s;
        g = gg;
        kq = threadIdx.x & 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int m = m0 + (threadIdx.x >> 2) + 64 * j;
            rv_[j] = m < g.Mtot;
            int mm = rv_[j] ? m : 0;
            int ow = mm % g.OW;
            int t = mm / g.OW;
            int oh = t % g.OH;
            n_[j] = t / g.OH;
            if (DGRAD) {
                bh_[j] = oh + g.ph;
                bw_[j] = ow + g.pw;
            } else {
                bh_[j] = oh * g.sh - g.ph;
                bw_[j] = ow * g.sw - g.pw;
            }
        }
    }
    __device__ __forceinline__ void prepare(int k0)
    {
        kbase = k0 + 4 * kq;
        if (VEC) {
            kv = kbase < g.Kdim;
            int tap = kbase / g.IC;
            cc = kbase - tap * g.IC;
            cr = tap / g.S;
            cs = tap - cr * g.S;
        }
    }
    __device__ __forceinline__ bool coords(int j, int r, int s, int &ih, int &iw) const
    {
        if (DGRAD) {
            int th = bh_[j] - r * g.dh, tw = bw_[j] - s * g.dw;
            if (th < 0 || tw < 0) return false;
            ih = th / g.sh;
            iw = tw / g.sw;
            if (ih * g.sh != th || iw * g.sw != tw) return false;
            return ih < g.IH && iw < g.IW;
        } else {
            ih = bh_[j] + r * g.dh;
            iw = bw_[j] + s * g.dw;
            return (unsigned)ih < (unsigned)g.IH && (unsigned)iw < (unsigned)g.IW;
        }
    }
    __device__ __forceinline__ float4 load(int j) const
    {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!rv_[j]) return v;
        if (VEC) {
            int ih, iw;
            if (kv && coords(j, cr, cs, ih, iw)) {
                size_t off = (((size_t)n_[j] * g.IH + ih) * g.IW + iw) * g.IC + cc;
                v = *reinterpret_cast<const float4 *>(src + off);
            }
        } else {
            float e[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                e[t] = 0.f;
                int kk = kbase + t;
                if (kk < g.Kdim) {
                    int tap = kk / g.IC, c = kk - tap * g.IC;
                    int r = tap / g.S, s = tap - r * g.S;
                    int ih, iw;
                    if (coords(j, r, s, ih, iw)) e[t] = src[(((size_t)n_[j] * g.IH + ih) * g.IW + iw) * g.IC + c];
                }
            }
            v = make_float4(e[0], e[1], e[2], e[3]);
        }
        return v;
    }
};

// B operand: dense row-major [Ncols][Kdim] filter bank of one group
template <bool VEC>
struct DenseBLoader {
    const float *__restrict__ w;
    int n0, Ncols, Kdim, kbase;
    __device__ __forceinline__ void init(const float *ww, int n0_, int Ncols_, int Kdim_)
    {
        w = ww;
        n0 = n0_;
        Ncols = Ncols_;
        Kdim = Kdim_;
    }
    __device__ __forceinline__ void prepare(int k0) { kbase = k0 + 4 * (threadIdx.x & 3); }
    __device__ __forceinline__ float4 load(int j) const
    {
        int n = n0 + (threadIdx.x >> 2) + 64 * j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n >= Ncols) return v;
        const float *p = w + (size_t)n * Kdim + kbase;
        if (VEC) {
            if (kbase < Kdim) v = *reinterpret_cast<const float4 *>(p);
        } else {
            if (kbase + 0 < Kdim) v.x = p[0];
            if (kbase + 1 < Kdim) v.y = p[1];
            if (kbase + 2 < Kdim) v.z = p[2];
            if (kbase + 3 < Kdim) v.w = p[3];
        }
        return v;
    }
};

// ------------------------------------------------------------------------------------------ fprop / dgrad
template <int BN, bool DGRAD, bool VEC>
__global__ __launch_bounds__(256) void conv_nt_kernel(const float *__restrict__ act, const float *__restrict__ wgt,
                                                      const float *__restrict__ bias, float *__restrict__ out,
                                                      ConvGeom g, size_t act_group_stride, size_t w_group_stride)
{
    using Cfg = GemmCfg<BN>;
    __shared__ __attribute__((aligned(16))) float smem[Cfg::LDS_FLOATS];
    const int grp = blockIdx.y;
    const int tile = xcd_remap(blockIdx.x, g.nMt * g.nNt);
    const int mt = tile / g.nNt, nt = tile - mt * g.nNt;   // N tiles of one M tile are neighbours (share A in L2)
    const int m0 = mt * GEMM_BM, n0 = nt * BN;

    ConvALoader<DGRAD, VEC> la;
    la.init(act + (size_t)grp * act_group_stride, g, m0);
    DenseBLoader<VEC> lb;
    lb.init(wgt + (size_t)grp * w_group_stride, n0, g.Ncols, g.Kdim);

    f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int a = 0; a < Cfg::TM; ++a)
#pragma unroll
        for (int b = 0; b < Cfg::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    nt_gemm_mainloop<BN>(la, lb, (g.Kdim + GEMM_BK - 1) / GEMM_BK, smem, acc);

    float *o = out + (size_t)grp * g.Mtot * g.Ncols;
#pragma unroll
    for (int tn = 0; tn < Cfg::TN; ++tn) {
        const int n = n0 + acc_col<BN>(tn);
        if (n >= g.Ncols) continue;
        const float bv = bias ? bias[(size_t)grp * g.Ncols + n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + acc_row<BN>(tm, r);
                if (m < g.Mtot) o[(size_t)m * g.Ncols + n] = acc[tm][tn][r] + bv;
            }
    }
}

// ------------------------------------------------------------------------------------------ wgrad
// dw[k][kk] = sum_m dy[m][k] * xg[m][kk], kk = (r, s, c), reduction over the pixels m of one group.
// Both operands are "M/N contiguous" in memory, so the LDS image is [pixel][row] and the MFMA operands
// are read with ds_read_b64 as (row 2i, row 2i+1): tile t of a wave covers rows {2i + t}.
constexpr int WG_BK = 16;          // pixels per step
constexpr int WG_LD = 132;         // 128 + 4 pad (keeps float4 stores aligned, shifts banks per pixel row)

struct WgradGeom {
    int OH, OW;            // P, Q
    int IH, IW, IC;        // H, W, C
    int R, S, sh, sw, ph, pw, dh, dw;
    int Mtot;              // pixels per group
    int K, RSC;
    int nKt, nCt;          // tiles along K (rows) and RSC (cols)
    int splits, chunk;     // split-K: pixels per split (multiple of WG_BK)
};

template <bool VEC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                         float *__restrict__ out, WgradGeom g, int groups)
{
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * WG_BK * WG_LD];
    float *At = smem;                       // [2][WG_BK][WG_LD]  dy tile   (pixel, k)
    float *Bt = smem + 2 * WG_BK * WG_LD;   // [2][WG_BK][WG_LD]  x gather  (pixel, kk)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int tile = blockIdx.x;
    const int kt_ = tile / g.nCt, ct = tile - kt_ * g.nCt;
    const int k0 = kt_ * 128, c0 = ct * 128;
    const int grp = blockIdx.y;
    const int split = blockIdx.z;
    const int m_begin = split * g.chunk;
    const int m_end = min(g.Mtot, m_begin + g.chunk);
    const float *dyg = dy + (size_t)grp * g.Mtot * g.K;
    const size_t x_group = (size_t)grp * (g.Mtot / (g.OH * g.OW)) * g.IH * g.IW * g.IC;
    const float *xg = x + x_group;

    // loader mapping: pixel = tid/32 + 8*j (j = 0,1), column quad = tid%32
    const int lp = tid >> 5, lq = tid & 31;
    // the thread's 4 B columns are fixed for the whole kernel
    const int kkb = c0 + 4 * lq;
    int br[4], bs[4], bc[4];
    bool bv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int kk = kkb + t;
        bv[t] = kk < g.RSC;
        int kc = bv[t] ? kk : 0;
        int tap = kc / g.IC;
        bc[t] = kc - tap * g.IC;
        br[t] = tap / g.S;
        bs[t] = tap - br[t] * g.S;
    }
    const int ka = k0 + 4 * lq;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[2], rb[2];
    auto gload = [&](int mstep) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = mstep + lp + 8 * j;
            ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end) {
                const float *pa = dyg + (size_t)m * g.K + ka;
                if (VEC) {
                    if (ka < g.K) ra[j] = *reinterpret_cast<const float4 *>(pa);
                } else {
                    if (ka + 0 < g.K) ra[j].x = pa[0];
                    if (ka + 1 < g.K) ra[j].y = pa[1];
                    if (ka + 2 < g.K) ra[j].z = pa[2];
                    if (ka + 3 < g.K) ra[j].w = pa[3];
                }
                const int ow = m % g.OW;
                const int t2 = m / g.OW;
                const int oh = t2 % g.OH, n = t2 / g.OH;
                const int bh = oh * g.sh - g.ph, bw = ow * g.sw - g.pw;
                if (VEC) {
                    const int ih = bh + br[0] * g.dh, iw = bw + bs[0] * g.dw;
                    if (bv[0] && (unsigned)ih < (unsigned)g.IH && (unsigned)iw < (unsigned)g.IW)
                        rb[j] = *reinterpret_cast<const float4 *>(
                            xg + (((size_t)n * g.IH + ih) * g.IW + iw) * g.IC + bc[0]);
                } else {
                    float e[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        e[t] = 0.f;
                        const int ih = bh + br[t] * g.dh, iw = bw + bs[t] * g.dw;
                        if (bv[t] && (unsigned)ih < (unsigned)g.IH && (unsigned)iw < (unsigned)g.IW)
                            e[t] = xg[(((size_t)n * g.IH + ih) * g.IW + iw) * g.IC + bc[t]];
                    }
                    rb[j] = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<float4 *>(&At[(buf * WG_BK + lp + 8 * j) * WG_LD + 4 * lq]) = ra[j];
            *reinterpret_cast<float4 *>(&Bt[(buf * WG_BK + lp + 8 * j) * WG_LD + 4 * lq]) = rb[j];
        }
    };

    const int nsteps = (m_end - m_begin + WG_BK - 1) / WG_BK;
    if (nsteps > 0) {
        gload(m_begin);
        sstore(0);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        const bool more = st + 1 < nsteps;
        if (more) gload(m_begin + (st + 1) * WG_BK);
        const float *Ac = At + cur * WG_BK * WG_LD + wm * 64 + 2 * li;
        const float *Bc = Bt + cur * WG_BK * WG_LD + wn * 64 + 2 * li;
#pragma unroll
        for (int s = 0; s < WG_BK / 2; ++s) {
            const float2 a = *reinterpret_cast<const float2 *>(Ac + (2 * s + lh) * WG_LD);
            const float2 b = *reinterpret_cast<const float2 *>(Bc + (2 * s + lh) * WG_LD);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1][1], 0, 0, 0);
        }
        if (more) sstore(cur ^ 1);
        __syncthreads();
    }

    // slab [split][group][K][RSC]
    float *o = out + ((size_t)split * groups + grp) * (size_t)g.K * g.RSC;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = c0 + wn * 64 + 2 * li + tn;
            if (col >= g.RSC) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = k0 + wm * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lh) + tm;
                if (row < g.K) o[(size_t)row * g.RSC + col] = acc[tm][tn][r];
            }
        }
}

__global__ void slab_reduce_kernel(const float *__restrict__ slabs, float *__restrict__ out, size_t n, int splits)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += slabs[(size_t)k * n + i];
        out[i] = s;
    }
}

// ------------------------------------------------------------------------------------------ column sums
__global__ void colsum_partial_kernel(const float *__restrict__ x, int64_t rows, int cols, float *__restrict__ part,
                                      int nchunks)
{
    // grid (ceil(cols/64), nchunks); block (64, 4): thread (c, ry) sums rows ry, ry+4, ... of its chunk
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int64_t per = (rows + nchunks - 1) / nchunks;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(rows, r0 + per);
    float s = 0.f;
    if (c < cols)
        for (int64_t r = r0 + threadIdx.y; r < r1; r += 4) s += x[r * cols + c];
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && c < cols)
        part[(size_t)blockIdx.y * cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void colsum_final_kernel(const float *__restrict__ part, int cols, int nchunks, float *__restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int k = 0; k < nchunks; ++k) s += part[(size_t)k * cols + c];
    out[c] = s;
}

// ------------------------------------------------------------------------------------------ host side
static int check_desc(const mqConvDesc *d, const char *who)
{
    MQ_REQUIRE(d != nullptr, "%s: null descriptor", who);
    MQ_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0 && d->P > 0 && d->Q > 0,
               "%s: non-positive dimension", who);
    MQ_REQUIRE(d->stride_h > 0 && d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0 && d->pad_h >= 0 && d->pad_w >= 0,
               "%s: bad stride/dilation/padding", who);
    MQ_REQUIRE(d->groups > 0 && d->N % d->groups == 0, "%s: N=%d not divisible by groups=%d", who, d->N, d->groups);
    // every output pixel must read a window that starts inside the padded input
    MQ_REQUIRE((int64_t)(d->P - 1) * d->stride_h - d->pad_h < d->H && (int64_t)(d->Q - 1) * d->stride_w - d->pad_w < d->W,
               "%s: output extent (%d,%d) inconsistent with input (%d,%d)", who, d->P, d->Q, d->H, d->W);
    MQ_REQUIRE((int64_t)d->N * d->H * d->W * d->C < (int64_t)1 << 40 && (int64_t)d->N * d->P * d->Q < INT32_MAX,
               "%s: tensor too large", who);
    return 0;
}

template <bool DGRAD>
static int launch_nt(const mqConvDesc *d, const float *act, const float *w, const float *bias, float *out, hipStream_t st)
{
    ConvGeom g;
    const int ng = d->N / d->groups;
    if (!DGRAD) {
        g.OH = d->P; g.OW = d->Q; g.IH = d->H; g.IW = d->W; g.IC = d->C;
        g.Ncols = d->K;
    } else {
        g.OH = d->H; g.OW = d->W; g.IH = d->P; g.IW = d->Q; g.IC = d->K;
        g.Ncols = d->C;
    }
    g.R = d->R; g.S = d->S; g.sh = d->stride_h; g.sw = d->stride_w; g.ph = d->pad_h; g.pw = d->pad_w;
    g.dh = d->dil_h; g.dw = d->dil_w;
    g.Mtot = ng * g.OH * g.OW;
    g.Kdim = d->R * d->S * g.IC;
    const size_t act_gs = (size_t)ng * g.IH * g.IW * g.IC;
    const size_t w_gs = (size_t)g.Ncols * g.Kdim;
    const bool vec = (g.IC % 4 == 0) && ((uintptr_t)act % 16 == 0) && ((uintptr_t)w % 16 == 0);
    const int BN = g.Ncols > 64 ? 128 : (g.Ncols > 32 ? 64 : 32);
    g.nMt = (int)cdiv(g.Mtot, GEMM_BM);
    g.nNt = (int)cdiv(g.Ncols, BN);
    dim3 grid(g.nMt * g.nNt, d->groups), block(256);
#define MQ_NT(BN_, V_) hipLaunchKernelGGL((conv_nt_kernel<BN_, DGRAD, V_>), grid, block, 0, st, act, w, bias, out, g, act_gs, w_gs)
    if (BN == 128) { if (vec) MQ_NT(128, true); else MQ_NT(128, false); }
    else if (BN == 64) { if (vec) MQ_NT(64, true); else MQ_NT(64, false); }
    else { if (vec) MQ_NT(32, true); else MQ_NT(32, false); }
#undef MQ_NT
    MQ_LAUNCH_CHECK(DGRAD ? "mq_conv_dgrad" : "mq_conv_fprop");
    return 0;
}

static void wgrad_plan(const mqConvDesc *d, WgradGeom &g)
{
    const int ng = d->N / d->groups;
    g.OH = d->P; g.OW = d->Q; g.IH = d->H; g.IW = d->W; g.IC = d->C;
    g.R = d->R; g.S = d->S; g.sh = d->stride_h; g.sw = d->stride_w; g.ph = d->pad_h; g.pw = d->pad_w;
    g.dh = d->dil_h; g.dw = d->dil_w;
    g.Mtot = ng * d->P * d->Q;
    g.K = d->K;
    g.RSC = d->R * d->S * d->C;
    g.nKt = (int)cdiv(g.K, 128);
    g.nCt = (int)cdiv(g.RSC, 128);
    // split the pixel reduction so that ~3 blocks per CU exist, each with >= 32 steps
    const int64_t tiles = (int64_t)g.nKt * g.nCt * d->groups;
    int64_t want = cdiv(256 * 3, tiles);
    int64_t max_splits = cdiv(g.Mtot, WG_BK * 32);
    int64_t splits = want < max_splits ? want : max_splits;
    if (splits < 1) splits = 1;
    if (splits > 512) splits = 512;
    int64_t chunk = cdiv(cdiv(g.Mtot, splits), WG_BK) * WG_BK;
    g.chunk = (int)chunk;
    g.splits = (int)cdiv(g.Mtot, chunk);
}

}  // namespace mq

using namespace mq;

extern "C" int mq_conv_fprop(const mqConvDesc *d, const float *x, const float *w, const float *bias, float *y,
                             mqStream stream)
{
    if (check_desc(d, "mq_conv_fprop")) return 1;
    MQ_REQUIRE(x && w && y, "mq_conv_fprop: null pointer");
    return launch_nt<false>(d, x, w, bias, y, S(stream));
}

extern "C" int mq_conv_dgrad(const mqConvDesc *d, const float *dy, const float *w_crsk, float *dx, mqStream stream)
{
    if (check_desc(d, "mq_conv_dgrad")) return 1;
    MQ_REQUIRE(dy && w_crsk && dx, "mq_conv_dgrad: null pointer");
    return launch_nt<true>(d, dy, w_crsk, nullptr, dx, S(stream));
}

extern "C" size_t mq_conv_wgrad_workspace(const mqConvDesc *d)
{
    if (check_desc(d, "mq_conv_wgrad_workspace")) return 0;
    WgradGeom g;
    wgrad_plan(d, g);
    if (g.splits <= 1) return 0;
    return (size_t)g.splits * d->groups * g.K * g.RSC * sizeof(float);
}

extern "C" int mq_conv_wgrad(const mqConvDesc *d, const float *x, const float *dy, float *dw, void *workspace,
                             size_t workspace_bytes, mqStream stream)
{
    if (check_desc(d, "mq_conv_wgrad")) return 1;
    MQ_REQUIRE(x && dy && dw, "mq_conv_wgrad: null pointer");
    WgradGeom g;
    wgrad_plan(d, g);
    const size_t n = (size_t)d->groups * g.K * g.RSC;
    const size_t need = g.splits > 1 ? (size_t)g.splits * n * sizeof(float) : 0;
    MQ_REQUIRE(workspace_bytes >= need && (need == 0 || workspace != nullptr),
               "mq_conv_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
    float *target = g.splits > 1 ? reinterpret_cast<float *>(workspace) : dw;
    const bool vec = (d->C % 4 == 0) && (d->K % 4 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dy % 16 == 0);
    dim3 grid(g.nKt * g.nCt, d->groups, g.splits), block(256);
    if (vec)
        hipLaunchKernelGGL((conv_wgrad_kernel<true>), grid, block, 0, S(stream), x, dy, target, g, d->groups);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<false>), grid, block, 0, S(stream), x, dy, target, g, d->groups);
    MQ_LAUNCH_CHECK("mq_conv_wgrad");
    if (g.splits > 1) {
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(ew_grid((int64_t)n, 256)), dim3(256), 0, S(stream), target, dw, n,
                           g.splits);
        MQ_LAUNCH_CHECK("mq_conv_wgrad(reduce)");
    }
    return 0;
}

extern "C" int mq_colsum(const float *x, int64_t rows, int cols, float *out, float *workspace, mqStream stream)
{
    MQ_REQUIRE(x && out && workspace && rows > 0 && cols > 0, "mq_colsum: bad arguments");
    const int nchunks = (int)(rows < 64 ? 1 : 64);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)cdiv(cols, 64), nchunks), dim3(64, 4), 0, S(stream), x, rows,
                       cols, workspace, nchunks);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv(cols, 256)), dim3(256), 0, S(stream), workspace, cols,
                       nchunks, out);
    MQ_LAUNCH_CHECK("mq_colsum");
    return 0;
}
