// Implicit-GEMM convolution (fprop / dgrad / wgrad) on f32 MFMA for gfx950, NHWC.
// See include/mqgan_hip.h (mq_conv_*) for the interface and gemm_core.h for the tile engine.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "gemm_core.h"

#ifndef MQ_NT_OCC
#define MQ_NT_OCC 4        // resident blocks per CU of the f32 fprop / dgrad kernels
#endif

namespace mq {

// LDS-DMA f32 kernels (igemm_h2.hip): fprop / dgrad of the layers with C % 16 == 0 and enough rows for 256-row tiles
bool d32_nt_ok(const mqConvDesc *d, bool dgrad);
#if MQ_VARIANTS
int d32_nt_launch(const mqConvDesc *d, bool dgrad, const float *act, const float *w, const float *bias, const uint8_t *mask, int d1,
                  int d2, int d3, int act_kind, float slope, float *out, hipStream_t st);
#endif
// Measured (tools/f32_micro.py, round 2): on the f32 engine the LDS-DMA structure is SLOWER than the register-staged
// kernels below (126-135 vs 131-142 TF/s on the 256-768 channel layers): f32 MFMA runs at the vector rate, so neither L2
// bandwidth nor staging instructions bound it, and one 512-thread block per CU loses more to the tile tail than four
// 256-thread blocks do.  The f32 instantiations stay available for A/B runs (MQGAN_D32=1) but are off by default.
static bool use_d32()
{
#if MQ_VARIANTS
    static const bool on = getenv("MQGAN_D32") != nullptr;
    return on;
#else
    return false;                       // the LDS-DMA f32 kernels are compiled into the test-variant library only (make variants)
#endif
}

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

// Optional fused epilogue of the fprop kernels: y = padded(row) ? 0 : leaky(conv + bias), i.e. the discriminators'
// conv -> LeakyReLU -> masked_fill (discriminators.py:219,232) without a second pass over the output.  The frame mask
// index of output row m is (m / d1) * d2 + (m % d3), as in the pointwise kernels.
//
// act 3 is the matching dgrad epilogue: the conv's INPUT x was produced by such a layer (x = padded ? 0 : leaky(h)), and the
// gradient the kernel hands back is the one with respect to h: dx * (x > 0 ? 1 : slope), zero on padded frames - the
// producer's LeakyReLU / mask backward without a pass of its own.  `gate` is x, laid out like the output dx.
struct NtEpi {
    const uint8_t *mask;       // nullptr: no masking
    int d1, d2, d3;
    float slope;
    int act;                   // 0: plain store, 2: LeakyReLU(slope) (+ mask), 3 (dgrad): LeakyReLU'(gate) (+ mask)
    const float *gate;
};

// ------------------------------------------------------------------------------------------ loaders
// Vector path (IC % 4 == 0): every 16-byte load goes through a buffer descriptor, so padding taps,
// ragged tile edges and K tails are "offset = OOB" (zeros, no memory access) - no branches, no
// selects on data.  The reduction index walks a (chunk, tap row, tap col, channel-in-chunk) grid in
// the channel-blocked order of common.h; TapWalker keeps that position and the matching element
// offset incrementally: a K-step costs a handful of adds/compares per thread and no integer division.
//
// One walker serves three geometries (see make_walk() on the host):
//   fprop            : ih = (oh*sh - ph) + jr*dh           taps jr < R,  js < S
//   dgrad, stride 1  : ih = (oh + ph)    - jr*dh
//   dgrad, by parity : ih = ((ih' * sh + a + ph) - r0*dh)/sh - jr*(rstep*dh/sh)   taps jr < nr, js < ns
// i.e. always  ih = bh + jr*step_h,  iw = bw + js*step_w  with per-row (bh, bw) and uniform steps.
struct WalkGeom {
    int IH, IW, IC;        // gathered tensor
    int nR, nS;            // tap grid of this GEMM
    int step_h, step_w;    // signed coordinate step per tap
    int CB, nchunks;       // channel blocking of the reduction
};

// WRAPS = carries per K-step: ceil(BK / CB) for the smallest CB served (f32: 1 when CB >= 16, else 4;
// bf16: 2 when CB >= 16, else 8)
// UNIFORM (CB == BK == 16, i.e. C % 16 == 0 on the f32 engine): every K-step is exactly one (chunk, tap) for the
// whole block and only the channel quad differs per thread, so the position and its offsets are wave-uniform -
// the compiler keeps them in SGPRs and advances them on the scalar unit; per slot only the bounds test remains.
template <int WRAPS, class E = EngF32, bool UNIFORM = false>
struct TapWalker {
    static constexpr int NS = E::NS;
    __amdgpu_buffer_rsrc_t rsrc;
    WalkGeom w;
    int rowbase_[NS], bh_[NS], bw_[NS];   // per row slot: element offset of (n, bh, bw, 0) and the base coordinates
    bool rv_[NS];
    int ci, js, jr, chunk;             // position of the thread's k-quad
    int hoff, woff, tapoff;            // jr*step_h, js*step_w, (jr*step_h*IW + js*step_w)*IC
    unsigned off_[NS];

    __device__ __forceinline__ void set_row(int j, bool valid, int n, int bh, int bw)
    {
        rv_[j] = valid;
        bh_[j] = bh;
        bw_[j] = bw;
        rowbase_[j] = ((n * w.IH + bh) * w.IW + bw) * w.IC + (UNIFORM ? ci : 0);
    }
    __device__ __forceinline__ void start(const float *base, size_t elems, const WalkGeom &wg)
    {
        w = wg;
        rsrc = make_rsrc(base, elems * E::ESZ);               // (for 2-byte operands `base` is a bf16 array in disguise)
        const int kb = E::EPL * (threadIdx.x % E::QPR);       // first k of this thread in step 0
        if (UNIFORM) {
            // the geometry is the same for the whole block; say so, so that the walk stays in SGPRs even when it was
            // selected per block (parity classes)
            w.nR = __builtin_amdgcn_readfirstlane(w.nR);
            w.nS = __builtin_amdgcn_readfirstlane(w.nS);
            w.step_h = __builtin_amdgcn_readfirstlane(w.step_h);
            w.step_w = __builtin_amdgcn_readfirstlane(w.step_w);
            w.nchunks = __builtin_amdgcn_readfirstlane(w.nchunks);
            ci = kb;
            chunk = (w.nR > 0 && w.nS > 0) ? 0 : w.nchunks;
            jr = js = 0;
            sync_offsets();
            return;
        }
        const int per = w.nR * w.nS * w.CB;
        chunk = per > 0 ? kb / per : w.nchunks;
        const int rem = kb - chunk * (per > 0 ? per : 0);
        const int tap = rem / w.CB;
        ci = rem - tap * w.CB;
        jr = w.nS > 0 ? tap / w.nS : 0;
        js = tap - jr * w.nS;
        sync_offsets();
    }
    __device__ __forceinline__ void sync_offsets()
    {
        hoff = jr * w.step_h;
        woff = js * w.step_w;
        tapoff = (hoff * w.IW + woff) * w.IC;
    }
    __device__ __forceinline__ void advance()
    {
        if (!UNIFORM) ci += E::BK;
#pragma unroll
        for (int i = 0; i < WRAPS; ++i) {
            const bool c1 = UNIFORM ? true : ci >= w.CB;
            if (!UNIFORM) ci -= c1 ? w.CB : 0;
            js += c1 ? 1 : 0;
            const bool c2 = js >= w.nS;
            js = c2 ? 0 : js;
            jr += c2 ? 1 : 0;
            const bool c3 = jr >= w.nR;
            jr = c3 ? 0 : jr;
            chunk += c3 ? 1 : 0;
        }
        sync_offsets();
    }
    __device__ __forceinline__ void stage(int k0)
    {
        if (k0 != 0) advance();                               // calls come with k0 = 0, BK, 2BK, ...
        const bool kv = chunk < w.nchunks;
        const int cc = chunk * w.CB + (UNIFORM ? 0 : ci);
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const bool ok = rv_[j] && kv && (unsigned)(bh_[j] + hoff) < (unsigned)w.IH && (unsigned)(bw_[j] + woff) < (unsigned)w.IW;
            off_[j] = ok ? (unsigned)(rowbase_[j] + tapoff + cc) * (unsigned)E::ESZ : OOB;
        }
    }
    __device__ __forceinline__ float4 fetch(int j) const { return buf_load4(rsrc, off_[j]); }
};

// Scalar fallback (IC % 4 != 0: the single-channel first layers, K = 1 logits convs): per-element decode.
template <bool DGRAD>
struct ScalarALoader {
    const float *__restrict__ src;
    ConvGeom g;
    int n_[2], bh_[2], bw_[2];
    bool rv_[2];
    int CB, kbase;
    __device__ __forceinline__ void init(const float *s, const ConvGeom &gg, int m0)
    {
        src = s;
        g = gg;
        CB = chan_block(g.IC);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int m = m0 + (threadIdx.x >> 2) + 64 * j;
            rv_[j] = m < g.Mtot;
            int mm = rv_[j] ? m : 0;
            int ow = mm % g.OW;
            int t = mm / g.OW;
            int oh = t % g.OH;
            n_[j] = t / g.OH;
            bh_[j] = DGRAD ? oh + g.ph : oh * g.sh - g.ph;
            bw_[j] = DGRAD ? ow + g.pw : ow * g.sw - g.pw;
        }
    }
    __device__ __forceinline__ void stage(int k0) { kbase = k0 + 4 * (threadIdx.x & 3); }
    __device__ __forceinline__ bool coords(int j, int r, int s, int &ih, int &iw) const
    {
        if (DGRAD) {
            int th = bh_[j] - r * g.dh, tw = bw_[j] - s * g.dw;
            if (th < 0 || tw < 0) return false;
            ih = th / g.sh;
            iw = tw / g.sw;
            if (ih * g.sh != th || iw * g.sw != tw) return false;
            return ih < g.IH && iw < g.IW;
        }
        ih = bh_[j] + r * g.dh;
        iw = bw_[j] + s * g.dw;
        return (unsigned)ih < (unsigned)g.IH && (unsigned)iw < (unsigned)g.IW;
    }
    __device__ __forceinline__ float4 fetch(int j) const
    {
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (rv_[j]) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kk = kbase + t;
                if (kk < g.Kdim) {
                    const int rscb = g.R * g.S * CB;
                    const int ch = kk / rscb, rem = kk - ch * rscb;
                    const int tap = rem / CB;
                    const int r = tap / g.S, s = tap - r * g.S, c = ch * CB + (rem - tap * CB);
                    int ih, iw;
                    if (coords(j, r, s, ih, iw)) e[t] = src[(((size_t)n_[j] * g.IH + ih) * g.IW + iw) * g.IC + c];
                }
            }
        }
        return make_float4(e[0], e[1], e[2], e[3]);
    }
};

// B operand: dense row-major [Ncols][Kdim] filter bank of one group
template <bool VEC, class E = EngF32, bool UNIFORM = false>     // UNIFORM: Kdim % BK == 0, so the K-tail test is scalar
struct DenseBLoader {
    static constexpr int NS = E::NS;
    const float *__restrict__ w;
    __amdgpu_buffer_rsrc_t rsrc;
    int n0, Ncols, Kdim, kbase;
    unsigned rowoff[NS], off_[NS];
    __device__ __forceinline__ void init(const float *ww, int n0_, int Ncols_, int Kdim_)
    {
        w = ww;
        n0 = n0_;
        Ncols = Ncols_;
        Kdim = Kdim_;
        rsrc = make_rsrc(ww, (size_t)Ncols_ * Kdim_ * E::ESZ);
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int n = n0 + (threadIdx.x / E::QPR) + E::ROWS * j;
            rowoff[j] = n < Ncols ? (unsigned)n * (unsigned)Kdim * (unsigned)E::ESZ : OOB;
            if (UNIFORM && n < Ncols) rowoff[j] += 16u * (threadIdx.x % E::QPR);
        }
    }
    __device__ __forceinline__ void stage(int k0)
    {
        if (UNIFORM) {
            const bool kv = k0 < Kdim;
#pragma unroll
            for (int j = 0; j < NS; ++j) off_[j] = (kv && rowoff[j] != OOB) ? rowoff[j] + (unsigned)k0 * 4u : OOB;
            return;
        }
        kbase = k0 + E::EPL * (threadIdx.x % E::QPR);
#pragma unroll
        for (int j = 0; j < NS; ++j)
            off_[j] = (rowoff[j] != OOB && kbase < Kdim) ? rowoff[j] + (unsigned)kbase * (unsigned)E::ESZ : OOB;
    }
    __device__ __forceinline__ float4 fetch(int j) const
    {
        if (VEC) return buf_load4(rsrc, off_[j]);
        int n = n0 + (threadIdx.x / E::QPR) + E::ROWS * j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n >= Ncols) return v;
        const float *p = w + (size_t)n * Kdim + kbase;
        if (kbase + 0 < Kdim) v.x = p[0];
        if (kbase + 1 < Kdim) v.y = p[1];
        if (kbase + 2 < Kdim) v.z = p[2];
        if (kbase + 3 < Kdim) v.w = p[3];
        return v;
    }
};

// MODE 4 (the Winograd-domain GEMMs: R = S = 1, stride 1, no padding, C % 16 == 0): both operands are dense row-major matrices
// [rows][Kdim].  No tap walk, no coordinates, no divisions: the row part of the address is a loop-invariant vector offset (rows past
// the end of the tile's matrix fall outside the descriptor and read zeros), the K-step is a scalar offset in the load instruction,
// and per K-step a slot costs ONE select (the steps the pipeline prefetches past the end of the reduction).
template <class E>
struct DenseRowLoader {
    static constexpr int NS = E::NS;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned row_[NS], off_[NS];
    int Kdim, soff;
    // base: first row of the tile; rows: valid rows from there on
    __device__ __forceinline__ void init(const float *base, int rows, int Kdim_)
    {
        Kdim = Kdim_;
        rsrc = make_rsrc(base, (size_t)(rows < 0 ? 0 : rows) * Kdim_ * E::ESZ);
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int r = (threadIdx.x / E::QPR) + E::ROWS * j;
            row_[j] = r < rows ? (unsigned)r * (unsigned)Kdim_ * (unsigned)E::ESZ + 16u * (threadIdx.x % E::QPR) : OOB;
        }
        soff = 0;
    }
    __device__ __forceinline__ void stage(int k0)
    {
        const bool kv = k0 < Kdim;                     // wave-uniform
        soff = kv ? k0 * E::ESZ : 0;
#pragma unroll
        for (int j = 0; j < NS; ++j) off_[j] = kv ? row_[j] : OOB;
    }
    __device__ __forceinline__ float4 fetch(int j) const { return buf_load4s(rsrc, off_[j], soff); }
};

template <int BN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[GemmCfg<BN>::TM][GemmCfg<BN>::TN])   // TM/TN do not depend on the engine
{
#pragma unroll
    for (int a = 0; a < GemmCfg<BN>::TM; ++a)
#pragma unroll
        for (int b = 0; b < GemmCfg<BN>::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
}

// ------------------------------------------------------------------------------------------ fprop / dgrad (stride 1)
// MODE 0: scalar loads, 1: vector loads with CB < 16, 2: vector loads with CB >= 16,
//      3: vector loads with C % 16 == 0 on the f32 engine (wave-uniform tap walk)
template <class E, int BN, bool DGRAD, int MODE>
__device__ __forceinline__ void conv_nt_body(const float *__restrict__ act, const float *__restrict__ wgt,
                                             const float *__restrict__ bias, float *__restrict__ out, const ConvGeom &g,
                                             const WalkGeom &wg, size_t act_group_stride, size_t w_group_stride, float *smem,
                                             const NtEpi &epi)
{
    using Cfg = GemmCfg<BN, E>;
    const int grp = blockIdx.y;
    const int tile = xcd_remap(blockIdx.x, g.nMt * g.nNt);
    const int mt = tile / g.nNt, nt = tile - mt * g.nNt;   // N tiles of one M tile are neighbours (share A in L2)
    const int m0 = mt * GEMM_BM, n0 = nt * BN;
    // operand pointers advance in elements of E::ESZ bytes (bf16 operands travel as float* in disguise)
    auto at = [](const float *p, size_t elems) { return reinterpret_cast<const float *>(reinterpret_cast<const char *>(p) + elems * E::ESZ); };
    const float *abase = at(act, (size_t)grp * act_group_stride);

    f32x16 acc[Cfg::TM][Cfg::TN];
    const int nk = (g.Kdim + E::BK - 1) / E::BK;
    if constexpr (MODE == 4) {
        static_assert(E::BK == 16 && !DGRAD, "dense mode: f32 / split engine forward GEMMs");
        DenseRowLoader<E> la, lb;
        la.init(at(abase, (size_t)m0 * g.Kdim), g.Mtot - m0, g.Kdim);
        lb.init(at(wgt, (size_t)grp * w_group_stride + (size_t)n0 * g.Kdim), g.Ncols - n0, g.Kdim);
        if constexpr (!E::SPLIT && !E::BF16) {
            nt_gemm_run<E, BN, true>(la, lb, nk, smem, acc);          // accumulators zeroed by the first MFMA (C = 0)
        } else {
            zero_acc<BN>(acc);
            nt_gemm_run<E, BN>(la, lb, nk, smem, acc);
        }
    } else {
    zero_acc<BN>(acc);
    constexpr bool UNI = MODE == 3;
    static_assert(!UNI || E::BK == 16, "uniform walk needs CB == BK");
    DenseBLoader<(MODE > 0), E, UNI> lb;
    lb.init(at(wgt, (size_t)grp * w_group_stride), n0, g.Ncols, g.Kdim);
    if constexpr (MODE == 0) {
        static_assert(!E::BF16 && !E::SPLIT, "scalar path is f32 only");
        ScalarALoader<DGRAD> la;
        la.init(abase, g, m0);
        nt_gemm_mainloop<BN>(la, lb, nk, smem, acc);
    } else {
        // The buffer descriptor is based at the first image this tile touches, so 32-bit offsets only have to span
        // the few images of one tile (tensors beyond 4 GB stay on the vector path).
        TapWalker<(MODE >= 2 ? E::BK / 16 : E::BK / 4), E, UNI> la;
        const int img0 = m0 / (g.OH * g.OW);
        const size_t img_elems = (size_t)g.IH * g.IW * g.IC;
        la.start(at(abase, img0 * img_elems), act_group_stride - img0 * img_elems, wg);
#pragma unroll
        for (int j = 0; j < E::NS; ++j) {
            const int m = m0 + (threadIdx.x / E::QPR) + E::ROWS * j;
            const bool v = m < g.Mtot;
            const int mm = v ? m : m0;
            const int ow = mm % g.OW, t = mm / g.OW;
            const int oh = t % g.OH, n = t / g.OH - img0;
            la.set_row(j, v, n, DGRAD ? oh + g.ph : oh * g.sh - g.ph, DGRAD ? ow + g.pw : ow * g.sw - g.pw);
        }
        nt_gemm_run<E, BN>(la, lb, nk, smem, acc);
    }
    }

    float *o = out + (size_t)grp * g.Mtot * g.Ncols;
    // fused activation: the padded flag of each of the tile's 128 rows is looked up once (LDS is free after the main
    // loop's last barrier)
    unsigned char *padf = reinterpret_cast<unsigned char *>(smem);
    const bool fused = !DGRAD && epi.act == 2;
    const bool gated = DGRAD && epi.act == 3;
    const float *gt = gated ? epi.gate + (size_t)grp * g.Mtot * g.Ncols : nullptr;
    if (fused || gated) {
        if (threadIdx.x < GEMM_BM) {
            const int m = m0 + threadIdx.x;
            unsigned char pd = 0;
            if (epi.mask != nullptr && m < g.Mtot) {
                const int64_t mg = (int64_t)grp * g.Mtot + m;
                pd = epi.mask[(mg / epi.d1) * epi.d2 + (mg % epi.d3)] != 0;
            }
            padf[threadIdx.x] = pd;
        }
        __syncthreads();
    }
    if (!(fused || gated)) {
        // Plain store (every Winograd GEMM, every layer without a fused activation) through a buffer descriptor that ends
        // behind the tile's last valid row: rows >= Mtot and columns >= Ncols fall outside it, so there are no compares and no
        // 64-bit address arithmetic per element - one 32-bit add of a scalar (row step x pitch) to the lane's offset.  The whole
        // offset is in the VECTOR operand on purpose: the scalar offset of a buffer instruction is not bounds-checked.  Vector
        // instructions come out of the f32 matrix pipe's time; with reductions only C deep the epilogue was a quarter of a
        // block's vector instructions (4.0 per MFMA at C = 128).
        const int rows_left = g.Mtot - m0 < GEMM_BM ? g.Mtot - m0 : GEMM_BM;
        const unsigned pitch = (unsigned)g.Ncols * 4u;
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(o + (size_t)m0 * g.Ncols, (size_t)rows_left * pitch);
        const int lane_row = acc_row<BN>(0, 0);                      // wave and lane part of the row index
#pragma unroll
        for (int tn = 0; tn < Cfg::TN; ++tn) {
            const int n = n0 + acc_col<BN>(tn);
            const bool colok = n < g.Ncols;
            const float bv = (bias && colok) ? bias[(size_t)grp * g.Ncols + n] : 0.f;
            // an invalid column starts beyond every valid offset and stays there: 0x80000000 + at most 128 rows x pitch
            const unsigned voff = colok ? (unsigned)lane_row * pitch + (unsigned)n * 4u : 0x80000000u;
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned rstep = (unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * pitch;      // uniform
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[tm][tn][r] + bv), ro, (int)(voff + rstep), 0, 0);
                }
        }
        return;
    }
#pragma unroll
    for (int tn = 0; tn < Cfg::TN; ++tn) {
        const int n = n0 + acc_col<BN>(tn);
        if (n >= g.Ncols) continue;
        const float bv = bias ? bias[(size_t)grp * g.Ncols + n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row<BN>(tm, r);
                const int m = m0 + row;
                float v = acc[tm][tn][r] + bv;
                if (fused) {
                    v = v > 0.f ? v : epi.slope * v;
                    v = padf[row] ? 0.f : v;
                }
                if (gated && m < g.Mtot) {
                    v = gt[(size_t)m * g.Ncols + n] > 0.f ? v : epi.slope * v;
                    v = padf[row] ? 0.f : v;
                }
                if (m < g.Mtot) o[(size_t)m * g.Ncols + n] = v;
            }
    }
}

template <int BN, bool DGRAD, int MODE>
__global__ __launch_bounds__(256, MQ_NT_OCC) void conv_nt_kernel(const float *__restrict__ act, const float *__restrict__ wgt,
                                                      const float *__restrict__ bias, float *__restrict__ out,
                                                      ConvGeom g, WalkGeom wg, size_t act_group_stride, size_t w_group_stride, NtEpi epi)
{
    __shared__ __attribute__((aligned(16))) float smem[GemmCfg<BN, EngF32>::LDS_FLOATS];
    conv_nt_body<EngF32, BN, DGRAD, MODE>(act, wgt, bias, out, g, wg, act_group_stride, w_group_stride, smem, epi);
}
// f32 via three-term bf16 split (gemm_core.h: EngSplit); vector path only
template <int BN, bool DGRAD, int MODE>
__global__ __launch_bounds__(256, 3) void conv_nt_x3_kernel(const float *__restrict__ act, const float *__restrict__ wgt,
                                                         const float *__restrict__ bias, float *__restrict__ out,
                                                         ConvGeom g, WalkGeom wg, size_t act_group_stride,
                                                         size_t w_group_stride, NtEpi epi)
{
    __shared__ __attribute__((aligned(16))) float smem[GemmCfg<BN, EngSplit>::LDS_FLOATS];
    conv_nt_body<EngSplit, BN, DGRAD, MODE>(act, wgt, bias, out, g, wg, act_group_stride, w_group_stride, smem, epi);
}
// bf16 operands already in HBM (x_h, w_h from mq_cast_bf16), f32 accumulate, f32 output; C % 16 == 0 only
template <int BN, bool DGRAD>
__global__ __launch_bounds__(256, (BN == 128 ? 3 : 4)) void conv_nt_h_kernel(const float *__restrict__ act, const float *__restrict__ wgt,
                                                        const float *__restrict__ bias, float *__restrict__ out, ConvGeom g,
                                                        WalkGeom wg, size_t act_group_stride, size_t w_group_stride, NtEpi epi)
{
    __shared__ __attribute__((aligned(16))) float smem[GemmCfg<BN, EngBF16H>::LDS_FLOATS];
    conv_nt_body<EngBF16H, BN, DGRAD, 2>(act, wgt, bias, out, g, wg, act_group_stride, w_group_stride, smem, epi);
}
// bf16 operands (rounded on the way into LDS), f32 accumulate and f32 tensors in HBM; vector path only
template <int BN, bool DGRAD, int MODE>
__global__ __launch_bounds__(256, (BN == 128 ? 3 : 4)) void conv_nt_bf16_kernel(const float *__restrict__ act, const float *__restrict__ wgt,
                                                           const float *__restrict__ bias, float *__restrict__ out,
                                                           ConvGeom g, WalkGeom wg, size_t act_group_stride,
                                                           size_t w_group_stride, NtEpi epi)
{
    __shared__ __attribute__((aligned(16))) float smem[GemmCfg<BN, EngBF16>::LDS_FLOATS];
    conv_nt_body<EngBF16, BN, DGRAD, MODE>(act, wgt, bias, out, g, wg, act_group_stride, w_group_stride, smem, epi);
}


// ------------------------------------------------------------------------------------------ dense 256 x 128 tile (round 6)
// The Winograd-domain products (MODE 4) with >= 128 output columns on a 256 x 128 block tile: four waves as 2 x 2, wave tile 128 x 64
// (eight 32 x 32 accumulators), two blocks per CU.  What changed against nt_gemm_mainloop (tools/microbench/f32_gemm_lab.hip measured
// each step, profiles/r6_f32_gemm_lab.txt): LDS rows of 16 floats without padding, the 16-byte chunk c of row r stored at chunk
// c ^ ((r >> 2) & 3) - fragment reads (ds_read_b128: four k's per lane and read, half the LDS instructions) and staging writes
// (ds_write_b128) are both conflict-free; and the K-step is software-pipelined: two groups of 32 MFMAs (the lane's two float4
// fragments), the fragments of the NEXT group and the LDS writes of the NEXT tile issued between the MFMAs of this one, the K-step's
// one barrier between its two groups, so no MFMA waits for an LDS read issued right in front of it.  Steady state, same box:
// 512 -> 512  140.4-141.5 TF/s against 135.6-138.7, 256 -> 256  132.6 against 127.9 (+2.0 ... +3.7 %).
// Accumulation order per accumulator: k = (j, 4 + j), j = 0 .. 3, then (8 + j, 12 + j) of every 16 - another order than the
// 128-row kernel's, the same exact f32 products (the convolutions are held to float64 at 2e-5, not to a bit pattern).
struct DenseRow4 {             // rows of a dense [rows][Kdim] f32 matrix: slot j = tile rows 64 j + (tid >> 2), the thread's 16-byte k-quad
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned row_[4], off_[4];
    int Kdim, soff;
    __device__ __forceinline__ void init(const float *base, int rows, int Kdim_)
    {
        Kdim = Kdim_;
        rsrc = make_rsrc(base, (size_t)(rows < 0 ? 0 : rows) * Kdim_ * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (threadIdx.x >> 2) + 64 * j;
            row_[j] = r < rows ? (unsigned)r * (unsigned)Kdim_ * 4u + 16u * (threadIdx.x & 3) : OOB;
        }
        soff = 0;
    }
    __device__ __forceinline__ void stage(int k0)
    {
        const bool kv = k0 < Kdim;                     // wave-uniform
        soff = kv ? k0 * 4 : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) off_[j] = kv ? row_[j] : OOB;
    }
    __device__ __forceinline__ float4 fetch(int j) const { return buf_load4s(rsrc, off_[j], soff); }
};

__global__ __launch_bounds__(256, 2) void conv_nt_dense256_kernel(const float *__restrict__ act, const float *__restrict__ wgt,
                                                                 const float *__restrict__ bias, float *__restrict__ out, ConvGeom g,
                                                                 size_t act_group_stride, size_t w_group_stride)
{
    constexpr int BM = 256, BN = 128, TM = 4, TN = 2, AS = 4, BS = 2;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * 16];
    const int grp = blockIdx.y;
    const int tile = xcd_remap(blockIdx.x, g.nMt * g.nNt);
    const int mt = tile / g.nNt, nt = tile - mt * g.nNt;
    const int m0 = mt * BM, n0 = nt * BN;
    DenseRow4 la, lb;
    la.init(act + (size_t)grp * act_group_stride + (size_t)m0 * g.Kdim, g.Mtot - m0, g.Kdim);
    lb.init(wgt + (size_t)grp * w_group_stride + (size_t)n0 * g.Kdim, g.Ncols - n0, g.Kdim);
    const int nk = g.Kdim / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    float *As = smem, *Bs = smem + 2 * BM * 16;
    const int ld_row = tid >> 2, ld_kq = tid & 3;
    const int st_off = ld_row * 16 + 4 * (ld_kq ^ ((ld_row >> 2) & 3));            // rows ld_row + 64 j share (row >> 2) & 3
    const int sw = (li >> 2) & 3;
    const int fo[2] = {4 * ((0 + lh) ^ sw), 4 * ((2 + lh) ^ sw)};
    const float *Ar = As + (wm * 128 + li) * 16, *Br = Bs + (wn * 64 + li) * 16;
    float4 ra[AS], rb[BS];
    float4 fa[2][TM], fb[2][TN];
    f32x16 acc[TM][TN];
    auto frag_a = [&](int buf, int s_, int t) { fa[s_][t] = *reinterpret_cast<const float4 *>(Ar + buf * BM * 16 + t * 32 * 16 + fo[s_]); };
    auto frag_b = [&](int buf, int s_, int t) { fb[s_][t] = *reinterpret_cast<const float4 *>(Br + buf * BN * 16 + t * 32 * 16 + fo[s_]); };
    auto put_a = [&](int buf, int j) { *reinterpret_cast<float4 *>(&As[buf * BM * 16 + st_off + 64 * 16 * j]) = ra[j]; };
    auto put_b = [&](int buf, int j) { *reinterpret_cast<float4 *>(&Bs[buf * BN * 16 + st_off + 64 * 16 * j]) = rb[j]; };
    // prologue: tile 0 -> LDS buffer 0, tile 1 -> staging registers, fragments (0, 0)
    la.stage(0);
    lb.stage(0);
#pragma unroll
    for (int j = 0; j < AS; ++j) ra[j] = la.fetch(j);
#pragma unroll
    for (int j = 0; j < BS; ++j) rb[j] = lb.fetch(j);
#pragma unroll
    for (int j = 0; j < AS; ++j) put_a(0, j);
#pragma unroll
    for (int j = 0; j < BS; ++j) put_b(0, j);
    la.stage(16);
    lb.stage(16);
#pragma unroll
    for (int j = 0; j < AS; ++j) ra[j] = la.fetch(j);
#pragma unroll
    for (int j = 0; j < BS; ++j) rb[j] = lb.fetch(j);
    la.stage(32);
    lb.stage(32);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TM; ++t) frag_a(0, 0, t);
#pragma unroll
    for (int t = 0; t < TN; ++t) frag_b(0, 0, t);

    // 8 MFMAs of one k pair (element e of fragment s) over the wave tile
    auto mfma8 = [&](int s_, int e, bool zero_first) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const float av = e == 0 ? fa[s_][tm].x : e == 1 ? fa[s_][tm].y : e == 2 ? fa[s_][tm].z : fa[s_][tm].w;
                const float bv = e == 0 ? fb[s_][tn].x : e == 1 ? fb[s_][tn].y : e == 2 ? fb[s_][tn].z : fb[s_][tn].w;
                if (zero_first) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, zero, 0, 0, 0);
                } else {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tm][tn], 0, 0, 0);
                }
            }
    };
    auto k_step = [&](int kt, auto cur_tag, auto first_tag) {
        constexpr int cur = decltype(cur_tag)::value;
        constexpr bool first = decltype(first_tag)::value;
        // ---- group 0: MFMAs on (kt, 0); fragments (kt, 1) early, the LDS writes of tile kt + 1 late
        mfma8(0, 0, first);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TM; ++t) frag_a(cur, 1, t);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(0, 1, false);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TN; ++t) frag_b(cur, 1, t);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(0, 2, false);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < AS; ++j) put_a(cur ^ 1, j);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(0, 3, false);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < BS; ++j) put_b(cur ^ 1, j);
        __syncthreads();                                   // tile kt + 1 is in LDS; every read of tile kt has been issued and awaited
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 1: MFMAs on (kt, 1); global loads of tile kt + 2, fragments (kt + 1, 0) a whole chunk of MFMAs ahead of their use
        mfma8(1, 0, false);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < AS; ++j) ra[j] = la.fetch(j);
#pragma unroll
        for (int t = 0; t < TM; ++t) frag_a(cur ^ 1, 0, t);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(1, 1, false);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < BS; ++j) rb[j] = lb.fetch(j);
#pragma unroll
        for (int t = 0; t < TN; ++t) frag_b(cur ^ 1, 0, t);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(1, 2, false);
        __builtin_amdgcn_sched_barrier(0);
        la.stage((kt + 3) * 16);
        lb.stage((kt + 3) * 16);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(1, 3, false);
        __builtin_amdgcn_sched_barrier(0);
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    k_step(0, c0{}, std::true_type{});
    int kt = 1;
    for (; kt + 1 < nk; kt += 2) {
        k_step(kt, c1{}, std::false_type{});
        k_step(kt + 1, c0{}, std::false_type{});
    }
    if (kt < nk) k_step(kt, c1{}, std::false_type{});

    // plain store (+ bias) through a buffer descriptor that ends behind the tile's last valid row, as conv_nt_body's
    float *o = out + (size_t)grp * g.Mtot * g.Ncols;
    const int rows_left = g.Mtot - m0 < BM ? g.Mtot - m0 : BM;
    const unsigned pitch = (unsigned)g.Ncols * 4u;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(o + (size_t)m0 * g.Ncols, (size_t)rows_left * pitch);
    const int lane_row = wm * 128 + 4 * lh;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wn * 64 + tn * 32 + li;
        const bool colok = n < g.Ncols;
        const float bv = (bias && colok) ? bias[(size_t)grp * g.Ncols + n] : 0.f;
        const unsigned voff = colok ? (unsigned)lane_row * pitch + (unsigned)n * 4u : 0x80000000u;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned rstep = (unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * pitch;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[tm][tn][r] + bv), ro, (int)(voff + rstep), 0, 0);
            }
    }
}

// ------------------------------------------------------------------------------------------ dgrad by stride parity
// For stride > 1 only the taps with (ih + ph - r*dh) % sh == 0 contribute to input pixel ih: 1/sh of
// the rows of the filter (and 1/sw of its columns).  The input pixels are therefore partitioned into
// sh*sw parity classes; every M tile lies inside one class and its GEMM runs over that class's taps
// only, so no MFMA is spent on structural zeros (4x fewer for the (2,2) discriminator layers).
struct ParityInfo {
    int ncls;
    int a[4], b[4];        // class: ih % sh == a, iw % sw == b
    int Ha[4], Wb[4];      // pixels of the class per image: Ha x Wb
    int tile0[5];          // prefix sum of M tiles per class
    int nr[4], ns[4];      // valid taps per class; they form arithmetic progressions r = r0 + j*rstep
    int r0[4], s0[4], rstep[4], sstep[4];
};

template <int WRAPS, class E = EngF32, bool UNIFORM = false>
struct ParityBLoader {     // filter bank [Ncols][R*S*IC] (channel-blocked), columns of the class's taps only
    static constexpr int NS = E::NS;
    __amdgpu_buffer_rsrc_t rsrc;
    int S, RS, CB, nchunks, nr, ns, r0, s0, rstep, sstep;
    unsigned rowoff[NS], off_[NS];
    int ci, js, jr, chunk;
    __device__ __forceinline__ void init(const float *w, int n0, int Ncols, int IC, int S_, int KdimFull, int nr_, int ns_,
                                         int r0_, int s0_, int rstep_, int sstep_)
    {
        S = S_; CB = chan_block(IC); nchunks = IC / CB; RS = KdimFull / IC;
        nr = nr_; ns = ns_; r0 = r0_; s0 = s0_; rstep = rstep_; sstep = sstep_;
        rsrc = make_rsrc(w, (size_t)Ncols * KdimFull * E::ESZ);
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int n = n0 + (threadIdx.x / E::QPR) + E::ROWS * j;
            rowoff[j] = n < Ncols ? (unsigned)n * (unsigned)KdimFull * (unsigned)E::ESZ : OOB;
            if (UNIFORM && n < Ncols) rowoff[j] += 16u * (threadIdx.x % E::QPR);
        }
        const int kb = E::EPL * (threadIdx.x % E::QPR);
        if (UNIFORM) {
            nr = __builtin_amdgcn_readfirstlane(nr); ns = __builtin_amdgcn_readfirstlane(ns);
            r0 = __builtin_amdgcn_readfirstlane(r0); s0 = __builtin_amdgcn_readfirstlane(s0);
            rstep = __builtin_amdgcn_readfirstlane(rstep); sstep = __builtin_amdgcn_readfirstlane(sstep);
            ci = 0;                                          // the thread's channel quad is folded into rowoff
            chunk = (nr > 0 && ns > 0) ? 0 : nchunks;
            jr = js = 0;
            return;
        }
        const int per = nr * ns * CB;
        chunk = per > 0 ? kb / per : nchunks;
        const int rem = kb - chunk * (per > 0 ? per : 0);
        const int tap = rem / CB;
        ci = rem - tap * CB;
        jr = ns > 0 ? tap / ns : 0;
        js = tap - jr * ns;
    }
    __device__ __forceinline__ void stage(int k0)
    {
        if (k0 != 0) {
            if (!UNIFORM) ci += E::BK;
#pragma unroll
            for (int i = 0; i < WRAPS; ++i) {
                const bool c1 = UNIFORM ? true : ci >= CB;
                if (!UNIFORM) ci -= c1 ? CB : 0;
                js += c1 ? 1 : 0;
                const bool c2 = js >= ns;
                js = c2 ? 0 : js;
                jr += c2 ? 1 : 0;
                const bool c3 = jr >= nr;
                jr = c3 ? 0 : jr;
                chunk += c3 ? 1 : 0;
            }
        }
        const int col = (chunk * RS + (r0 + jr * rstep) * S + s0 + js * sstep) * CB + ci;
        const bool kv = chunk < nchunks;
#pragma unroll
        for (int j = 0; j < NS; ++j) off_[j] = (kv && rowoff[j] != OOB) ? rowoff[j] + (unsigned)col * (unsigned)E::ESZ : OOB;
    }
    __device__ __forceinline__ float4 fetch(int j) const { return buf_load4(rsrc, off_[j]); }
};

// WRAPS_ = 0 selects the wave-uniform walk (C % 16 == 0 on the f32 engine, one carry per K-step)
template <class E, int BN, int WRAPS_>
__device__ __forceinline__ void conv_dgrad_parity_body(const float *__restrict__ dy, const float *__restrict__ wgt,
                                                       float *__restrict__ dx, const ConvGeom &g, const ParityInfo &pi,
                                                       size_t act_group_stride, size_t w_group_stride, int ng, float *smem,
                                                       const NtEpi &epi)
{
    using Cfg = GemmCfg<BN, E>;
    constexpr bool UNI = WRAPS_ == 0;
    constexpr int WRAPS = UNI ? 1 : WRAPS_;
    static_assert(!UNI || E::BK == 16, "uniform walk needs CB == BK");
    const int grp = blockIdx.y;
    const int total_mt = pi.tile0[pi.ncls];
    // classes have different tap counts (= block run times): keep dispatch order so every XCD gets the same mix
    const int tile = xcd_pair_remap(blockIdx.x, total_mt * g.nNt, g.nNt);
    const int mtg = tile / g.nNt, nt = tile - mtg * g.nNt;
    int cls = 0;
#pragma unroll
    for (int c = 1; c < 4; ++c)
        if (c < pi.ncls && mtg >= pi.tile0[c]) cls = c;
    const int m0 = (mtg - pi.tile0[cls]) * GEMM_BM, n0 = nt * BN;
    const int Ha = pi.Ha[cls], Wb = pi.Wb[cls], a = pi.a[cls], b = pi.b[cls], nr = pi.nr[cls], ns = pi.ns[cls];
    const int r0 = pi.r0[cls], s0 = pi.s0[cls], rstep = pi.rstep[cls], sstep = pi.sstep[cls];
    const int Mc = ng * Ha * Wb;
    const int Kdim = nr * ns * g.IC;
    const int shift_h = g.sh == 2 ? 1 : 0, shift_w = g.sw == 2 ? 1 : 0;

    WalkGeom wg;
    wg.IH = g.IH; wg.IW = g.IW; wg.IC = g.IC; wg.nR = nr; wg.nS = ns;
    wg.step_h = -((rstep * g.dh) >> shift_h);        // exact: the class's taps keep (ih + ph - r*dh) divisible by sh
    wg.step_w = -((sstep * g.dw) >> shift_w);
    wg.CB = chan_block(g.IC); wg.nchunks = g.IC / wg.CB;
    TapWalker<WRAPS, E, UNI> la;
    const int img0 = (Ha * Wb) > 0 ? m0 / (Ha * Wb) : 0;
    const size_t img_elems = (size_t)g.IH * g.IW * g.IC;
    auto at = [](const float *p, size_t elems) { return reinterpret_cast<const float *>(reinterpret_cast<const char *>(p) + elems * E::ESZ); };
    la.start(at(dy, (size_t)grp * act_group_stride + img0 * img_elems), act_group_stride - img0 * img_elems, wg);
#pragma unroll
    for (int j = 0; j < E::NS; ++j) {
        const int m = m0 + (threadIdx.x / E::QPR) + E::ROWS * j;
        const bool v = m < Mc;
        const int mm = v ? m : m0;
        const int iwq = mm % Wb, t = mm / Wb;
        const int ihq = t % Ha, n = t / Ha - img0;
        la.set_row(j, v, n, (ihq * g.sh + a + g.ph - r0 * g.dh) >> shift_h, (iwq * g.sw + b + g.pw - s0 * g.dw) >> shift_w);
    }
    ParityBLoader<WRAPS, E, UNI> lb;
    lb.init(at(wgt, (size_t)grp * w_group_stride), n0, g.Ncols, g.IC, g.S, g.Kdim, nr, ns, r0, s0, rstep, sstep);

    f32x16 acc[Cfg::TM][Cfg::TN];
    zero_acc<BN>(acc);
    nt_gemm_run<E, BN>(la, lb, (Kdim + E::BK - 1) / E::BK, smem, acc);

    const size_t gbase = (size_t)grp * ng * g.OH * g.OW;             // first pixel of the group
    float *o = dx + gbase * g.Ncols;
    const bool gated = epi.act == 3;                                 // LeakyReLU' / mask of the layer that produced the input
    const float *gt = gated ? epi.gate + gbase * g.Ncols : nullptr;
    auto pixel_of = [&](int m) -> size_t {                           // class row -> pixel of the group
        const int iwq = m % Wb;
        const int t = m / Wb;
        const int ihq = t % Ha, n = t / Ha;
        return ((size_t)n * g.OH + ihq * g.sh + a) * g.OW + iwq * g.sw + b;
    };
    // Pixel index and padded flag of the tile's 128 rows, worked out ONCE per row by 128 threads (LDS is free after the main
    // loop's last barrier) instead of once per accumulator register: the two divisions and two remainders of pixel_of() were
    // ~60 vector instructions, 32 times per lane - 1.9 of the 4.2 vector instructions per MFMA this kernel issued, each of them
    // 4 of the 64 cycles of an MFMA on the f32 pipe.  A group has fewer than 2^32 pixels (host check).
    unsigned char *padf = reinterpret_cast<unsigned char *>(smem);
    unsigned *rowpix = reinterpret_cast<unsigned *>(smem) + GEMM_BM / 4;
    if (threadIdx.x < GEMM_BM) {
        const int m = m0 + threadIdx.x;
        unsigned char pd = 0;
        unsigned pix = 0xFFFFFFFFu;
        if (m < Mc) {
            pix = (unsigned)pixel_of(m);
            if (gated && epi.mask != nullptr) {
                const int64_t mg = (int64_t)(gbase + pix);
                pd = epi.mask[(mg / epi.d1) * epi.d2 + (mg % epi.d3)] != 0;
            }
        }
        padf[threadIdx.x] = pd;
        rowpix[threadIdx.x] = pix;
    }
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row<BN>(tm, r);
            const unsigned pix = rowpix[row];
            if (pix == 0xFFFFFFFFu) continue;
            const size_t base = (size_t)pix * g.Ncols;
            const bool pd = gated && padf[row] != 0;
#pragma unroll
            for (int tn = 0; tn < Cfg::TN; ++tn) {
                const int col = n0 + acc_col<BN>(tn);
                if (col < g.Ncols) {
                    float v = acc[tm][tn][r];
                    if (gated) {
                        v = gt[base + col] > 0.f ? v : epi.slope * v;
                        v = pd ? 0.f : v;
                    }
                    o[base + col] = v;
                }
            }
        }
}

template <int BN, int WRAPS>
__global__ __launch_bounds__(256, MQ_NT_OCC) void conv_dgrad_parity_kernel(const float *__restrict__ dy, const float *__restrict__ wgt,
                                                                float *__restrict__ dx, ConvGeom g, ParityInfo pi,
                                                                size_t act_group_stride, size_t w_group_stride, int ng, NtEpi epi)
{
    __shared__ __attribute__((aligned(16))) float smem[GemmCfg<BN, EngF32>::LDS_FLOATS];
    conv_dgrad_parity_body<EngF32, BN, WRAPS>(dy, wgt, dx, g, pi, act_group_stride, w_group_stride, ng, smem, epi);
}
template <int BN, int WRAPS>
__global__ __launch_bounds__(256, 3) void conv_dgrad_parity_x3_kernel(const float *__restrict__ dy, const float *__restrict__ wgt,
                                                                   float *__restrict__ dx, ConvGeom g, ParityInfo pi,
                                                                   size_t act_group_stride, size_t w_group_stride, int ng, NtEpi epi)
{
    __shared__ __attribute__((aligned(16))) float smem[GemmCfg<BN, EngSplit>::LDS_FLOATS];
    conv_dgrad_parity_body<EngSplit, BN, WRAPS>(dy, wgt, dx, g, pi, act_group_stride, w_group_stride, ng, smem, epi);
}
template <int BN>
__global__ __launch_bounds__(256, (BN == 128 ? 3 : 4)) void conv_dgrad_parity_h_kernel(const float *__restrict__ dy,
                                                                  const float *__restrict__ wgt, float *__restrict__ dx,
                                                                  ConvGeom g, ParityInfo pi, size_t act_group_stride,
                                                                  size_t w_group_stride, int ng, NtEpi epi)
{
    __shared__ __attribute__((aligned(16))) float smem[GemmCfg<BN, EngBF16H>::LDS_FLOATS];
    conv_dgrad_parity_body<EngBF16H, BN, 2>(dy, wgt, dx, g, pi, act_group_stride, w_group_stride, ng, smem, epi);
}
template <int BN, int WRAPS>
__global__ __launch_bounds__(256, (BN == 128 ? 3 : 4)) void conv_dgrad_parity_bf16_kernel(const float *__restrict__ dy,
                                                                     const float *__restrict__ wgt, float *__restrict__ dx,
                                                                     ConvGeom g, ParityInfo pi, size_t act_group_stride,
                                                                     size_t w_group_stride, int ng, NtEpi epi)
{
    __shared__ __attribute__((aligned(16))) float smem[GemmCfg<BN, EngBF16>::LDS_FLOATS];
    conv_dgrad_parity_body<EngBF16, BN, WRAPS>(dy, wgt, dx, g, pi, act_group_stride, w_group_stride, ng, smem, epi);
}

// ------------------------------------------------------------------------------------------ wgrad
// dw[k][kk] = sum_m dy[m][k] * xg[m][kk], kk = (r, s, c), reduction over the pixels m of one group.
// Both operands are "M/N contiguous" in memory, so the LDS image is [pixel][row] and the MFMA operands
// are read with ds_read_b64 as (row 2i, row 2i+1): tile t of a wave covers rows {2i + t}.
constexpr int WG_BK = 16;          // pixels per step (f32 engine; the bf16 engine takes 32)

struct WgradGeom {
    int OH, OW;            // P, Q
    int IH, IW, IC;        // H, W, C
    int R, S, sh, sw, ph, pw, dh, dw;
    int Mtot;              // pixels per group
    int K, RSC;
    int nKt, nCt;          // tiles along K (rows) and RSC (cols)
    int bmk, bnc;          // block tile
    int splits, chunk;     // split-K: pixels per split (multiple of WG_BK)
};

// Block tile BMK (output channels k) x BNC (filter columns kk), 4 waves of 64 x 64 each:
//   <128, 128> general;  <64, 256> when K <= 64 (refiner top level, logits convs) so that rows are not wasted.
// f32 engine only (r4), for the shapes those two pad by 4x and more:
//   <256, 32>  filter rows of at most 32 columns (the 1x1 GEMMs behind the C = 1 first layers: 25 taps, 24 / 12 band taps):
//              4 waves of 64 x 32, one column tile per wave;
//   <64, 64>   K <= 64 and at most 64 columns (the 64 -> 64 level in the Winograd domain): every wave holds the whole 64 x 64
//              tile and takes a quarter of each step's pixels; the four partial tiles are added in a fixed order at the end.
// MODE 0: scalar loads (C or K not a multiple of 4); 1: vector loads, pixel coordinates by division;
//      2: vector loads, pixel coordinates walked incrementally per thread (OW >= 16: at most BKP/16 row wraps per step);
//      3: OW >= BKP - the first pixel of the step is walked on the scalar unit and every thread adds its constant
//         pixel delta (at most one row wrap inside a step): half the vector work of mode 2.
//
// bf16 engine (BF16 = true): 32 pixels per step; the tiles are rounded to bf16 on their way into LDS and kept
// [pixel][row] with 2*rows + 64 bytes per pixel (64 mod 256: the four pixel rows of a transposed read fall on
// disjoint bank quarters).  The MFMA needs 8 consecutive pixels of one row per lane, i.e. the transpose of the
// image: ds_read_b64_tr_b16 delivers it (lane 4q+p of a 16-lane group addresses pixel p0+q, rows 4p..4p+3 and
// lane i receives row i of the 4 pixels), two reads per 32x32x16 operand.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// ENG: 0 = f32 MFMA, 1 = bf16 operands, 2 = f32 by three-term bf16 split (three LDS images per operand, 16 pixels/step)
template <int BMK, int BNC, int ENG>
struct WgradCfg {
    static constexpr bool BF16 = ENG != 0;                           // bf16 LDS images, transposed reads
    static constexpr int NIMG = ENG == 2 ? 3 : 1;
    static constexpr int BKP = ENG == 1 ? 32 : WG_BK;                // pixels per step
    static constexpr int LDA = BF16 ? BMK / 2 + 16 : BMK + 4;        // row pitch in 4-byte units
    static constexpr int LDB = BF16 ? BNC / 2 + 16 : BNC + 4;        // (f32: +4 keeps float4 stores aligned, shifts banks)
    static constexpr int LDS_FLOATS = NIMG * 2 * BKP * (LDA + LDB);
    static constexpr bool WSPLIT = BMK == 64 && BNC == 64;           // waves split the pixels of a step instead of the tile
    static constexpr int NTN = BNC >= 64 ? 2 : 1;                    // 32-column MFMA tiles per wave
    static_assert(ENG == 0 || (BNC >= 128), "the narrow tiles exist for the f32 engine only");
    static_assert(!WSPLIT || LDS_FLOATS >= 64 * 64, "the partial-tile reduction reuses the operand buffers");
};

template <int BMK, int BNC, int MODE, int ENG>
__device__ __forceinline__ void conv_wgrad_body(const float *__restrict__ x, const float *__restrict__ dy,
                                                float *__restrict__ out, float *__restrict__ bias_out, const WgradGeom &g,
                                                int groups, float *smem)
{
    using WC = WgradCfg<BMK, BNC, ENG>;
    constexpr bool BF16 = WC::BF16, X3 = ENG == 2;
    constexpr int IMG_A = 2 * WC::BKP * WC::LDA, IMG_B = 2 * WC::BKP * WC::LDB;     // one image, both buffers
    constexpr int BKP = WC::BKP;
    constexpr int LDA = WC::LDA, LDB = WC::LDB;
    constexpr int AP = BMK * BKP / 1024, BP = BNC * BKP >= 1024 ? BNC * BKP / 1024 : 1;   // load passes per thread and step (BKP pixels x BMK / BNC floats)
    constexpr int AQ = BMK / 4, BQ = BNC / 4;            // float4 quads per pixel row
    constexpr bool WSPLIT = WC::WSPLIT;
    constexpr int NTN = WC::NTN;
    constexpr int WNW = WSPLIT ? 4 : (BNC >= 64 ? BNC / 64 : 1);   // waves along the column axis (WSPLIT: along the pixels)
    constexpr bool BHALF = BNC * BKP < 1024;             // fewer B quads than threads: the upper threads stage nothing
    float *At = smem;                                    // [2][BKP][LDA]  dy tile   (pixel, k)
    float *Bt = smem + WC::NIMG * 2 * BKP * LDA;         // [2][BKP][LDB]  x gather  (pixel, kk)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WSPLIT ? 0 : wave / WNW, wn = WSPLIT ? 0 : wave % WNW, li = lane & 31, lh = lane >> 5;
    // All tiles of one (split, group) read the same pixel range of x and dy.  Blocks are dispatched round-robin
    // over the 8 XCDs, so the linear block id is remapped such that each XCD works through whole (split, group)
    // sets one after the other: the pixel range is fetched into ONE L2 instead of all eight.
    const int ntile = gridDim.x;
    const int lin = blockIdx.x + ntile * (blockIdx.y + gridDim.y * blockIdx.z);
    const int virt = xcd_remap(lin, ntile * gridDim.y * gridDim.z);
    const int tile = virt % ntile;
    const int sg = virt / ntile;
    const int grp = sg % gridDim.y;
    const int split = sg / gridDim.y;
    const int kt_ = tile / g.nCt, ct = tile - kt_ * g.nCt;
    const int k0 = kt_ * BMK, c0 = ct * BNC;
    // bias gradient = column sums of dy: the blocks of the first column tile already stream every dy tile of
    // their rows through registers, so they add them up on the way (bias_out: [split][group][K] partials)
    const bool do_bias = bias_out != nullptr && ct == 0;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const int m_begin = split * g.chunk;
    const int m_end = min(g.Mtot, m_begin + g.chunk);
    const int ng = g.Mtot / (g.OH * g.OW);
    const float *dyg = dy + (size_t)grp * g.Mtot * g.K;
    const float *xg = x + (size_t)grp * ng * g.IH * g.IW * g.IC;

    // A: pixel = tid / AQ + (256 / AQ) * j, quad = tid % AQ.   B: pixel = tid / BQ + (256 / BQ) * j, quad = tid % BQ.
    const int apix = tid / AQ, aq = tid % AQ, bpix = tid / BQ, bq = tid % BQ;
    const int ka = k0 + 4 * aq;
    const int kkb = c0 + 4 * bq;                         // the thread's 4 filter columns are fixed for the whole kernel
    int br[4], bs[4], bc[4];
    bool bv[4];
#pragma unroll
    // Columns are enumerated in PLAIN (tap, channel) order so that the 32/64 quads of one pixel row read one
    // contiguous run of channels (512 B - 1 KB per pixel); the results are scattered into the channel-blocked
    // filter layout (common.h: kidx) by the epilogue.
    for (int t = 0; t < 4; ++t) {
        int kk = kkb + t;
        bv[t] = kk < g.RSC && !(BHALF && bpix >= BKP);
        int kc = bv[t] ? kk : 0;
        const int tap = kc / g.IC;
        bc[t] = kc - tap * g.IC;
        br[t] = tap / g.S;
        bs[t] = tap - br[t] * g.S;
    }

    f32x16 acc[2][NTN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NTN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Three-phase K-step as in gemm_core.h: (1) issue the loads of step st+1 from staged buffer offsets,
    // (2) MFMAs of step st with the offset arithmetic of step st+2 in their shadow, (3) wait + LDS stores +
    // barrier.  Out-of-range pixels / taps / channels read zeros through the descriptors.
    // descriptors are based at the first image of this split: 32-bit offsets only span the split's pixel range
    const int img0 = m_begin / (g.OH * g.OW);
    const int pix0 = img0 * g.OH * g.OW;
    const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(dyg + (size_t)pix0 * g.K, (size_t)(g.Mtot - pix0) * g.K * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(xg + (size_t)img0 * g.IH * g.IW * g.IC,
                                                  (size_t)(ng - img0) * g.IH * g.IW * g.IC * sizeof(float));
    float4 ra[AP], rb[BP];
    unsigned oa[AP], ob[BP];
    int am[AP], bm[BP];                                  // pixel index per slot
    int bow[BP], boh[BP], bbase[BP];                     // MODE 2: walked coordinates (ow*sw, oh*sh) and element base
    const int tap_h = br[0] * g.dh - g.ph, tap_w = bs[0] * g.dw - g.pw;
    const int tap_off = (tap_h * g.IW + tap_w) * g.IC + bc[0];

    auto decode_b = [&](int j) {                         // (re)compute walked state of B slot j from bm[j]
        const int mm = bm[j] < g.Mtot ? bm[j] : pix0;
        const int ow = mm % g.OW, t2 = mm / g.OW;
        const int oh = t2 % g.OH, n = t2 / g.OH - img0;
        bow[j] = ow * g.sw;
        boh[j] = oh * g.sh;
        bbase[j] = ((n * g.IH + boh[j]) * g.IW + bow[j]) * g.IC;
    };
    // MODE 3 state: wave-uniform position of the staged step's first pixel + per-thread constants
    int s_p = m_begin, s_ow, s_oh, s_base4;
    int dA[AP], vA[AP], dB[BP], vB[BP], tw[BP];
    {
        const int ow = s_p % g.OW, t2 = s_p / g.OW;
        s_ow = ow;
        s_oh = t2 % g.OH;
        s_base4 = (((t2 / g.OH - img0) * g.IH + s_oh * g.sh) * g.IW + ow * g.sw) * g.IC * 4;
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            dA[j] = apix + (256 / AQ) * j;
            vA[j] = (dA[j] * g.K + ka) * 4;
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            dB[j] = bpix + (256 / BQ) * j;
            vB[j] = (dB[j] * g.sw * g.IC + tap_off) * 4;
            tw[j] = dB[j] * g.sw + tap_w;
        }
    }
    int so_a = 0, so_b = 0;                              // MODE 4: wave-uniform byte offsets of the step's first pixel (scalar operand)
    auto offsets = [&]() {
        if constexpr (MODE == 4) {
            // dense (R = S = 1, stride 1, no padding: the Winograd-domain weight gradients): x is the [pixels][C] matrix itself.
            // Per slot one compare + select; the pixel walk is a scalar offset in the load instructions.
            const int left = m_end - s_p;
            so_a = (s_p - pix0) * g.K * 4;
            so_b = (s_p - pix0) * g.IC * 4;
#pragma unroll
            for (int j = 0; j < AP; ++j) oa[j] = ((dA[j] < left) & (ka < g.K)) ? (unsigned)vA[j] : OOB;
#pragma unroll
            for (int j = 0; j < BP; ++j) ob[j] = ((dB[j] < left) & bv[0]) ? (unsigned)vB[j] : OOB;
            return;
        }
        if constexpr (MODE == 3) {
            const int left = m_end - s_p;                    // pixels of this split from the step's first one on
            const int s_a = (s_p - pix0) * g.K * 4;
#pragma unroll
            for (int j = 0; j < AP; ++j) oa[j] = ((dA[j] < left) & (ka < g.K)) ? (unsigned)(vA[j] + s_a) : OOB;
            const int wrapd = g.OW - s_ow;                   // deltas >= wrapd fall on the next output row
            const bool last_row = s_oh + 1 == g.OH;          // ... which is row 0 of the next image
            const int rowjump = ((g.sh * g.IW - g.OW * g.sw) + (last_row ? (g.IH - g.OH * g.sh) * g.IW : 0)) * g.IC * 4;
            const bool okA = (unsigned)(s_oh * g.sh + tap_h) < (unsigned)g.IH;
            const bool okB = (unsigned)((last_row ? 0 : (s_oh + 1) * g.sh) + tap_h) < (unsigned)g.IH;
            const int s_iw = s_ow * g.sw;
#pragma unroll
            for (int j = 0; j < BP; ++j) {
                const bool c = dB[j] >= wrapd;
                const int iw = s_iw + tw[j] - (c ? g.OW * g.sw : 0);
                const bool ok = (dB[j] < left) & bv[0] & (c ? okB : okA) & ((unsigned)iw < (unsigned)g.IW);
                ob[j] = ok ? (unsigned)(vB[j] + s_base4 + (c ? rowjump : 0)) : OOB;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < AP; ++j) oa[j] = (am[j] < m_end && ka < g.K) ? (unsigned)((am[j] - pix0) * g.K + ka) * 4u : OOB;
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            if (MODE != 2) decode_b(j);
            const bool ok = bm[j] < m_end && bv[0] && (unsigned)(boh[j] + tap_h) < (unsigned)g.IH &&
                            (unsigned)(bow[j] + tap_w) < (unsigned)g.IW;
            ob[j] = ok ? (unsigned)(bbase[j] + tap_off) * 4u : OOB;
        }
    };
    auto advance = [&]() {                               // all slots move on by BKP pixels
        if constexpr (MODE == 4) {
            s_p += BKP;
            return;
        }
        if constexpr (MODE == 3) {                           // scalar unit only
            s_p += BKP;
            s_ow += BKP;
            s_base4 += BKP * g.sw * g.IC * 4;
            const bool w1 = s_ow >= g.OW;
            s_ow -= w1 ? g.OW : 0;
            s_oh += w1 ? 1 : 0;
            s_base4 += w1 ? (g.sh * g.IW - g.OW * g.sw) * g.IC * 4 : 0;
            const bool w2 = s_oh >= g.OH;
            s_oh = w2 ? 0 : s_oh;
            s_base4 += w2 ? (g.IH - g.OH * g.sh) * g.IW * g.IC * 4 : 0;
            return;
        }
#pragma unroll
        for (int j = 0; j < AP; ++j) am[j] += BKP;
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            bm[j] += BKP;
            if (MODE == 2) {                             // OW >= 16: at most BKP/16 row wraps (each maybe an image wrap)
                bow[j] += BKP * g.sw;
                bbase[j] += BKP * g.sw * g.IC;
#pragma unroll
                for (int i = 0; i < BKP / 16; ++i) {
                    const bool w1 = bow[j] >= g.OW * g.sw;
                    bow[j] -= w1 ? g.OW * g.sw : 0;
                    boh[j] += w1 ? g.sh : 0;
                    bbase[j] += w1 ? (g.sh * g.IW - g.OW * g.sw) * g.IC : 0;
                    const bool w2 = boh[j] >= g.OH * g.sh;
                    boh[j] -= w2 ? g.OH * g.sh : 0;
                    bbase[j] += w2 ? (g.IH - g.OH * g.sh) * g.IW * g.IC : 0;
                }
            }
        }
    };
    auto fetch = [&]() {
        if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < AP; ++j) ra[j] = buf_load4s(rs_dy, oa[j], so_a);
#pragma unroll
            for (int j = 0; j < BP; ++j) rb[j] = buf_load4s(rs_x, ob[j], so_b);
        } else if (MODE > 0) {
#pragma unroll
            for (int j = 0; j < AP; ++j) ra[j] = buf_load4(rs_dy, oa[j]);
#pragma unroll
            for (int j = 0; j < BP; ++j) rb[j] = buf_load4(rs_x, ob[j]);
        } else {
#pragma unroll
            for (int j = 0; j < AP; ++j) {
                ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (am[j] < m_end) {
                    const float *pa = dyg + (size_t)am[j] * g.K + ka;
                    if (ka + 0 < g.K) ra[j].x = pa[0];
                    if (ka + 1 < g.K) ra[j].y = pa[1];
                    if (ka + 2 < g.K) ra[j].z = pa[2];
                    if (ka + 3 < g.K) ra[j].w = pa[3];
                }
            }
#pragma unroll
            for (int j = 0; j < BP; ++j) {
                float e[4] = {0.f, 0.f, 0.f, 0.f};
                const int m = bm[j];
                if (m < m_end) {
                    const int ow = m % g.OW, t2 = m / g.OW;
                    const int oh = t2 % g.OH, n = t2 / g.OH;
                    const int bh = oh * g.sh - g.ph, bw = ow * g.sw - g.pw;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int ih = bh + br[t] * g.dh, iw = bw + bs[t] * g.dw;
                        if (bv[t] && (unsigned)ih < (unsigned)g.IH && (unsigned)iw < (unsigned)g.IW)
                            e[t] = xg[(((size_t)n * g.IH + ih) * g.IW + iw) * g.IC + bc[t]];
                    }
                }
                rb[j] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };
    auto sstore = [&](int buf, auto bias_tag) {
        if constexpr (decltype(bias_tag)::value) {
#pragma unroll
            for (int j = 0; j < AP; ++j) {
                bsum.x += ra[j].x; bsum.y += ra[j].y; bsum.z += ra[j].z; bsum.w += ra[j].w;
            }
        }
        if constexpr (X3) {
#pragma unroll
            for (int j = 0; j < AP; ++j) {
                uint2 h, m, l;
                split3_bf16x4(ra[j], h, m, l);
                float *d = &At[(buf * BKP + apix + (256 / AQ) * j) * LDA + 2 * aq];
                *reinterpret_cast<uint2 *>(d) = h;
                *reinterpret_cast<uint2 *>(d + IMG_A) = m;
                *reinterpret_cast<uint2 *>(d + 2 * IMG_A) = l;
            }
#pragma unroll
            for (int j = 0; j < BP; ++j) {
                uint2 h, m, l;
                split3_bf16x4(rb[j], h, m, l);
                float *d = &Bt[(buf * BKP + bpix + (256 / BQ) * j) * LDB + 2 * bq];
                *reinterpret_cast<uint2 *>(d) = h;
                *reinterpret_cast<uint2 *>(d + IMG_B) = m;
                *reinterpret_cast<uint2 *>(d + 2 * IMG_B) = l;
            }
        } else if constexpr (BF16) {
#pragma unroll
            for (int j = 0; j < AP; ++j)
                *reinterpret_cast<uint2 *>(&At[(buf * BKP + apix + (256 / AQ) * j) * LDA + 2 * aq]) = pack_bf16x4(ra[j]);
#pragma unroll
            for (int j = 0; j < BP; ++j)
                *reinterpret_cast<uint2 *>(&Bt[(buf * BKP + bpix + (256 / BQ) * j) * LDB + 2 * bq]) = pack_bf16x4(rb[j]);
        } else {
#pragma unroll
            for (int j = 0; j < AP; ++j)
                *reinterpret_cast<float4 *>(&At[(buf * BKP + apix + (256 / AQ) * j) * LDA + 4 * aq]) = ra[j];
            if (!BHALF || bpix < BKP) {
#pragma unroll
                for (int j = 0; j < BP; ++j)
                    *reinterpret_cast<float4 *>(&Bt[(buf * BKP + bpix + (256 / BQ) * j) * LDB + 4 * bq]) = rb[j];
            }
        }
    };

#pragma unroll
    for (int j = 0; j < AP; ++j) am[j] = m_begin + apix + (256 / AQ) * j;
#pragma unroll
    for (int j = 0; j < BP; ++j) {
        bm[j] = m_begin + bpix + (256 / BQ) * j;
        decode_b(j);
    }
    const int nsteps = (m_end - m_begin + BKP - 1) / BKP;
    // transposed-read lane constants (bf16): 16-lane group G -> rows 16*(G&1).., pixel half h = G>>1
    const int trq = (lane & 15) >> 2, trp = lane & 3, trg = (lane >> 4) & 1;
    // the K loop exists twice: the blocks that also sum the bias gradient pay for those adds, the others do not
    auto run = [&](auto bias_tag) {
    offsets();
    fetch();
    sstore(0, bias_tag);
    advance();
    offsets();
    __syncthreads();
    // one K-step with the LDS buffer index as a compile-time constant (loop unrolled by two below)
    auto w_step = [&](auto cur_tag) {
        constexpr int cur = decltype(cur_tag)::value;
        fetch();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (X3) {
            const float *Ac = At + (cur * BKP + 8 * lh + trq) * LDA + wm * 32 + 8 * trg + 2 * trp;
            const float *Bc = Bt + (cur * BKP + 8 * lh + trq) * LDB + wn * 32 + 8 * trg + 2 * trp;
            bf16x8 a[3][2], b[3][2];
#pragma unroll
            for (int gi = 0; gi < 3; ++gi)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(Ac + gi * IMG_A + 16 * t));
                    const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(Ac + gi * IMG_A + 4 * LDA + 16 * t));
                    a[gi][t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
                    const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(Bc + gi * IMG_B + 16 * t));
                    const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(Bc + gi * IMG_B + 4 * LDB + 16 * t));
                    b[gi][t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};      // h*h first (see gemm_core.h)
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[p]][tm], b[PB[p]][tn], acc[tm][tn], 0, 0, 0);
        } else if constexpr (BF16) {
            // 4-byte units: pixel (8h + q) of the step, rows wm*64 + 16*trg + 4p  (2 bf16 per unit)
            const float *Ac = At + (cur * BKP + 8 * lh + trq) * LDA + wm * 32 + 8 * trg + 2 * trp;
            const float *Bc = Bt + (cur * BKP + 8 * lh + trq) * LDB + wn * 32 + 8 * trg + 2 * trp;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(Ac + 16 * s * LDA + 16 * t));
                    const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(Ac + (16 * s + 4) * LDA + 16 * t));
                    a[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
                    const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(Bc + 16 * s * LDB + 16 * t));
                    const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(Bc + (16 * s + 4) * LDB + 16 * t));
                    b[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
            }
        } else {
            const float *Ac = At + cur * BKP * LDA + wm * 64 + 2 * li;
            if constexpr (NTN == 2) {
                const float *Bc = Bt + cur * BKP * LDB + wn * 64 + 2 * li;
                constexpr int NS = WSPLIT ? BKP / 8 : BKP / 2;              // pixel pairs this wave multiplies
                const int s0 = WSPLIT ? wave * NS : 0;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const float2 a = *reinterpret_cast<const float2 *>(Ac + (2 * (s0 + s) + lh) * LDA);
                    const float2 b = *reinterpret_cast<const float2 *>(Bc + (2 * (s0 + s) + lh) * LDB);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1][1], 0, 0, 0);
                }
            } else {
                const float *Bc = Bt + cur * BKP * LDB + li;               // one 32-column tile, column li
#pragma unroll
                for (int s = 0; s < BKP / 2; ++s) {
                    const float2 a = *reinterpret_cast<const float2 *>(Ac + (2 * s + lh) * LDA);
                    const float b = Bc[(2 * s + lh) * LDB];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, acc[0][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, acc[1][0], 0, 0, 0);
                }
            }
        }
        advance();
        offsets();
        __builtin_amdgcn_sched_barrier(0);
        sstore(cur ^ 1, bias_tag);
        __syncthreads();
    };
    int st = 0;
    for (; st + 1 < nsteps; st += 2) {
        w_step(std::integral_constant<int, 0>{});
        w_step(std::integral_constant<int, 1>{});
    }
    if (st < nsteps) w_step(std::integral_constant<int, 0>{});
    };
    if (do_bias) run(std::true_type{});
    else run(std::false_type{});

    if (do_bias) {       // fixed-order reduction over the 256/AQ pixel rows of threads (the last barrier of the loop has passed)
        float *red = smem;
        *reinterpret_cast<float4 *>(&red[apix * BMK + 4 * aq]) = bsum;
        __syncthreads();
        if (tid < BMK) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 256 / AQ; ++r) t += red[r * BMK + tid];
            if (k0 + tid < g.K) bias_out[((size_t)split * groups + grp) * g.K + k0 + tid] = t;
        }
    }

    if constexpr (WSPLIT) {
        // the four waves hold partial sums of the same tile: ((w3 + w2) + w1) + w0 through LDS, element (register, lane)
        float *red = smem;
        for (int w = 3; w >= 1; --w) {
            __syncthreads();                             // the operand buffers / the previous round's values are no longer read
            if (wave == w) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                        for (int r = 0; r < 16; ++r) red[((tm * 2 + tn) * 16 + r) * 64 + lane] = acc[tm][tn][r];
            }
            __syncthreads();
            if (wave == w - 1) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[tm][tn][r] = red[((tm * 2 + tn) * 16 + r) * 64 + lane] + acc[tm][tn][r];
            }
        }
        if (wave != 0) return;
    }
    // slab [split][group][K][RSC]
    float *o = out + ((size_t)split * groups + grp) * (size_t)g.K * g.RSC;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < NTN; ++tn) {
            const int colp = c0 + wn * 64 + (BF16 ? 32 * tn + li : (NTN == 2 ? 2 * li + tn : li));   // plain (tap, channel) index
            if (colp >= g.RSC) continue;
            const int tap = colp / g.IC;
            const int col = kidx(tap, colp - tap * g.IC, g.R * g.S, chan_block(g.IC));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int row = k0 + wm * 64 + (BF16 ? 32 * tm + rr : 2 * rr + tm);
                if (row < g.K) o[(size_t)row * g.RSC + col] = acc[tm][tn][r];
            }
        }
}

template <int BMK, int BNC, int MODE>
__global__ __launch_bounds__(256, (BMK == 128 && MODE > 0) ? 4 : 3) void conv_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                         float *__restrict__ out, float *__restrict__ bias_out,
                                                         WgradGeom g, int groups)
{
    __shared__ __attribute__((aligned(16))) float smem[WgradCfg<BMK, BNC, 0>::LDS_FLOATS];
    conv_wgrad_body<BMK, BNC, MODE, 0>(x, dy, out, bias_out, g, groups, smem);
}
template <int BMK, int BNC, int MODE>
__global__ __launch_bounds__(256, BMK == 128 ? 3 : 2) void conv_wgrad_bf16_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                              float *__restrict__ out, float *__restrict__ bias_out,
                                                              WgradGeom g, int groups)
{
    __shared__ __attribute__((aligned(16))) float smem[WgradCfg<BMK, BNC, 1>::LDS_FLOATS];
    conv_wgrad_body<BMK, BNC, MODE, 1>(x, dy, out, bias_out, g, groups, smem);
}
template <int BMK, int BNC, int MODE>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                               float *__restrict__ out, float *__restrict__ bias_out,
                                                               WgradGeom g, int groups)
{
    __shared__ __attribute__((aligned(16))) float smem[WgradCfg<BMK, BNC, 2>::LDS_FLOATS];
    conv_wgrad_body<BMK, BNC, MODE, 2>(x, dy, out, bias_out, g, groups, smem);
}

// out[i] = sum_k slabs[k][i] in the fixed order k = 0, 1, ... (deterministic).  float4 per thread, four slabs in
// flight per accumulation step so that the loads overlap (the sums are still formed in slab order).
template <bool VEC>      // VEC: n % 4 == 0 and 16-byte aligned pointers
__global__ void slab_reduce_kernel(const float *__restrict__ slabs, float *__restrict__ out, size_t n, int splits)
{
    const size_t n4 = VEC ? n >> 2 : 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 *p = reinterpret_cast<const float4 *>(slabs) + i;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 4 <= splits; k += 4) {
            const float4 a = p[(size_t)k * n4], b = p[(size_t)(k + 1) * n4], c = p[(size_t)(k + 2) * n4], d = p[(size_t)(k + 3) * n4];
            s.x = (((s.x + a.x) + b.x) + c.x) + d.x;
            s.y = (((s.y + a.y) + b.y) + c.y) + d.y;
            s.z = (((s.z + a.z) + b.z) + c.z) + d.z;
            s.w = (((s.w + a.w) + b.w) + c.w) + d.w;
        }
        for (; k < splits; ++k) {
            const float4 a = p[(size_t)k * n4];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        reinterpret_cast<float4 *>(out)[i] = s;
    }
    if (VEC) return;
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
// grid (ceil(cols/64)), block (64, 16): thread (c, ty) sums chunks ty, ty+16, ...; fixed-order LDS combine
__global__ void colsum_final_kernel(const float *__restrict__ part, int cols, int nchunks, float *__restrict__ out)
{
    __shared__ float red[16][64];
    const int c = blockIdx.x * 64 + threadIdx.x;
    float s = 0.f;
    if (c < cols)
        for (int k = threadIdx.y; k < nchunks; k += 16) s += part[(size_t)k * cols + c];
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        out[c] = t;
    }
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

// hmode: act and w are bf16 arrays (mq_cast_bf16 copies of the f32 tensors), passed as float* in disguise
template <bool DGRAD>
static int launch_nt(const mqConvDesc *d, const float *act, const float *w, const float *bias, float *out, hipStream_t st,
                     bool hmode = false, NtEpi epi = NtEpi{nullptr, 1, 0, 1, 0.f, 0, nullptr})
{
    const bool want_bf16 = d->precision == MQ_PRECISION_BF16 && !hmode;
    const bool want_x3 = d->precision == MQ_PRECISION_F32X3;
    if (use_d32() && epi.act != 3 && d->precision == MQ_PRECISION_F32 && !hmode && ((uintptr_t)act % 16 == 0) &&
        ((uintptr_t)w % 16 == 0) && d32_nt_ok(d, DGRAD))
#if MQ_VARIANTS
        return d32_nt_launch(d, DGRAD, act, w, bias, epi.mask, epi.d1, epi.d2, epi.d3, epi.act, epi.slope, out, st);
#else
        return ::mq::fail("MQGAN_D32 needs the test-variant library");
#endif
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
    // 32-bit element offsets inside one tile: the images a 128-row tile can touch must span < 2 GB
    const size_t img_bytes = (size_t)g.IH * g.IW * g.IC * (hmode ? 2 : sizeof(float));
    const size_t tile_imgs = (size_t)cdiv(GEMM_BM, (int64_t)g.OH * g.OW) + 1;
    bool vec = (g.IC % 4 == 0) && ((uintptr_t)act % 16 == 0) && ((uintptr_t)w % 16 == 0) &&
               img_bytes * tile_imgs < 0x7FFFFFF0ull && w_gs * sizeof(float) < 0xFFFFFFF0ull;
    const bool strided = g.sh > 1 || g.sw > 1;
    if (hmode)
        MQ_REQUIRE(vec && g.IC % 16 == 0 && (!strided || !DGRAD || (g.sh <= 2 && g.sw <= 2)),
                   "bf16-operand conv needs 16-byte aligned tensors, a reduction channel count that is a multiple of 16 "
                   "and strides <= 2 (got C=%d)", g.IC);
    const bool parity = DGRAD && vec && strided && g.sh <= 2 && g.sw <= 2;
    if (DGRAD && strided && !parity) vec = false;      // generic strided dgrad only exists on the scalar path
    // column tile: the one that wastes the fewest MFMAs on padding, weighted by the measured relative
    // efficiency of each tile shape (wider tiles reuse the A fragments more)
    int BN = 128;
    {
        const int cand[3] = {128, 64, 32};
        const double eff[3] = {1.0, 0.9, 0.75};
        double best = 1e30;
        for (int i = 0; i < 3; ++i) {
            const double cost = (double)cdiv(g.Ncols, cand[i]) * cand[i] / eff[i];
            if (cost < best - 1e-9) { best = cost; BN = cand[i]; }
        }
    }
    const int cb = chan_block(g.IC);
    const bool wide = cb >= 16;
    const bool bf16 = want_bf16 && vec;                // scalar-path layers (C = 1, K = 1 ...) are tiny: they stay f32
    const bool x3 = want_x3 && vec;
    g.nMt = (int)cdiv(g.Mtot, GEMM_BM);
    g.nNt = (int)cdiv(g.Ncols, BN);
    if (parity) {
        ParityInfo pi;
        memset(&pi, 0, sizeof(pi));
        int t0 = 0;
        for (int a = 0; a < g.sh; ++a)
            for (int b = 0; b < g.sw; ++b) {
                const int c = pi.ncls++;
                pi.a[c] = a; pi.b[c] = b;
                pi.Ha[c] = g.OH > a ? (g.OH - a + g.sh - 1) / g.sh : 0;
                pi.Wb[c] = g.OW > b ? (g.OW - b + g.sw - 1) / g.sw : 0;
                // valid taps of a class form an arithmetic progression (strides <= 2)
                int first = -1, second = -1;
                for (int r = 0; r < g.R; ++r)
                    if ((((a + g.ph - r * g.dh) % g.sh) + g.sh) % g.sh == 0) {
                        if (first < 0) first = r; else if (second < 0) second = r;
                        pi.nr[c]++;
                    }
                pi.r0[c] = first < 0 ? 0 : first;
                pi.rstep[c] = second < 0 ? 1 : second - first;
                first = second = -1;
                for (int s = 0; s < g.S; ++s)
                    if ((((b + g.pw - s * g.dw) % g.sw) + g.sw) % g.sw == 0) {
                        if (first < 0) first = s; else if (second < 0) second = s;
                        pi.ns[c]++;
                    }
                pi.s0[c] = first < 0 ? 0 : first;
                pi.sstep[c] = second < 0 ? 1 : second - first;
                pi.tile0[c] = t0;
                t0 += (int)cdiv((int64_t)ng * pi.Ha[c] * pi.Wb[c], GEMM_BM);
            }
        pi.tile0[pi.ncls] = t0;
        MQ_REQUIRE((int64_t)ng * g.OH * g.OW < 0xFFFFFFFFll, "mq_conv_dgrad: a batch group of the strided input gradient has 2^32 pixels or more");
        dim3 pgrid(t0 * g.nNt, d->groups);
#define MQ_PAR(BN_, W_) do { kname("conv_dgrad_parity_kernel<%d, %d>", BN_, W_); hipLaunchKernelGGL((conv_dgrad_parity_kernel<BN_, W_>), pgrid, dim3(256), 0, st, act, w, out, g, pi, act_gs, w_gs, ng, epi); } while (0)
#define MQ_PARH(BN_, W_) do { kname("conv_dgrad_parity_bf16_kernel<%d, %d>", BN_, W_); hipLaunchKernelGGL((conv_dgrad_parity_bf16_kernel<BN_, W_>), pgrid, dim3(256), 0, st, act, w, out, g, pi, act_gs, w_gs, ng, epi); } while (0)
        if (hmode) {
#define MQ_PARHH(BN_) do { kname("conv_dgrad_parity_h_kernel<%d>", BN_); hipLaunchKernelGGL((conv_dgrad_parity_h_kernel<BN_>), pgrid, dim3(256), 0, st, act, w, out, g, pi, act_gs, w_gs, ng, epi); } while (0)
            if (BN == 128) MQ_PARHH(128); else if (BN == 64) MQ_PARHH(64); else MQ_PARHH(32);
#undef MQ_PARHH
        } else if (x3) {
#define MQ_PARX(BN_, W_) do { kname("conv_dgrad_parity_x3_kernel<%d, %d>", BN_, W_); hipLaunchKernelGGL((conv_dgrad_parity_x3_kernel<BN_, W_>), pgrid, dim3(256), 0, st, act, w, out, g, pi, act_gs, w_gs, ng, epi); } while (0)
            const int wr = g.IC % 16 == 0 ? 0 : (wide ? 1 : 4);
            if (BN == 128) { if (wr == 0) MQ_PARX(128, 0); else if (wr == 1) MQ_PARX(128, 1); else MQ_PARX(128, 4); }
            else if (BN == 64) { if (wr == 0) MQ_PARX(64, 0); else if (wr == 1) MQ_PARX(64, 1); else MQ_PARX(64, 4); }
            else { if (wr == 0) MQ_PARX(32, 0); else if (wr == 1) MQ_PARX(32, 1); else MQ_PARX(32, 4); }
#undef MQ_PARX
        } else if (bf16) {
            if (BN == 128) { if (wide) MQ_PARH(128, 2); else MQ_PARH(128, 8); }
            else if (BN == 64) { if (wide) MQ_PARH(64, 2); else MQ_PARH(64, 8); }
            else { if (wide) MQ_PARH(32, 2); else MQ_PARH(32, 8); }
        } else if (g.IC % 16 == 0) {
            if (BN == 128) MQ_PAR(128, 0); else if (BN == 64) MQ_PAR(64, 0); else MQ_PAR(32, 0);
        } else if (BN == 128) { if (wide) MQ_PAR(128, 1); else MQ_PAR(128, 4); }
        else if (BN == 64) { if (wide) MQ_PAR(64, 1); else MQ_PAR(64, 4); }
        else { if (wide) MQ_PAR(32, 1); else MQ_PAR(32, 4); }
#undef MQ_PAR
#undef MQ_PARH
        MQ_LAUNCH_CHECK("mq_conv_dgrad(parity)");
        return 0;
    }
    WalkGeom wg;
    wg.IH = g.IH; wg.IW = g.IW; wg.IC = g.IC; wg.nR = g.R; wg.nS = g.S;
    wg.step_h = DGRAD ? -g.dh : g.dh;
    wg.step_w = DGRAD ? -g.dw : g.dw;
    wg.CB = cb; wg.nchunks = g.IC / cb;
    int mode = vec ? (wide ? ((g.IC % 16 == 0 && !bf16) ? 3 : 2) : 1) : 0;
    // dense GEMM (what the Winograd forms multiply with): 1x1, stride 1, no padding - the activation tensor IS the A matrix
    if (!DGRAD && mode == 3 && !hmode && d->R == 1 && d->S == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 0 && d->pad_w == 0 &&
        d->P == d->H && d->Q == d->W && (size_t)GEMM_BM * g.Kdim * sizeof(float) < 0x7FFFFFF0ull && getenv("MQGAN_NO_DENSE") == nullptr)
        mode = 4;
    // round 6: the dense products with >= 128 output columns and enough rows for 256-row tiles on the pipelined 256 x 128 kernel
    // (MQGAN_NO_DENSE256=1: the 128-row kernel, for A/B runs and the tests that compare the two)
    if (!DGRAD && mode == 4 && BN == 128 && !x3 && !bf16 && !hmode && epi.act == 0 && g.Kdim % 16 == 0 && g.Mtot >= 1024 &&
        (size_t)256 * g.Kdim * sizeof(float) < 0x7FFFFFF0ull && getenv("MQGAN_NO_DENSE256") == nullptr) {
        ConvGeom g2 = g;
        g2.nMt = (int)cdiv(g.Mtot, 256);
        kname("conv_nt_dense256_kernel");
        hipLaunchKernelGGL(conv_nt_dense256_kernel, dim3(g2.nMt * g2.nNt, d->groups), dim3(256), 0, st, act, w, bias, out, g2, act_gs, w_gs);
        MQ_LAUNCH_CHECK("mq_conv_fprop(dense256)");
        return 0;
    }
    dim3 grid(g.nMt * g.nNt, d->groups), block(256);
#define MQ_NT(BN_, M_) do { kname("conv_nt_kernel<%d, %s, %d>", BN_, tf(DGRAD), M_); hipLaunchKernelGGL((conv_nt_kernel<BN_, DGRAD, M_>), grid, block, 0, st, act, w, bias, out, g, wg, act_gs, w_gs, epi); } while (0)
#define MQ_NTH(BN_, M_) do { kname("conv_nt_bf16_kernel<%d, %s, %d>", BN_, tf(DGRAD), M_); hipLaunchKernelGGL((conv_nt_bf16_kernel<BN_, DGRAD, M_>), grid, block, 0, st, act, w, bias, out, g, wg, act_gs, w_gs, epi); } while (0)
#define MQ_NTX(BN_, M_) do { kname("conv_nt_x3_kernel<%d, %s, %d>", BN_, tf(DGRAD), M_); hipLaunchKernelGGL((conv_nt_x3_kernel<BN_, DGRAD, M_>), grid, block, 0, st, act, w, bias, out, g, wg, act_gs, w_gs, epi); } while (0)
#define MQ_NTHH(BN_) do { kname("conv_nt_h_kernel<%d, %s>", BN_, tf(DGRAD)); hipLaunchKernelGGL((conv_nt_h_kernel<BN_, DGRAD>), grid, block, 0, st, act, w, bias, out, g, wg, act_gs, w_gs, epi); } while (0)
#define MQ_NT3(BN_) do { if (hmode) MQ_NTHH(BN_); else if (x3) { if (mode == 4) { if constexpr (!DGRAD) MQ_NTX(BN_, 4); } else if (mode == 3) MQ_NTX(BN_, 3); else if (mode == 2) MQ_NTX(BN_, 2); else MQ_NTX(BN_, 1); } \
                         else if (bf16) { if (mode == 2) MQ_NTH(BN_, 2); else MQ_NTH(BN_, 1); } \
                         else if (mode == 4) { if constexpr (!DGRAD) MQ_NT(BN_, 4); } \
                         else if (mode == 3) MQ_NT(BN_, 3); else if (mode == 2) MQ_NT(BN_, 2); else if (mode == 1) MQ_NT(BN_, 1); \
                         else MQ_NT(BN_, 0); } while (0)
    if (BN == 128) MQ_NT3(128);
    else if (BN == 64) MQ_NT3(64);
    else MQ_NT3(32);
#undef MQ_NT3
#undef MQ_NTHH
#undef MQ_NTX
#undef MQ_NTH
#undef MQ_NT
    MQ_LAUNCH_CHECK(DGRAD ? "mq_conv_dgrad" : "mq_conv_fprop");
    return 0;
}

static bool wgrad_vec_shape(const mqConvDesc *d) { return d->C % 4 == 0 && d->K % 4 == 0; }

static void wgrad_plan(const mqConvDesc *d, WgradGeom &g)
{
    const int bk = (d->precision == MQ_PRECISION_BF16 && wgrad_vec_shape(d)) ? 32 : WG_BK;
    const int ng = d->N / d->groups;
    g.OH = d->P; g.OW = d->Q; g.IH = d->H; g.IW = d->W; g.IC = d->C;
    g.R = d->R; g.S = d->S; g.sh = d->stride_h; g.sw = d->stride_w; g.ph = d->pad_h; g.pw = d->pad_w;
    g.dh = d->dil_h; g.dw = d->dil_w;
    g.Mtot = ng * d->P * d->Q;
    g.K = d->K;
    g.RSC = d->R * d->S * d->C;
    g.bmk = g.K <= 64 ? 64 : 128;
    g.bnc = g.K <= 64 ? 256 : 128;
    if (d->precision == MQ_PRECISION_F32 && wgrad_vec_shape(d) && getenv("MQGAN_NO_NARROW_WGRAD") == nullptr) {
        if (g.RSC <= 32 && g.K > 32) { g.bmk = 256; g.bnc = 32; }         // skinny filter rows: no column padding
        else if (g.K <= 64 && g.RSC <= 64) { g.bmk = 64; g.bnc = 64; }
    }
    g.nKt = (int)cdiv(g.K, g.bmk);
    g.nCt = (int)cdiv(g.RSC, g.bnc);
    // Split the pixel reduction so that the grid fills whole rounds of the resident block slots (256 CUs x the
    // blocks a CU holds): among the split counts that keep >= 32 K-steps per split, take the one with the best
    // fill  tiles*s / (rounds * slots), charging a little for every extra round of slabs (write + re-read).
    const int64_t tiles = (int64_t)g.nKt * g.nCt * d->groups;
    const bool x3 = d->precision == MQ_PRECISION_F32X3 && wgrad_vec_shape(d);
    const int64_t slots = x3 ? 512 : 256 * ((g.bmk == 128 ? 4 : 3) - (bk == 32 ? 1 : 0));   // resident blocks per CU (registers / LDS)
    int64_t max_splits = cdiv(g.Mtot, bk * 32);
    if (max_splits > 512) max_splits = 512;
    if (max_splits < 1) max_splits = 1;
    int64_t lo = slots / tiles, hi = 4 * slots / tiles + 1;      // from "one round" up to about four rounds
    if (lo < 1) lo = 1;
    if (lo > max_splits) lo = max_splits;
    if (hi > max_splits) hi = max_splits;
    int64_t splits = lo;
    double best = -1.0;
    for (int64_t sp = lo; sp <= hi; ++sp) {
        const int64_t blocks = tiles * sp;
        const double rounds = (double)cdiv(blocks, slots);
        const double score = (double)blocks / (rounds * (double)slots) - 0.004 * rounds;
        if (score > best + 1e-12) { best = score; splits = sp; }
    }
    // 32-bit offsets inside one split: the images a split touches (plus two of slack) must span < 2 GB of x and of dy
    {
        const int64_t pq = (int64_t)d->P * d->Q;
        const int64_t img = 4 * std::max((int64_t)d->H * d->W * d->C, pq * (int64_t)d->K);
        while (splits < max_splits && (cdiv(cdiv(g.Mtot, splits), pq) + 2) * img >= 0x7FFFFFF0ll) ++splits;
    }
    int64_t chunk = cdiv(cdiv(g.Mtot, splits), bk) * bk;
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

extern "C" int mq_conv_fprop_leaky(const mqConvDesc *d, const float *x, const float *w, const float *bias, const uint8_t *mask,
                                   int64_t d1, int64_t d2, int64_t d3, float slope, float *y, mqStream stream)
{
    if (check_desc(d, "mq_conv_fprop_leaky")) return 1;
    MQ_REQUIRE(x && w && y, "mq_conv_fprop_leaky: null pointer");
    MQ_REQUIRE(d->C % 4 == 0, "mq_conv_fprop_leaky: the fused epilogue exists on the vector path only (C %% 4 == 0, got %d)", d->C);
    MQ_REQUIRE(mask == nullptr || (d1 > 0 && d3 > 0 && d2 >= 0 && d1 < INT32_MAX && d2 < INT32_MAX && d3 < INT32_MAX),
               "mq_conv_fprop_leaky: bad mask map");
    NtEpi epi{mask, (int)(mask ? d1 : 1), (int)(mask ? d2 : 0), (int)(mask ? d3 : 1), slope, 2, nullptr};
    return launch_nt<false>(d, x, w, bias, y, S(stream), false, epi);
}

extern "C" int mq_conv_fprop_h(const mqConvDesc *d, const uint16_t *x_h, const uint16_t *w_h, const float *bias, float *y,
                               mqStream stream)
{
    if (check_desc(d, "mq_conv_fprop_h")) return 1;
    MQ_REQUIRE(x_h && w_h && y, "mq_conv_fprop_h: null pointer");
    return launch_nt<false>(d, reinterpret_cast<const float *>(x_h), reinterpret_cast<const float *>(w_h), bias, y, S(stream), true);
}

extern "C" int mq_conv_dgrad_h(const mqConvDesc *d, const uint16_t *dy_h, const uint16_t *w_crsk_h, float *dx, mqStream stream)
{
    if (check_desc(d, "mq_conv_dgrad_h")) return 1;
    MQ_REQUIRE(dy_h && w_crsk_h && dx, "mq_conv_dgrad_h: null pointer");
    return launch_nt<true>(d, reinterpret_cast<const float *>(dy_h), reinterpret_cast<const float *>(w_crsk_h), nullptr, dx,
                           S(stream), true);
}

extern "C" int mq_conv_dgrad(const mqConvDesc *d, const float *dy, const float *w_crsk, float *dx, mqStream stream)
{
    if (check_desc(d, "mq_conv_dgrad")) return 1;
    MQ_REQUIRE(dy && w_crsk && dx, "mq_conv_dgrad: null pointer");
    return launch_nt<true>(d, dy, w_crsk, nullptr, dx, S(stream));
}

// dgrad of a conv whose input x came out of a LeakyReLU + masked_fill layer, returning the gradient with respect to that
// layer's PRE-activation: dx * (x > 0 ? 1 : slope), zero on the producer's padded frames (mask index of input pixel m:
// (m / d1) * d2 + m % d3, m = (n * H + h) * W + w).  Same engines and summation order as mq_conv_dgrad.
extern "C" int mq_conv_dgrad_gated(const mqConvDesc *d, const float *dy, const float *w_crsk, const float *x, const uint8_t *mask,
                                   int64_t d1, int64_t d2, int64_t d3, float slope, float *dx, mqStream stream)
{
    if (check_desc(d, "mq_conv_dgrad_gated")) return 1;
    MQ_REQUIRE(dy && w_crsk && x && dx, "mq_conv_dgrad_gated: null pointer");
    MQ_REQUIRE(!mask || (d1 > 0 && d3 > 0 && d1 < INT32_MAX && d2 < INT32_MAX && d3 < INT32_MAX && d2 >= 0),
               "mq_conv_dgrad_gated: bad mask map");
    NtEpi epi{mask, (int)(mask ? d1 : 1), (int)(mask ? d2 : 0), (int)(mask ? d3 : 1), slope, 3, x};
    return launch_nt<true>(d, dy, w_crsk, nullptr, dx, S(stream), false, epi);
}

static int colsum_launch(const float *x, int64_t rows, int cols, float *out, float *workspace, hipStream_t st);

// bias partials live behind the slabs: [splits][groups][K] (vector path) or the column-sum scratch (scalar path)
static size_t wgrad_bias_floats(const mqConvDesc *d, const WgradGeom &g)
{
    const size_t a = (size_t)g.splits * d->groups * g.K, b = (size_t)1024 * g.K;
    return a > b ? a : b;
}

extern "C" size_t mq_conv_wgrad_workspace(const mqConvDesc *d)
{
    if (check_desc(d, "mq_conv_wgrad_workspace")) return 0;
    WgradGeom g;
    wgrad_plan(d, g);
    if (g.splits <= 1) return 0;
    return (size_t)g.splits * d->groups * g.K * g.RSC * sizeof(float);
}

extern "C" size_t mq_conv_wgrad_bias_workspace(const mqConvDesc *d)
{
    if (check_desc(d, "mq_conv_wgrad_bias_workspace")) return 0;
    WgradGeom g;
    wgrad_plan(d, g);
    return mq_conv_wgrad_workspace(d) + wgrad_bias_floats(d, g) * sizeof(float);
}

static int wgrad_impl(const mqConvDesc *d, const float *x, const float *dy, float *dw, float *dbias, void *workspace,
                      size_t workspace_bytes, hipStream_t st, const char *who)
{
    if (check_desc(d, who)) return 1;
    MQ_REQUIRE(x && dy && dw, "%s: null pointer", who);
    WgradGeom g;
    wgrad_plan(d, g);
    const size_t n = (size_t)d->groups * g.K * g.RSC;
    const size_t slab_bytes = g.splits > 1 ? (size_t)g.splits * n * sizeof(float) : 0;
    const size_t need = slab_bytes + (dbias ? wgrad_bias_floats(d, g) * sizeof(float) : 0);
    MQ_REQUIRE(workspace_bytes >= need && (need == 0 || workspace != nullptr), "%s: workspace too small (%zu < %zu)", who,
               workspace_bytes, need);
    float *target = g.splits > 1 ? reinterpret_cast<float *>(workspace) : dw;
    float *bias_ws = dbias ? reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + slab_bytes) : nullptr;
    // 32-bit offsets inside one split: its images must span < 2 GB of x and of dy
    const size_t split_imgs = (size_t)cdiv(g.chunk, (int64_t)d->P * d->Q) + 2;
    const bool vec = wgrad_vec_shape(d) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dy % 16 == 0) &&
                     split_imgs * d->P * d->Q * g.K * sizeof(float) < 0x7FFFFFF0ull &&
                     split_imgs * d->H * d->W * d->C * sizeof(float) < 0x7FFFFFF0ull;
    dim3 grid(g.nKt * g.nCt, d->groups, g.splits), block(256);
    const bool bf16 = d->precision == MQ_PRECISION_BF16 && wgrad_vec_shape(d);
    int mode = vec ? (d->Q >= (bf16 ? 32 : WG_BK) ? 3 : (d->Q >= WG_BK ? 2 : 1)) : 0;
    const bool x3 = d->precision == MQ_PRECISION_F32X3 && wgrad_vec_shape(d);
    // dense (the Winograd-domain weight gradients): 1x1, stride 1, no padding - x is the [pixels][C] matrix itself
    if (mode > 0 && !bf16 && d->R == 1 && d->S == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 0 && d->pad_w == 0 &&
        d->P == d->H && d->Q == d->W && getenv("MQGAN_NO_DENSE") == nullptr)
        mode = 4;
    MQ_REQUIRE(!(bf16 || x3) || vec, "%s: bf16 engines need 16-byte aligned tensors below the 2 GB-per-split span", who);
    // fused bias partials: vector path only; one split writes straight into dbias
    float *bias_target = (dbias && mode > 0) ? (g.splits > 1 ? bias_ws : dbias) : nullptr;
#define MQ_WG(A_, B_, M_) do { kname("conv_wgrad_kernel<%d, %d, %d>", A_, B_, M_); hipLaunchKernelGGL((conv_wgrad_kernel<A_, B_, M_>), grid, block, 0, st, x, dy, target, bias_target, g, d->groups); } while (0)
#define MQ_WGH(A_, B_, M_) do { kname("conv_wgrad_bf16_kernel<%d, %d, %d>", A_, B_, M_); hipLaunchKernelGGL((conv_wgrad_bf16_kernel<A_, B_, M_>), grid, block, 0, st, x, dy, target, bias_target, g, d->groups); } while (0)
#define MQ_WGX(A_, B_, M_) do { kname("conv_wgrad_x3_kernel<%d, %d, %d>", A_, B_, M_); hipLaunchKernelGGL((conv_wgrad_x3_kernel<A_, B_, M_>), grid, block, 0, st, x, dy, target, bias_target, g, d->groups); } while (0)
#define MQ_WG3(A_, B_) do { if (x3) { if (mode == 4) MQ_WGX(A_, B_, 4); else if (mode == 3) MQ_WGX(A_, B_, 3); else MQ_WGX(A_, B_, 1); } \
                            else if (bf16) { if (mode == 3) MQ_WGH(A_, B_, 3); else if (mode == 2) MQ_WGH(A_, B_, 2); else MQ_WGH(A_, B_, 1); } \
                            else if (mode == 4) MQ_WG(A_, B_, 4); else if (mode == 3) MQ_WG(A_, B_, 3); else if (mode == 1) MQ_WG(A_, B_, 1); else MQ_WG(A_, B_, 0); } while (0)
#define MQ_WGF(A_, B_) do { if (mode == 4) MQ_WG(A_, B_, 4); else if (mode == 3) MQ_WG(A_, B_, 3); else if (mode == 1) MQ_WG(A_, B_, 1); else MQ_WG(A_, B_, 0); } while (0)
    if (g.bnc == 32 || g.bnc == 64) {
        MQ_REQUIRE(!bf16 && !x3, "%s: the narrow weight-gradient tiles belong to the f32 engine", who);
        if (g.bnc == 32) MQ_WGF(256, 32);
        else MQ_WGF(64, 64);
    } else if (g.bmk == 64) MQ_WG3(64, 256);
    else MQ_WG3(128, 128);
#undef MQ_WGF
#undef MQ_WG3
#undef MQ_WGX
#undef MQ_WGH
#undef MQ_WG
    MQ_LAUNCH_CHECK(who);
    if (g.splits > 1) {
        const bool v4 = n % 4 == 0 && (uintptr_t)target % 16 == 0 && (uintptr_t)dw % 16 == 0;
        if (v4) hipLaunchKernelGGL(slab_reduce_kernel<true>, dim3(ew_grid((int64_t)n / 4, 256)), dim3(256), 0, st, target, dw, n, g.splits);
        else hipLaunchKernelGGL(slab_reduce_kernel<false>, dim3(ew_grid((int64_t)n, 256)), dim3(256), 0, st, target, dw, n, g.splits);
        if (bias_target) {
            const size_t nb = (size_t)d->groups * g.K;
            hipLaunchKernelGGL(slab_reduce_kernel<false>, dim3(ew_grid((int64_t)nb, 256)), dim3(256), 0, st, bias_target, dbias,
                               nb, g.splits);
        }
        MQ_LAUNCH_CHECK(who);
    }
    if (dbias && mode == 0) {      // scalar path (K or C not a multiple of 4): stand-alone column sums per group
        for (int grp = 0; grp < d->groups; ++grp)
            if (colsum_launch(dy + (size_t)grp * g.Mtot * g.K, g.Mtot, g.K, dbias + (size_t)grp * g.K, bias_ws, st)) return 1;
    }
    return 0;
}

extern "C" int mq_conv_wgrad(const mqConvDesc *d, const float *x, const float *dy, float *dw, void *workspace,
                             size_t workspace_bytes, mqStream stream)
{
    return wgrad_impl(d, x, dy, dw, nullptr, workspace, workspace_bytes, S(stream), "mq_conv_wgrad");
}

extern "C" int mq_conv_wgrad_bias(const mqConvDesc *d, const float *x, const float *dy, float *dw, float *dbias,
                                  void *workspace, size_t workspace_bytes, mqStream stream)
{
    MQ_REQUIRE(dbias != nullptr, "mq_conv_wgrad_bias: null dbias");
    return wgrad_impl(d, x, dy, dw, dbias, workspace, workspace_bytes, S(stream), "mq_conv_wgrad_bias");
}

static int colsum_launch(const float *x, int64_t rows, int cols, float *out, float *workspace, hipStream_t st)
{
    // enough row chunks for ~2048 blocks in total, each with at least 32 rows; at most 1024 (workspace)
    const int64_t colblocks = cdiv(cols, 64);
    int64_t want = cdiv(2048, colblocks);
    if (want > rows / 32) want = rows / 32;
    if (want > 1024) want = 1024;
    if (want < 1) want = 1;
    const int nchunks = (int)want;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)cdiv(cols, 64), nchunks), dim3(64, 4), 0, st, x, rows, cols,
                       workspace, nchunks);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv(cols, 64)), dim3(64, 16), 0, st, workspace, cols, nchunks, out);
    MQ_LAUNCH_CHECK("mq_colsum");
    return 0;
}

extern "C" int mq_colsum(const float *x, int64_t rows, int cols, float *out, float *workspace, mqStream stream)
{
    MQ_REQUIRE(x && out && workspace && rows > 0 && cols > 0, "mq_colsum: bad arguments");
    return colsum_launch(x, rows, cols, out, workspace, S(stream));
}
