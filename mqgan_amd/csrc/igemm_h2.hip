// bf16-STORAGE implicit-GEMM convolution for gfx950 ("H2" kernels, MQ_PRECISION_BF16 with bf16 activations).
// Interface: include/mqgan_hip.h (mq_h2_*, mq_conv_*_h2).
//
// What differs from igemm.hip's bf16 engine (f32 tensors, operands rounded on their way into LDS):
//   * activations, output gradients and the effective weights are bf16 IN HBM (what the reference's autocast region
//     stores, train.py:220,523): half the bytes from L2 / HBM, no conversion in the kernel;
//   * block = 512 threads = 8 waves (4 along M x 2 along N), tile 256 x (64*TN) x 64: at 128 x 128 a bf16 MFMA tile
//     needs 64 B/clk/CU from L2 at full matrix rate, more than an XCD's L2 delivers - 256 x 256 needs 32;
//   * both operand tiles are staged by LDS-DMA (buffer_load_dwordx4 ... lds): no VGPR staging, no v_cvt_pk, no
//     ds_write, no LDS address arithmetic in the loop.  Padding taps, ragged tile edges and K tails are
//     "offset beyond the descriptor" - the DMA then writes zeros;
//   * LDS rows are 128 bytes (64 bf16 k's); the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7 so that the
//     16 rows a ds_read_b128 lane group touches fall on 16 distinct bank quads.  The DMA writes lane-linear, so the
//     swizzle is applied to the per-lane SOURCE address and again on the fragment reads;
//   * a ring of NSTAGE tiles (3 where LDS allows, else 2), one barrier per K-step, counted vmcnt.
// Reduction order: (channel chunk of CB, tap row, tap col, channel in chunk) with CB = 64 (C % 64 == 0: one tap per
// K-step, the whole walk on the scalar unit) or CB = 32 (two consecutive taps per K-step).  Every DMA instruction
// moves 8 pixels x 128 contiguous bytes (CB = 64) - whole cache lines.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "gemm_core.h"

namespace mq {

typedef __attribute__((address_space(3))) void lds_void;
// generic pointer into a __shared__ array -> LDS (address space 3) pointer: the low 32 bits are the LDS offset
__device__ __forceinline__ lds_void *lds_ptr(const void *p) { return reinterpret_cast<lds_void *>(reinterpret_cast<uintptr_t>(p)); }

constexpr int H2_BM = 256, H2_BK = 64, H2_ROWB = 128;

struct H2Parity {              // dgrad by stride parity (see igemm.hip): per class geometry
    int ncls;
    int a[4], b[4], Ha[4], Wb[4], tile0[5];
    int nr[4], ns[4], r0[4], s0[4], rstep[4], sstep[4];
};

struct H2Geom {
    int OH, OW;                // rows m of the GEMM enumerate (n, oh, ow)
    int IH, IW, IC;            // gathered tensor (bf16, NHWC)
    int Mtot, Ncols;
    int nMt, nNt;
    int R, S;                  // full filter (weight image layout)
    int nR, nS;                // tap grid walked (= R, S except for parity classes)
    int step_h, step_w;        // signed coordinate step per tap
    int mul_h, mul_w, add_h, add_w;   // base coordinate of a row: bh = oh * mul_h + add_h
    int CB, nchunks;           // channel block of the reduction, IC / CB
    int Kp;                    // weight row pitch in elements (multiple of 64)
    int sh, sw, ph, pw, dh, dw;
    int ng;                    // images per group
    // conv_h2s_kernel on ONE stride-parity class of a strided input gradient (launch_h2): the taps walked are r0 + jr * rstep,
    // s0 + js * sstep of the full filter, and row (n, oh, ow) of the class is pixel (oh * osh + oa, ow * osw + ob) of the full
    // OHf x OWf output.  A plain launch has 0 / 1 / 0 / 1, 1 / 1 / 0 / 0 and OHf = OH, OWf = OW.
    int tap_r0, tap_rstep, tap_s0, tap_sstep;
    int osh, osw, oa, ob, OHf, OWf;
};

struct H2Epi {
    const float *bias;         // [groups * Ncols] or null
    const uint8_t *mask;       // frame mask (1 = padded) or null; index of output row m: (m / d1) * d2 + m % d3
    int d1, d2, d3;
    int act;                   // 0: none, 2: LeakyReLU(slope) then masked_fill, 3 (dgrad): LeakyReLU'(gate), masked
    float slope;
    int out_f32;               // 1: f32 output, 0: bf16
    const unsigned short *gate;   // act 3: the conv's bf16 input x = padded ? 0 : leaky(h), laid out like the output dx;
                                  // the kernel returns the gradient with respect to h: dx * (x > 0 ? 1 : slope), masked
};

__device__ __forceinline__ unsigned short f2bf(float v)
{
    return __builtin_bit_cast(unsigned short, (__bf16)v);          // v_cvt_pk_bf16_f32 (RNE, NaN-safe)
}

// Element policies.  The same structure serves the exact-product f32 engine (MQ_PRECISION_F32: f32 tensors,
// v_mfma_f32_32x32x2_f32): a K-step is then 32 k's (the same 128-byte LDS rows), each lane half reads one float2 per
// k-quad = the operands of two MFMA steps, and every accumulator sums k in the order 4q, 4q+2, 4q+1, 4q+3 - the order of
// igemm.hip's f32 engine, so the two are bit-identical.  There the gain is not bandwidth but issue slots: f32 MFMA runs
// at the vector rate and every other instruction a wave issues comes out of the matrix pipe's time; LDS-DMA removes the
// staging loads, the LDS stores and their address arithmetic from the loop.
struct ElemBF16 { static constexpr int ESZ = 2; static constexpr bool F32 = false; };
struct ElemF32 { static constexpr int ESZ = 4; static constexpr bool F32 = true; };

// ---- epilogue of the H2 kernels: bias, activation / gate, store (f32 straight from the accumulators, bf16 transposed through a
// per-wave LDS patch).  row_ok(m): output row m of the tile's enumeration exists;  out_index(m): element offset of its column 0.
// W16: 16-column tiles per wave (the wave tile is 16 * W16 columns wide; 2 * TN except in the 96-column kernel, where TN is unused)
// NW: waves of the block (two per 64 rows of the tile: 8 for the 256-row kernels, 4 for conv_h2_small_kernel)
template <class ET, int TN, bool MF16, int W16, int NW, class RowOk, class OutIdx>
__device__ __forceinline__ void h2_epilogue(f32x16 (&acc)[MF16 ? 1 : 2][MF16 ? 1 : TN], f32x4v (&acc16)[MF16 ? 4 : 1][MF16 ? W16 : 1],
                                            const H2Geom &g, const H2Epi &epi, void *__restrict__ out, char *smem, int m0, int n0, int grp,
                                            RowOk &&row_ok, OutIdx &&out_index)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    // ---- epilogue.  LDS is free once every wave's DMAs have landed and its fragment reads are back.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    static_assert(MF16 || W16 == 2 * TN, "wave tile width");
    constexpr int STG_ROWB = 32 * W16;                            // bytes of one staged bf16 row: 16 * W16 columns
    constexpr int STG_WAVE = 32 * STG_ROWB;                       // one wave stages 32 rows (one tm) at a time
    unsigned char *padf = reinterpret_cast<unsigned char *>(smem) + NW * STG_WAVE;
    const bool fused = epi.act == 2, gated = epi.act == 3;
    if (fused || gated) {
        if (tid < 32 * NW) {
            const int m = m0 + tid;
            unsigned char pd = 0;
            if (epi.mask != nullptr && row_ok(m)) {
                const int64_t mg = (int64_t)(out_index(m) / (size_t)g.Ncols);    // output pixel, counted over all groups
                pd = epi.mask[(mg / epi.d1) * epi.d2 + (mg % epi.d3)] != 0;
            }
            padf[tid] = pd;
        }
        __syncthreads();
    }
    auto bf2f = [](unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); };
    // visit the accumulator elements of one 32-row half of the wave tile: fn(row in half, column in wave tile, value)
    auto for_half = [&](auto half_tag, auto &&fn) {
        constexpr int half = decltype(half_tag)::value;
        if constexpr (MF16) {
#pragma unroll
            for (int th = 0; th < 2; ++th)
#pragma unroll
                for (int tn = 0; tn < W16; ++tn)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        fn(th * 16 + (lane >> 4) * 4 + j, tn * 16 + (lane & 15), acc16[2 * half + th][tn][j]);
        } else {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) fn((r & 3) + 8 * (r >> 2) + 4 * lh, tn * 32 + li, acc[half][tn][r]);
        }
    };
    // bias of this lane's columns: column index inside the wave tile -> slot
    constexpr int NBV = MF16 ? W16 : TN;
    float bv[NBV];
#pragma unroll
    for (int tn = 0; tn < NBV; ++tn) {
        const int n = n0 + wn * 16 * W16 + (MF16 ? tn * 16 + (lane & 15) : tn * 32 + li);
        bv[tn] = (epi.bias && n < g.Ncols) ? epi.bias[(size_t)grp * g.Ncols + n] : 0.f;
    }
    auto bias_of = [&](int cl) { return bv[MF16 ? cl >> 4 : cl >> 5]; };
    if (ET::F32 || epi.out_f32) {
        // f32 output: a lane-group's 32 columns are 128 contiguous bytes per row - stored straight from the accumulators
        float *of = reinterpret_cast<float *>(out);
        auto store_half = [&](auto half_tag) {
            constexpr int half = decltype(half_tag)::value;
            for_half(half_tag, [&](int rl, int cl, float a) {
                const int row = wm * 64 + half * 32 + rl;
                const int m = m0 + row, n = n0 + wn * 16 * W16 + cl;
                if (row_ok(m) && n < g.Ncols) {
                    float v = a + bias_of(cl);
                    if (fused) {
                        v = v > 0.f ? v : epi.slope * v;
                        v = padf[row] ? 0.f : v;
                    }
                    if (gated) {
                        v = bf2f(epi.gate[out_index(m) + n]) > 0.f ? v : epi.slope * v;
                        v = padf[row] ? 0.f : v;
                    }
                    of[out_index(m) + n] = v;
                }
            });
        };
        store_half(std::integral_constant<int, 0>{});
        store_half(std::integral_constant<int, 1>{});
        return;
    }
    // bf16 output: 2-byte stores straight from the MFMA layout (one column per lane) were measured SLOWER than f32 output
    // with twice the bytes.  Each wave transposes its tile through a private LDS patch instead - written element-wise,
    // read back row-wise - and stores 16 bytes per lane: 4 * TN lanes cover the 64 * TN contiguous bytes of a row.
    unsigned short *oh_ = reinterpret_cast<unsigned short *>(out);
    char *stg = smem + wave * STG_WAVE;
    auto stage_half = [&](auto half_tag) {
        constexpr int half = decltype(half_tag)::value;
        constexpr int CPR = 2 * W16;                                  // 16-byte chunks per row
        constexpr int NIT = (32 * CPR + 63) / 64;
        // gated store: this lane's chunks of the gate tensor are requested before the tile is staged, so their latency
        // hides behind the LDS transpose instead of sitting in front of every store
        uint4 gq[NIT];
        if (gated) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int q = i * 64 + lane;
                const int rl = q / CPR, c16 = q - rl * CPR;
                const int m = m0 + wm * 64 + half * 32 + rl;
                const int n = n0 + wn * 16 * W16 + c16 * 8;
                gq[i] = make_uint4(0, 0, 0, 0);
                if (rl < 32 && row_ok(m) && n < g.Ncols) gq[i] = *reinterpret_cast<const uint4 *>(epi.gate + out_index(m) + n);
            }
        }
        for_half(half_tag, [&](int rl, int cl, float a) {
            float v = a + bias_of(cl);
            if (fused) {
                v = v > 0.f ? v : epi.slope * v;
                v = padf[wm * 64 + half * 32 + rl] ? 0.f : v;
            }
            *reinterpret_cast<unsigned short *>(stg + rl * STG_ROWB + cl * 2) = f2bf(v);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // own writes (the patch is private to the wave)
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int q = i * 64 + lane;
            const int rl = q / CPR, c16 = q - rl * CPR;
            const int m = m0 + wm * 64 + half * 32 + rl;
            const int n = n0 + wn * 16 * W16 + c16 * 8;
            if (rl < 32 && row_ok(m) && n < g.Ncols) {
                uint4 v = *reinterpret_cast<const uint4 *>(stg + rl * STG_ROWB + c16 * 16);
                if (gated) {
                    // the staged values are the bf16 gradient an ungated launch would have stored; gate them as the
                    // stand-alone pointwise backward does (f32 product, one more rounding): bit-identical to the two passes
                    const bool pd = padf[wm * 64 + half * 32 + rl] != 0;
                    const unsigned vw[4] = {v.x, v.y, v.z, v.w}, gw[4] = {gq[i].x, gq[i].y, gq[i].z, gq[i].w};
                    unsigned ow[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo = bf2f((unsigned short)(vw[e] & 0xFFFFu)), hi = bf2f((unsigned short)(vw[e] >> 16));
                        lo = bf2f((unsigned short)(gw[e] & 0xFFFFu)) > 0.f ? lo : epi.slope * lo;
                        hi = bf2f((unsigned short)(gw[e] >> 16)) > 0.f ? hi : epi.slope * hi;
                        lo = pd ? 0.f : lo;
                        hi = pd ? 0.f : hi;
                        ow[e] = (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
                    }
                    v = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                }
                *reinterpret_cast<uint4 *>(oh_ + out_index(m) + n) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // reads done before the second half overwrites the patch
    };
    stage_half(std::integral_constant<int, 0>{});
    stage_half(std::integral_constant<int, 1>{});
}

// TN: 32-column MFMA tiles per wave along N (block tile N = 64 * TN); CBK: channel block in ELEMENTS (bf16: 64 or 32,
// f32: 16, i.e. 128- or 64-byte runs); PARITY: dgrad by stride parity (tap subset and output scatter per class)
template <class ET, int TN, int NSTAGE, int CBK, bool PARITY, bool MF16 = false>
__device__ __forceinline__ void conv_h2_body(const void *__restrict__ act_v, const void *__restrict__ wgt_v,
                                             void *__restrict__ out, const H2Geom &g, const H2Parity &pi, const H2Epi &epi,
                                             size_t act_group_stride, char *smem)
{
    constexpr int ESZ = ET::ESZ;
    constexpr int BN = 64 * TN;
    constexpr int STAGE = (H2_BM + BN) * H2_ROWB;
    constexpr int NSUB = 128 / (CBK * ESZ);        // (chunk, tap) positions per K-step
    constexpr int CPB = CBK * ESZ / 16;            // 16-byte chunks per position
    const char *act = reinterpret_cast<const char *>(act_v);
    const char *wgt = reinterpret_cast<const char *>(wgt_v);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int grp = blockIdx.y;

    // ---- tile of this block
    int m0, n0, Mc, OWc, OHc, nR, nS, step_h, step_w, tap_r0 = 0, tap_s0 = 0, tap_rstep = 1, tap_sstep = 1;
    int cls_a = 0, cls_b = 0;
    if constexpr (PARITY) {
        const int total_mt = pi.tile0[pi.ncls];
        const int tile = xcd_pair_remap(blockIdx.x, total_mt * g.nNt, g.nNt);
        const int mtg = tile / g.nNt, nt = tile - mtg * g.nNt;
        int cls = 0;
#pragma unroll
        for (int c = 1; c < 4; ++c)
            if (c < pi.ncls && mtg >= pi.tile0[c]) cls = c;
        m0 = (mtg - pi.tile0[cls]) * H2_BM;
        n0 = nt * BN;
        OHc = pi.Ha[cls]; OWc = pi.Wb[cls];
        Mc = g.ng * OHc * OWc;
        nR = pi.nr[cls]; nS = pi.ns[cls];
        tap_r0 = pi.r0[cls]; tap_s0 = pi.s0[cls]; tap_rstep = pi.rstep[cls]; tap_sstep = pi.sstep[cls];
        cls_a = pi.a[cls]; cls_b = pi.b[cls];
        const int shift_h = g.sh == 2 ? 1 : 0, shift_w = g.sw == 2 ? 1 : 0;
        step_h = -((tap_rstep * g.dh) >> shift_h);      // exact: the class's taps keep (ih + ph - r*dh) divisible by sh
        step_w = -((tap_sstep * g.dw) >> shift_w);
    } else {
        const int tile = xcd_remap(blockIdx.x, g.nMt * g.nNt);
        const int mt = tile / g.nNt, nt = tile - mt * g.nNt;
        m0 = mt * H2_BM;
        n0 = nt * BN;
        OHc = g.OH; OWc = g.OW; Mc = g.Mtot;
        nR = g.nR; nS = g.nS; step_h = g.step_h; step_w = g.step_w;
    }
    nR = __builtin_amdgcn_readfirstlane(nR); nS = __builtin_amdgcn_readfirstlane(nS);
    step_h = __builtin_amdgcn_readfirstlane(step_h); step_w = __builtin_amdgcn_readfirstlane(step_w);

    // ---- DMA slots.  Piece p = wave + 8 j covers tile rows 8p .. 8p+7; lane -> row 8p + (lane >> 3), physical
    // chunk lane & 7; the logical chunk it must fetch undoes the read-side swizzle ((row >> 1) & 7 = (4*wave + (lane >> 4)) & 7).
    const int lrow = lane >> 3;
    const int lc = (lane & 7) ^ ((((wave & 1) << 2) | (lane >> 4)) & 7);
    const int sub = lc / CPB;                                  // which of the step's positions this lane loads
    const int cin_b = (lc % CPB) * 16;                         // byte offset inside the position's channel run
    const int img0 = (OHc * OWc) > 0 ? m0 / (OHc * OWc) : 0;
    const size_t img_elems = (size_t)g.IH * g.IW * g.IC;
    const char *abase = act + ((size_t)grp * act_group_stride + (size_t)img0 * img_elems) * ESZ;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(abase, (act_group_stride - (size_t)img0 * img_elems) * ESZ);
    const char *wbase = wgt + ((size_t)grp * g.Ncols + n0) * g.Kp * ESZ;
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(wbase, (size_t)(g.Ncols - n0 < BN ? g.Ncols - n0 : BN) * g.Kp * ESZ);
    int bh_[4], bw_[4], rowbase_[4];
    bool rv_[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + 8 * (wave + 8 * j) + lrow;
        const bool v = m < Mc;
        const int mm = v ? m : m0;
        const int ow = mm % OWc, t = mm / OWc;
        const int oh = t % OHc, n = t / OHc - img0;
        int bh, bw;
        if constexpr (PARITY) {
            const int shift_h = g.sh == 2 ? 1 : 0, shift_w = g.sw == 2 ? 1 : 0;
            bh = (oh * g.sh + cls_a + g.ph - tap_r0 * g.dh) >> shift_h;
            bw = (ow * g.sw + cls_b + g.pw - tap_s0 * g.dw) >> shift_w;
        } else {
            bh = oh * g.mul_h + g.add_h;
            bw = ow * g.mul_w + g.add_w;
        }
        rv_[j] = v;
        bh_[j] = bh;
        bw_[j] = bw;
        rowbase_[j] = (((n * g.IH + bh) * g.IW + bw) * g.IC) * ESZ + cin_b;
    }
    unsigned rowoffB[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = 8 * (wave + 8 * j) + lrow;
        rowoffB[j] = (n0 + row < g.Ncols) ? (unsigned)row * (unsigned)g.Kp * (unsigned)ESZ + (unsigned)cin_b : OOB;
    }
    // ---- scalar walk over (chunk, tap row, tap col): NSUB walkers, walker u starts at position u, moves NSUB per step
    int w_js[NSUB], w_jr[NSUB], w_ch[NSUB];
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
        int js = u, jr = 0, ch = (nR > 0 && nS > 0) ? 0 : g.nchunks;
        if (nS > 0) { while (js >= nS) { js -= nS; ++jr; } }
        if (nR > 0) { while (jr >= nR) { jr -= nR; ++ch; } }
        w_js[u] = js; w_jr[u] = jr; w_ch[u] = ch;
    }
    const int RS = g.R * g.S;
    // One tile's DMA work in pieces, so that the 16x16x32 loop can put a piece after every row of MFMAs instead of a
    // hundred scalar / vector instructions in front of them: walk() advances the scalar tap walkers and leaves the tile's
    // offsets in (hoff, woff, aoff, bcol, kv); dma_a(stage, j) / dma_b(stage, j) issue one 8 KB piece each.
    int hoff = 0, woff = 0, aoff = 0, bcol = 0;
    bool kv = false;
    auto walk = [&]() {
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            const int h_u = w_jr[u] * step_h, w_u = w_js[u] * step_w;
            const int a_u = ((h_u * g.IW + w_u) * g.IC + w_ch[u] * CBK) * ESZ;
            const int tr = PARITY ? tap_r0 + w_jr[u] * tap_rstep : w_jr[u];
            const int ts = PARITY ? tap_s0 + w_js[u] * tap_sstep : w_js[u];
            const int b_u = ((w_ch[u] * RS + tr * g.S + ts) * CBK) * ESZ;
            const bool k_u = w_ch[u] < g.nchunks;
            if (u == 0 || sub == u) { hoff = h_u; woff = w_u; aoff = a_u; bcol = b_u; kv = k_u; }
            // advance this walker by NSUB positions (scalar unit)
#pragma unroll
            for (int i = 0; i < NSUB; ++i) {
                ++w_js[u];
                const bool c2 = w_js[u] >= nS;
                w_js[u] = c2 ? 0 : w_js[u];
                w_jr[u] += c2 ? 1 : 0;
                const bool c3 = w_jr[u] >= nR;
                w_jr[u] = c3 ? 0 : w_jr[u];
                w_ch[u] += c3 ? 1 : 0;
            }
        }
    };
    auto dma_a = [&](int stage, int j) {
#ifdef MQ_H2_ABLATE
        if ((MQ_H2_ABLATE & 8) && j > 0) return;
#endif
        const bool ok = rv_[j] & kv & ((unsigned)(bh_[j] + hoff) < (unsigned)g.IH) & ((unsigned)(bw_[j] + woff) < (unsigned)g.IW);
        const unsigned vo = ok ? (unsigned)(rowbase_[j] + aoff) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, lds_ptr(smem + stage * STAGE + wave * 1024 + j * 8192), 16, vo, 0, 0, 0);
    };
    auto dma_b = [&](int stage, int j) {
#ifdef MQ_H2_ABLATE
        if ((MQ_H2_ABLATE & 4) && j > 0) return;
#endif
        const unsigned vo = ((rowoffB[j] != OOB) & kv) ? rowoffB[j] + (unsigned)bcol : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, lds_ptr(smem + stage * STAGE + H2_BM * H2_ROWB + wave * 1024 + j * 8192), 16,
                                                 vo, 0, 0, 0);
    };
    auto issue = [&](int stage) {
        walk();
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_a(stage, j);
#pragma unroll
        for (int j = 0; j < TN; ++j) dma_b(stage, j);
    };

    // accumulators: 2 x TN tiles of 32x32 (16 registers each) or, MF16, 4 x 2TN tiles of 16x16 (4 registers each)
    f32x16 acc[MF16 ? 1 : 2][MF16 ? 1 : TN];
    f32x4v acc16[MF16 ? 4 : 1][MF16 ? 2 * TN : 1];
#pragma unroll
    for (int a = 0; a < (MF16 ? 1 : 2); ++a)
#pragma unroll
        for (int b = 0; b < (MF16 ? 1 : TN); ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
    for (int a = 0; a < (MF16 ? 4 : 1); ++a)
#pragma unroll
        for (int b = 0; b < (MF16 ? 2 * TN : 1); ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[a][b][r] = 0.f;

    // Fragment reads.  bf16: lane (r, h) reads the 16 bytes of chunk 2s + h (8 k's) of row r per 32x32x16 MFMA.
    // f32: lane (r, h) reads the float2 at byte 8h of chunk q (k = 4q + 2h, 4q + 2h + 1) per pair of 32x32x2 MFMAs.
    // Either way the chunk index is XORed with (row >> 1) & 7, which depends on the lane only.
    constexpr int NFR = ET::F32 ? 8 : 4;
    const int sw = (li >> 1) & 7;
    int aoffs[NFR], boffs[NFR];
#pragma unroll
    for (int s = 0; s < NFR; ++s) {
        const int chunk = ET::F32 ? (s ^ sw) : ((2 * s) ^ (lh ^ sw));
        const int inner = ET::F32 ? 8 * lh : 0;
        aoffs[s] = (wm * 64 + li) * H2_ROWB + (chunk << 4) + inner;
        boffs[s] = H2_BM * H2_ROWB + (wn * 32 * TN + li) * H2_ROWB + (chunk << 4) + inner;
    }
    const int npos = g.nchunks * nR * nS;
    const int nk = (npos + NSUB - 1) / NSUB;
    constexpr int LPT = 4 + TN;                                   // DMA instructions per wave and tile
#pragma unroll
    for (int p = 0; p < NSTAGE - 1; ++p) issue(p);
    if constexpr (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
    __builtin_amdgcn_s_barrier();

    if constexpr (MF16) {
        // v_mfma_f32_16x16x32_bf16: the same bytes from LDS per flop and the same cycles per flop as 32x32x16, but the chip
        // holds a higher clock on this shape under load (MI355X_MICROARCH.md, DVFS give-back item 7).  Lane (r = l & 15,
        // g = l >> 4) holds k = 8g .. 8g+7 of a 32-deep substep: chunk 4s + g of row r, swizzled with (r >> 1) & 7.
        const int r16 = lane & 15, g4 = lane >> 4, sw16 = (r16 >> 1) & 7;
        int a16[2], b16[2];
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            a16[sx] = (wm * 64 + r16) * H2_ROWB + (((4 * sx + g4) ^ sw16) << 4);
            b16[sx] = H2_BM * H2_ROWB + (wn * 32 * TN + r16) * H2_ROWB + (((4 * sx + g4) ^ sw16) << 4);
        }
        // Four phases per K-step (substep x column half), two register sets: the fragments of the next phase are requested
        // before the 4 x TN MFMAs of the current one, and the K-step's barrier sits in front of the last phase, whose MFMAs
        // cover the first reads of the next tile.  The DMA work of the next ring slot is spread over the MFMA rows of the
        // phase behind the barrier (walker + activation pieces) and of the following phase (weight pieces): both waves of a
        // SIMD leave the barrier together, so a block of scalar work in front of the MFMAs would idle the pipe for both.
        bf16x8 fa[2][4], fb[2][TN];
        // MQ_H2_ABLATE (tools/h2_ablate.sh; never in the shipped library): WRONG results on purpose, to see which traffic the loop's
        // time follows - bit 0: a quarter of the A fragment reads, bit 1: 1 / TN of the B fragment reads, bit 2: one B DMA piece per
        // tile, bit 3: one A DMA piece per tile; in conv_wgrad_h2_body bit 4: one fragment read per operand and substep, bit 5: a
        // quarter of the dy DMA, bit 6: a quarter of the x DMA
#ifndef MQ_H2_ABLATE
#define MQ_H2_ABLATE 0
#endif
        auto ld_a = [&](const char *base, int sx, auto set_tag) {
            constexpr int set = decltype(set_tag)::value;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                if ((MQ_H2_ABLATE & 1) && tt > 0) { fa[set][tt] = fa[set][0]; continue; }
                fa[set][tt] = *reinterpret_cast<const bf16x8 *>(base + a16[sx] + tt * 16 * H2_ROWB);
            }
        };
        auto ld_b = [&](const char *base, int sx, int h, auto set_tag) {
            constexpr int set = decltype(set_tag)::value;
#pragma unroll
            for (int tt = 0; tt < TN; ++tt) {
                if ((MQ_H2_ABLATE & 2) && tt > 0) { fb[set][tt] = fb[set][0]; continue; }
                fb[set][tt] = *reinterpret_cast<const bf16x8 *>(base + b16[sx] + (h * TN + tt) * 16 * H2_ROWB);
            }
        };
        auto mm_row = [&](auto sa_tag, auto sb_tag, auto h_tag, auto tm_tag) {
            constexpr int sa = decltype(sa_tag)::value, sb = decltype(sb_tag)::value, h = decltype(h_tag)::value;
            constexpr int tm = decltype(tm_tag)::value;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc16[tm][h * TN + tn] =
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[sa][tm], fb[sb][tn], acc16[tm][h * TN + tn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        auto mm = [&](auto sa_tag, auto sb_tag, auto h_tag) {
            mm_row(sa_tag, sb_tag, h_tag, I0{});
            mm_row(sa_tag, sb_tag, h_tag, I1{});
            mm_row(sa_tag, sb_tag, h_tag, I2{});
            mm_row(sa_tag, sb_tag, h_tag, I3{});
        };
        int st_c = 0, st_i = NSTAGE - 1;
        walk();
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_a(st_i, j);
        ld_a(smem, 0, I0{});
        ld_b(smem, 0, 0, I0{});
        for (int t = 0; t < nk; ++t) {
            const char *base = smem + st_c * STAGE;
            ld_b(base, 0, 1, I1{});
            __builtin_amdgcn_sched_barrier(0);
            mm_row(I0{}, I0{}, I0{}, I0{});
            dma_b(st_i, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm_row(I0{}, I0{}, I0{}, I1{});
            if constexpr (TN > 1) dma_b(st_i, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm_row(I0{}, I0{}, I0{}, I2{});
            if constexpr (TN > 2) dma_b(st_i, 2);
            __builtin_amdgcn_sched_barrier(0);
            mm_row(I0{}, I0{}, I0{}, I3{});
            if constexpr (TN > 3) dma_b(st_i, 3);
            __builtin_amdgcn_sched_barrier(0);
            ld_a(base, 1, I1{});
            ld_b(base, 1, 0, I0{});
            __builtin_amdgcn_sched_barrier(0);
            mm(I0{}, I1{}, I1{});
            ld_b(base, 1, 1, I1{});
            __builtin_amdgcn_sched_barrier(0);
            mm(I1{}, I0{}, I0{});
            // ring slot t + 1 landed (its last piece was issued three phases ago); every fragment read of slot t is back
            if constexpr (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPT) : "memory");   // slot t + 2 stays in flight
            __builtin_amdgcn_s_barrier();
            st_c = st_c + 1 == NSTAGE ? 0 : st_c + 1;
            st_i = st_i + 1 == NSTAGE ? 0 : st_i + 1;
            const char *nbase = smem + st_c * STAGE;
            ld_a(nbase, 0, I0{});
            ld_b(nbase, 0, 0, I0{});
            __builtin_amdgcn_sched_barrier(0);
            mm_row(I1{}, I1{}, I1{}, I0{});
            walk();
            dma_a(st_i, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm_row(I1{}, I1{}, I1{}, I1{});
            dma_a(st_i, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm_row(I1{}, I1{}, I1{}, I2{});
            dma_a(st_i, 2);
            __builtin_amdgcn_sched_barrier(0);
            mm_row(I1{}, I1{}, I1{}, I3{});
            dma_a(st_i, 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the last iteration's activation pieces (all out of range: zeros) still write LDS, which the epilogue reuses
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    } else {
        // Fragment pipeline: two register sets.  The fragments of substep s + 1 are requested BEFORE the MFMAs of substep s are
        // issued, so an LDS read has a whole substep (8 x 32 cycles of bf16 MFMAs, 16 x 64 of f32 ones) to come back; the
        // K-step's barrier sits in front of the LAST substep's MFMAs, which then cover the first fragment reads of the next
        // tile and the DMA issue at the top of the next iteration.  (r2 counters before this: waves parked at waitcnt / barrier
        // 37 % of their cycles with the reads issued two MFMAs ahead of their use.)
        using Frag = typename std::conditional<ET::F32, float2, bf16x8>::type;
        Frag fa[2][2], fb[2][TN];
        auto load_frags = [&](int stage, auto s_tag, auto set_tag) {
            constexpr int sx = decltype(s_tag)::value, set = decltype(set_tag)::value;
            const char *base = smem + stage * STAGE;
    #pragma unroll
            for (int t = 0; t < 2; ++t) fa[set][t] = *reinterpret_cast<const Frag *>(base + aoffs[sx] + t * 32 * H2_ROWB);
    #pragma unroll
            for (int t = 0; t < TN; ++t) fb[set][t] = *reinterpret_cast<const Frag *>(base + boffs[sx] + t * 32 * H2_ROWB);
        };
        auto mfmas = [&](auto set_tag) {
            constexpr int set = decltype(set_tag)::value;
            if constexpr (ET::F32) {
    #pragma unroll
                for (int tm = 0; tm < 2; ++tm)
    #pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][tm].x, fb[set][tn].x, acc[tm][tn], 0, 0, 0);
    #pragma unroll
                for (int tm = 0; tm < 2; ++tm)
    #pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][tm].y, fb[set][tn].y, acc[tm][tn], 0, 0, 0);
            } else {
    #pragma unroll
                for (int tm = 0; tm < 2; ++tm)
    #pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][tm], fb[set][tn], acc[tm][tn], 0, 0, 0);
            }
        };
        // Tile t + NSTAGE - 1 goes into the stage that was read in iteration t - 1: every wave has passed the barrier that
        // ended that iteration.  Past the end of K the walkers sit beyond the last chunk and the DMA writes zeros.
        int st_c = 0, st_i = NSTAGE - 1;
        load_frags(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        for (int t = 0; t < nk; ++t) {
            const int st_n = st_c + 1 == NSTAGE ? 0 : st_c + 1;
            auto substep = [&](auto s_tag) {
                constexpr int sx = decltype(s_tag)::value;
                if constexpr (sx + 1 < NFR) {
                    load_frags(st_c, std::integral_constant<int, sx + 1>{}, std::integral_constant<int, (sx + 1) & 1>{});
                } else {
                    // every fragment of this tile is in registers or on its way: retire the reads (the stage may then be
                    // overwritten), make sure the next tile has landed, meet the other waves, and start on the next tile
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if constexpr (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");   // tile t + 1 landed, t + 2 stays in flight
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    load_frags(st_n, std::integral_constant<int, 0>{}, std::integral_constant<int, (sx + 1) & 1>{});
                }
                __builtin_amdgcn_sched_barrier(0);
                mfmas(std::integral_constant<int, sx & 1>{});
                __builtin_amdgcn_sched_barrier(0);
            };
            substep(std::integral_constant<int, 0>{});
            issue(st_i);                     // behind the first MFMA group: the DMA's address arithmetic rides in its shadow
            __builtin_amdgcn_sched_barrier(0);
            substep(std::integral_constant<int, 1>{});
            substep(std::integral_constant<int, 2>{});
            substep(std::integral_constant<int, 3>{});
            if constexpr (NFR == 8) {
                substep(std::integral_constant<int, 4>{});
                substep(std::integral_constant<int, 5>{});
                substep(std::integral_constant<int, 6>{});
                substep(std::integral_constant<int, 7>{});
            }
            st_c = st_n;
            st_i = st_i + 1 == NSTAGE ? 0 : st_i + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // trailing (zero-filling) DMAs must not outlive the block's LDS

    auto out_index = [&](int m) -> size_t {                       // element offset of column 0 of output row m
        if constexpr (PARITY) {
            const int iwq = m % OWc, t2 = m / OWc;
            const int ihq = t2 % OHc, nimg = t2 / OHc;
            return ((((size_t)grp * g.ng + nimg) * g.OH + ihq * g.sh + cls_a) * g.OW + iwq * g.sw + cls_b) * g.Ncols;
        } else {
            return ((size_t)grp * Mc + m) * g.Ncols;
        }
    };
    h2_epilogue<ET, TN, MF16, 2 * TN, 8>(acc, acc16, g, epi, out, smem, m0, n0, grp, [&](int m) { return m < Mc; }, out_index);
}

// ------------------------------------------------------------------------------------------ short reductions (r5)
// The 1x1 layers at the edges of the networks (32 -> 128, 64 -> 384, 96 -> 32 channels ...) have one to four K-steps.  On
// conv_h2_kernel a block is then a prologue, a handful of MFMAs and a 64-128 KB store, one after the other, with ONE block per CU
// (two or three stages of a 256-row tile fill the LDS): 2.1 TB/s of a traffic that is x once + y once.  This kernel gives up the
// ring - the K-steps it has are too few to fill one - and the latency hiding moves to the blocks: 128-row tiles, four waves, one
// stage of (128 + 64 TN) rows = 24-48 KB, three blocks per CU at different points of their lives.  Same K order, same MFMAs and
// the same epilogue as conv_h2_kernel: bit-identical (test_h2_small_kernel_is_bit_identical).
constexpr int H2Q_BM = 128;

template <int TN, int CBK>
__device__ __forceinline__ void conv_h2_small_body(const void *__restrict__ act_v, const void *__restrict__ wgt_v, void *__restrict__ out,
                                                   const H2Geom &g, const H2Epi &epi, size_t act_group_stride, char *smem)
{
    constexpr int BN = 64 * TN;
    constexpr int NSUB = 128 / (CBK * 2);          // (chunk, tap) positions per K-step
    constexpr int CPB = CBK * 2 / 16;              // 16-byte chunks per position
    const char *act = reinterpret_cast<const char *>(act_v);
    const char *wgt = reinterpret_cast<const char *>(wgt_v);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int grp = blockIdx.y;
    const int tile = xcd_remap(blockIdx.x, g.nMt * g.nNt);
    const int mt = tile / g.nNt, nt = tile - mt * g.nNt;
    const int m0 = mt * H2Q_BM, n0 = nt * BN;
    const int Mc = g.Mtot;
    const int nR = __builtin_amdgcn_readfirstlane(g.nR), nS = __builtin_amdgcn_readfirstlane(g.nS);
    const int step_h = __builtin_amdgcn_readfirstlane(g.step_h), step_w = __builtin_amdgcn_readfirstlane(g.step_w);

    // ---- DMA slots (conv_h2_body with four waves: piece p = wave + 4 j covers tile rows 8 p .. 8 p + 7)
    const int lrow = lane >> 3;
    const int lc = (lane & 7) ^ ((((wave & 1) << 2) | (lane >> 4)) & 7);
    const int sub = lc / CPB;
    const int cin_b = (lc % CPB) * 16;
    const int img0 = m0 / (g.OH * g.OW);
    const size_t img_elems = (size_t)g.IH * g.IW * g.IC;
    const char *abase = act + ((size_t)grp * act_group_stride + (size_t)img0 * img_elems) * 2;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(abase, (act_group_stride - (size_t)img0 * img_elems) * 2);
    const char *wbase = wgt + ((size_t)grp * g.Ncols + n0) * g.Kp * 2;
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(wbase, (size_t)(g.Ncols - n0 < BN ? g.Ncols - n0 : BN) * g.Kp * 2);
    int bh_[4], bw_[4], rowbase_[4];
    bool rv_[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + 8 * (wave + 4 * j) + lrow;
        const bool v = m < Mc;
        const int mm = v ? m : m0;
        const int ow = mm % g.OW, t = mm / g.OW;
        const int oh = t % g.OH, n = t / g.OH - img0;
        rv_[j] = v;
        bh_[j] = oh * g.mul_h + g.add_h;
        bw_[j] = ow * g.mul_w + g.add_w;
        rowbase_[j] = (((n * g.IH + bh_[j]) * g.IW + bw_[j]) * g.IC) * 2 + cin_b;
    }
    unsigned rowoffB[2 * TN];
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) {
        const int row = 8 * (wave + 4 * j) + lrow;
        rowoffB[j] = (n0 + row < g.Ncols) ? (unsigned)row * (unsigned)g.Kp * 2u + (unsigned)cin_b : OOB;
    }
    int w_js[NSUB], w_jr[NSUB], w_ch[NSUB];
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
        int js = u, jr = 0, ch = (nR > 0 && nS > 0) ? 0 : g.nchunks;
        if (nS > 0) { while (js >= nS) { js -= nS; ++jr; } }
        if (nR > 0) { while (jr >= nR) { jr -= nR; ++ch; } }
        w_js[u] = js; w_jr[u] = jr; w_ch[u] = ch;
    }
    const int RS = g.R * g.S;
    int hoff = 0, woff = 0, aoff = 0, bcol = 0;
    bool kv = false;
    auto walk = [&]() {
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            const int h_u = w_jr[u] * step_h, w_u = w_js[u] * step_w;
            const int a_u = ((h_u * g.IW + w_u) * g.IC + w_ch[u] * CBK) * 2;
            const int b_u = ((w_ch[u] * RS + w_jr[u] * g.S + w_js[u]) * CBK) * 2;
            const bool k_u = w_ch[u] < g.nchunks;
            if (u == 0 || sub == u) { hoff = h_u; woff = w_u; aoff = a_u; bcol = b_u; kv = k_u; }
#pragma unroll
            for (int i = 0; i < NSUB; ++i) {
                ++w_js[u];
                const bool c2 = w_js[u] >= nS;
                w_js[u] = c2 ? 0 : w_js[u];
                w_jr[u] += c2 ? 1 : 0;
                const bool c3 = w_jr[u] >= nR;
                w_jr[u] = c3 ? 0 : w_jr[u];
                w_ch[u] += c3 ? 1 : 0;
            }
        }
    };
    f32x16 acc_unused[1][1];
    f32x4v acc16[4][2 * TN];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2 * TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[a][b][r] = 0.f;
    const int r16 = lane & 15, g4 = lane >> 4, sw16 = (r16 >> 1) & 7;
    int a16[2], b16[2];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
        a16[sx] = (wm * 64 + r16) * H2_ROWB + (((4 * sx + g4) ^ sw16) << 4);
        b16[sx] = H2Q_BM * H2_ROWB + (wn * 32 * TN + r16) * H2_ROWB + (((4 * sx + g4) ^ sw16) << 4);
    }
    const int npos = g.nchunks * nR * nS;
    const int nk = (npos + NSUB - 1) / NSUB;
    for (int t = 0; t < nk; ++t) {
        walk();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = rv_[j] & kv & ((unsigned)(bh_[j] + hoff) < (unsigned)g.IH) & ((unsigned)(bw_[j] + woff) < (unsigned)g.IW);
            const unsigned vo = ok ? (unsigned)(rowbase_[j] + aoff) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, lds_ptr(smem + wave * 1024 + j * 4096), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) {
            const unsigned vo = ((rowoffB[j] != OOB) & kv) ? rowoffB[j] + (unsigned)bcol : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, lds_ptr(smem + H2Q_BM * H2_ROWB + wave * 1024 + j * 4096), 16, vo, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            bf16x8 fa[4], fb[2 * TN];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) fa[tt] = *reinterpret_cast<const bf16x8 *>(smem + a16[sx] + tt * 16 * H2_ROWB);
#pragma unroll
            for (int tt = 0; tt < 2 * TN; ++tt) fb[tt] = *reinterpret_cast<const bf16x8 *>(smem + b16[sx] + tt * 16 * H2_ROWB);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2 * TN; ++tn)
                    acc16[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm], fb[tn], acc16[tm][tn], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // every wave's reads are back before the next K-step's DMA lands
    }
    h2_epilogue<ElemBF16, TN, true, 2 * TN, 4>(acc_unused, acc16, g, epi, out, smem, m0, n0, grp, [&](int m) { return m < Mc; },
                                               [&](int m) -> size_t { return ((size_t)grp * Mc + m) * g.Ncols; });
}

template <int TN, int CBK>
__global__ __launch_bounds__(256, TN <= 3 ? 3 : 2) void conv_h2_small_kernel(const unsigned short *__restrict__ act, const unsigned short *__restrict__ wgt,
                                                            void *__restrict__ out, H2Geom g, H2Epi epi, size_t act_group_stride)
{
    // one stage; the epilogue's four staging patches (4 x 32 x 64 TN bytes) + 128 row flags fit inside it
    __shared__ __attribute__((aligned(1024))) char smem[(H2Q_BM + 64 * TN) * H2_ROWB + 256];
    conv_h2_small_body<TN, CBK>(act, wgt, out, g, epi, act_group_stride, smem);
}

// ------------------------------------------------------------------------------------------ shared-row variant (r4)
// What the loop above is bound by (tools/h2_ablate.sh): its time follows the bytes the LDS-DMA moves (with a quarter of
// the DMA left: -22..-26 %), not the fragment reads (with 30 % of them left: -8..-9 %).  For filters with three stride-1 taps along the columns the
// three activation tiles of (chunk, tap row, tap col 0..2) are ONE tile shifted by a pixel - if the GEMM's rows enumerate the
// output pixels over a width padded by two (n, oh, owp), owp in [0, OW + 2): row q + 1 is then always the pixel to the right, the
// two extra positions per image row read the zero padding that a real neighbour needs, and their own outputs are never stored
// (2 / (OW + 2) of the MFMAs: 1.4 % at 144 columns).  So the activation tile is staged ONCE per (chunk, tap row) as 256 + 2 rows
// and the three column taps read their fragments at row offsets 0, 1, 2; only the weight tile changes every K-step:
//   DMA bytes per three K-steps   3 x (32 + 8 TN) KB  ->  33 + 3 x 8 TN KB      (TN = 3: 168 -> 105 KB)
// Same reduction order (chunk, tap row, tap col, channel), same MFMAs: results bit-identical to conv_h2_kernel.
// LDS: two 264-row activation images (the second is filled during the three K-steps of the first) + two weight stages.
// bf16, 64-channel chunks (C % 64 == 0), S = 3, 5 or 7 stride-1 taps along the columns (256 + S - 1 rows, S K-steps per image, padded
// width OW + S - 1), dilation 1 there, stride 1 along the rows; 16x16x32 MFMAs.
constexpr int H2S_AROWS = 264;                                    // 256 + S - 1 <= 262 rows in whole 8-row DMA pieces
constexpr int H2S_AEXT = H2S_AROWS * H2_ROWB;

template <int TN, int S>
__device__ __forceinline__ void conv_h2s_body(const void *__restrict__ act_v, const void *__restrict__ wgt_v, void *__restrict__ out,
                                              const H2Geom &g, const H2Epi &epi, size_t act_group_stride, char *smem)
{
    constexpr int BN = 64 * TN;
    constexpr int BSZ = BN * H2_ROWB;
    char *smemA = smem, *smemB = smem + 2 * H2S_AEXT;
    const char *act = reinterpret_cast<const char *>(act_v);
    const char *wgt = reinterpret_cast<const char *>(wgt_v);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int grp = blockIdx.y;
    static_assert(S >= 2 && S <= 7, "column taps");
    const int OWp = g.OW + S - 1;
    const int Mp = g.ng * g.OH * OWp;                             // GEMM rows over the padded width (host: nMt = ceil(Mp / 256))
    const int tile = xcd_remap(blockIdx.x, g.nMt * g.nNt);
    const int mt = tile / g.nNt, nt = tile - mt * g.nNt;
    const int m0 = mt * H2_BM, n0 = nt * BN;
    const int step_h = __builtin_amdgcn_readfirstlane(g.step_h);
    const int smin = g.step_w < 0 ? (S - 1) * g.step_w : 0;       // lowest column offset among the S taps
    const int nR = __builtin_amdgcn_readfirstlane(g.nR);

    // ---- DMA slots (see conv_h2_body).  Activation pieces p = wave + 8 j, j = 0 .. 3, and piece 32 (wave 0 only) cover the
    // extended rows 8 p .. 8 p + 7; extended row e is the padded-width position q = m0 + e.
    const int lrow = lane >> 3;
    const int lc = (lane & 7) ^ ((((wave & 1) << 2) | (lane >> 4)) & 7);
    const int cin_b = lc * 16;
    const int img0 = m0 / (g.OH * OWp);
    const size_t img_elems = (size_t)g.IH * g.IW * g.IC;
    const char *abase = act + ((size_t)grp * act_group_stride + (size_t)img0 * img_elems) * 2;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(abase, (act_group_stride - (size_t)img0 * img_elems) * 2);
    const char *wbase = wgt + ((size_t)grp * g.Ncols + n0) * g.Kp * 2;
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(wbase, (size_t)(g.Ncols - n0 < BN ? g.Ncols - n0 : BN) * g.Kp * 2);
    int bh_[5], rowbase_[5];
    bool rv_[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int pc = j < 4 ? wave + 8 * j : 32;
        const int e = 8 * pc + lrow;
        const int q = m0 + e;
        bool v = e < H2_BM + S - 1 && q < Mp && (j < 4 || wave == 0);
        const int qq = v ? q : m0;
        const int owp = qq % OWp, t = qq / OWp;
        const int oh = t % g.OH, n = t / g.OH - img0;
        const int col = owp + g.add_w + smin;                      // input column of this extended row
        v = v && (unsigned)col < (unsigned)g.IW;
        const int bh = oh * g.mul_h + g.add_h;
        rv_[j] = v;
        bh_[j] = bh;
        rowbase_[j] = (((n * g.IH + bh) * g.IW + (v ? col : 0)) * g.IC) * 2 + cin_b;
    }
    unsigned rowoffB[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = 8 * (wave + 8 * j) + lrow;
        rowoffB[j] = (n0 + row < g.Ncols) ? (unsigned)row * (unsigned)g.Kp * 2u + (unsigned)cin_b : OOB;
    }
    // ---- scalar walks: the weight tile moves every K-step over (chunk, tap row, tap col), the activation image every third one
    const int RS = g.R * g.S;
    int b_js = 0, b_jr = 0, b_ch = nR > 0 ? 0 : g.nchunks;         // next weight tile to stage
    int a_jr = 0, a_ch = b_ch;                                     // next activation image to stage
    int bcol = 0, hoffA = 0, aoffA = 0;
    bool kvB = false, kvA = false;
    auto walk_b = [&]() {
        bcol = ((b_ch * RS + (g.tap_r0 + b_jr * g.tap_rstep) * g.S + g.tap_s0 + b_js * g.tap_sstep) * 64) * 2;
        kvB = b_ch < g.nchunks;
        ++b_js;
        const bool c2 = b_js >= S;
        b_js = c2 ? 0 : b_js;
        b_jr += c2 ? 1 : 0;
        const bool c3 = b_jr >= nR;
        b_jr = c3 ? 0 : b_jr;
        b_ch += c3 ? 1 : 0;
    };
    auto walk_a = [&]() {
        hoffA = a_jr * step_h;
        aoffA = ((hoffA * g.IW) * g.IC + a_ch * 64) * 2;
        kvA = a_ch < g.nchunks;
        ++a_jr;
        const bool c3 = a_jr >= nR;
        a_jr = c3 ? 0 : a_jr;
        a_ch += c3 ? 1 : 0;
    };
    // branch-free: a scalar branch inside the K-step makes the compiler drain its wait counters at the join.  Piece 32 belongs to
    // wave 0; the other waves issue the same instruction out of range into a 1 KB patch of their own behind the stages (zeros)
    auto dma_ax = [&](int stage, auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        const bool ok = rv_[j] & kvA & ((unsigned)(bh_[j] + hoffA) < (unsigned)g.IH);
        const unsigned vo = ok ? (unsigned)(rowbase_[j] + aoffA) : OOB;
        char *dst = smemA + stage * H2S_AEXT + wave * 1024 + j * 8192;
        if constexpr (j == 4) dst = wave == 0 ? smemA + stage * H2S_AEXT + 32 * 1024 : smemB + 2 * BSZ + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, lds_ptr(dst), 16, vo, 0, 0, 0);
    };
    auto dma_b = [&](int stage, int j) {
        const unsigned vo = ((rowoffB[j] != OOB) & kvB) ? rowoffB[j] + (unsigned)bcol : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, lds_ptr(smemB + stage * BSZ + wave * 1024 + j * 8192), 16, vo, 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;

    f32x16 acc_unused[1][1];
    f32x4v acc16[4][2 * TN];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2 * TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[a][b][r] = 0.f;

    // ---- fragment addresses (16x16x32: lane (r = l & 15, g = l >> 4) holds k = 8 g .. 8 g + 7 of a 32-deep substep).  A row of
    // the activation image shifted by d pixels is physical row R + d: its chunk swizzle is ((R + d) >> 1) & 7
    const int r16 = lane & 15, g4 = lane >> 4;
    int b16[2];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) b16[sx] = (wn * 32 * TN + r16) * H2_ROWB + (((4 * sx + g4) ^ ((r16 >> 1) & 7)) << 4);
    const int nsuper = g.nchunks * nR;

    bf16x8 fa[2][4], fb[2][TN];
    // fragment offsets of the three column taps: tap js reads column base + js * step_w = image row q + d, d = js (forward) or
    // 2 - js (input gradient); row R + d carries the chunk swizzle ((R + d) >> 1) & 7.  Indexed by compile-time js only.
    int a16t[S][2];
#pragma unroll
    for (int js = 0; js < S; ++js) {
        const int rr = r16 + (g.step_w < 0 ? S - 1 - js : js);
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) a16t[js][sx] = (wm * 64 + rr) * H2_ROWB + (((4 * sx + g4) ^ ((rr >> 1) & 7)) << 4);
    }
    auto ld_a = [&](const char *base, auto js_tag, auto sx_tag, auto set_tag) {
        constexpr int set = decltype(set_tag)::value, js = decltype(js_tag)::value, sx = decltype(sx_tag)::value;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) fa[set][tt] = *reinterpret_cast<const bf16x8 *>(base + a16t[js][sx] + tt * 16 * H2_ROWB);
    };
    auto ld_b = [&](const char *base, int sx, int h, auto set_tag) {
        constexpr int set = decltype(set_tag)::value;
#pragma unroll
        for (int tt = 0; tt < TN; ++tt) fb[set][tt] = *reinterpret_cast<const bf16x8 *>(base + b16[sx] + (h * TN + tt) * 16 * H2_ROWB);
    };
    auto mm_row = [&](auto sa_tag, auto sb_tag, auto h_tag, auto tm_tag) {
        constexpr int sa = decltype(sa_tag)::value, sb = decltype(sb_tag)::value, h = decltype(h_tag)::value;
        constexpr int tm = decltype(tm_tag)::value;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            acc16[tm][h * TN + tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[sa][tm], fb[sb][tn], acc16[tm][h * TN + tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mm = [&](auto sa_tag, auto sb_tag, auto h_tag) {
        mm_row(sa_tag, sb_tag, h_tag, I0{});
        mm_row(sa_tag, sb_tag, h_tag, I1{});
        mm_row(sa_tag, sb_tag, h_tag, I2{});
        mm_row(sa_tag, sb_tag, h_tag, I3{});
    };

    // ---- prologue: image 0 and weight tile 0
    walk_a();
    dma_ax(0, I0{}); dma_ax(0, I1{}); dma_ax(0, I2{}); dma_ax(0, I3{}); dma_ax(0, I4{});
    walk_b();
#pragma unroll
    for (int j = 0; j < TN; ++j) dma_b(0, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int sA = 0, sB = 0;
    ld_a(smemA, I0{}, I0{}, I0{});
    ld_b(smemB, 0, 0, I0{});
    walk_a();                                                      // image 1: staged behind the K-steps js = 0, 1 of image 0
    // one K-step, column tap js (compile-time: the three taps of an image are unrolled, nothing in here branches)
    auto k_step = [&](auto js_tag) {
        constexpr int js = decltype(js_tag)::value;
        constexpr int jsn = js == S - 1 ? 0 : js + 1;
        const char *A = smemA + sA * H2S_AEXT, *B = smemB + sB * BSZ;
        ld_b(B, 0, 1, I1{});
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I0{}, I0{}, I0{}, I0{});
        walk_b();                                                  // weight tile t + 1 into the other stage (read last in K-step t - 1)
        dma_b(sB ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I0{}, I0{}, I0{}, I1{});
        if constexpr (TN > 1) dma_b(sB ^ 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I0{}, I0{}, I0{}, I2{});
        if constexpr (TN > 2) dma_b(sB ^ 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I0{}, I0{}, I0{}, I3{});
        if constexpr (TN > 3) dma_b(sB ^ 1, 3);
        __builtin_amdgcn_sched_barrier(0);
        ld_a(A, js_tag, I1{}, I1{});
        ld_b(B, 1, 0, I0{});
        __builtin_amdgcn_sched_barrier(0);
        mm(I0{}, I1{}, I1{});
        ld_b(B, 1, 1, I1{});
        __builtin_amdgcn_sched_barrier(0);
        mm(I1{}, I0{}, I0{});
        // everything staged so far has landed (the weight tile of K-step t + 1; the image pieces issued one K-step ago); every
        // fragment read of this K-step is back
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int sAn = js == S - 1 ? sA ^ 1 : sA;
        sB ^= 1;
        ld_a(smemA + sAn * H2S_AEXT, std::integral_constant<int, jsn>{}, I0{}, I0{});
        ld_b(smemB + sB * BSZ, 0, 0, I0{});
        __builtin_amdgcn_sched_barrier(0);
        // the next image goes into the stage the PREVIOUS image was read from, its five piece slots spread over the K-steps of the
        // first S - 1 column taps (slot j behind tap (j (S - 1)) / 5: at most three per K-step) - each has a whole K-step to land
        // (S = 2: there is only tap 0 to hide behind - slots 0 .. 2 and, doubled up, 3 and 4)
        constexpr auto at = [](int j) { return (j * (S - 1)) / 5; };
        constexpr int first = at(0) == js ? 0 : at(1) == js ? 1 : at(2) == js ? 2 : at(3) == js ? 3 : at(4) == js ? 4 : 5;
        mm_row(I1{}, I1{}, I1{}, I0{});
        if constexpr (first < 5) dma_ax(sA ^ 1, std::integral_constant<int, first < 5 ? first : 0>{});
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I1{}, I1{}, I1{}, I1{});
        if constexpr (first + 1 < 5 && at(first + 1 < 5 ? first + 1 : 4) == js) dma_ax(sA ^ 1, std::integral_constant<int, first + 1 < 5 ? first + 1 : 0>{});
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I1{}, I1{}, I1{}, I2{});
        if constexpr (first + 2 < 5 && at(first + 2 < 5 ? first + 2 : 4) == js) dma_ax(sA ^ 1, std::integral_constant<int, first + 2 < 5 ? first + 2 : 0>{});
        if constexpr (S == 2 && js == 0) dma_ax(sA ^ 1, I3{});
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I1{}, I1{}, I1{}, I3{});
        if constexpr (S == 2 && js == 0) dma_ax(sA ^ 1, I4{});
        if constexpr (js == S - 1) walk_a();
        __builtin_amdgcn_sched_barrier(0);
        sA = sAn;
    };
    for (int u = 0; u < nsuper; ++u) {
        k_step(I0{});
        k_step(I1{});
        if constexpr (S > 2) k_step(I2{});
        if constexpr (S > 3) k_step(I3{});
        if constexpr (S > 4) k_step(I4{});
        if constexpr (S > 5) k_step(std::integral_constant<int, 5>{});
        if constexpr (S > 6) k_step(std::integral_constant<int, 6>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // trailing (zero-filling) DMAs must not outlive the block's LDS

    // which of the tile's 256 rows are real output pixels, and which: one division per row here instead of one per stored element.
    // The table sits behind the epilogue's own LDS use (8 staging patches + 256 flags)
    __builtin_amdgcn_s_barrier();                                 // ... of every wave
    int *s_pix = reinterpret_cast<int *>(smem + 8 * 32 * 64 * TN + 256);
    if (tid < H2_BM) {
        const int m = m0 + tid;
        const int owp = m % OWp, t = m / OWp;
        const int oh = t % g.OH, nimg = t / g.OH;
        s_pix[tid] = (m < Mp && owp < g.OW) ? (nimg * g.OHf + oh * g.osh + g.oa) * g.OWf + owp * g.osw + g.ob : -1;
    }
    const int npix = g.ng * g.OHf * g.OWf;
    h2_epilogue<ElemBF16, TN, true, 2 * TN, 8>(acc_unused, acc16, g, epi, out, smem, m0, n0, grp,
                                    [&](int m) { return s_pix[m - m0] >= 0; },
                                    [&](int m) -> size_t { return ((size_t)grp * npix + (size_t)s_pix[m - m0]) * g.Ncols; });
}

// CLASS: the launch is one stride-parity class of a strided input gradient (same code; a symbol of its own, so that a profiler's
// per-kernel averages do not mix these quarter-size launches with the stride-1 layers of the same <TN, S>)
template <int TN, int S, bool CLASS = false>
__global__ __launch_bounds__(512, 2) void conv_h2s_kernel(const unsigned short *__restrict__ act, const unsigned short *__restrict__ wgt,
                                                          void *__restrict__ out, H2Geom g, H2Epi epi, size_t act_group_stride)
{
    __shared__ __attribute__((aligned(1024))) char smem[2 * H2S_AEXT + 2 * 64 * TN * H2_ROWB + 8 * 1024];   // + the waves' dummy patches
    conv_h2s_body<TN, S>(act, wgt, out, g, epi, act_group_stride, smem);
}

// ------------------------------------------------------------------------------------------ shared rows on 96-channel chunks (r5)
// The refiner's outer levels have 96 / 288 channels (hifimusic): no multiple of 64, so their 3x3 layers ran on conv_h2_kernel with
// 32-channel chunks and a 128-column tile for 96 output channels - every activation byte staged for 96 of 128 columns, three
// times (once per column tap): 64 useful flop per DMA byte against 155-230 on the wide levels, and 0.17-0.26 of the bf16 peak.
// Here an LDS row is one pixel's 96-channel chunk (192 bytes), a K-step is one tap x 96 channels (three 32-deep MFMA substeps),
// the block tile is 256 x 96 (8 waves as 4 x 64 rows by 2 x 48 columns = 4 x 3 tiles of 16 x 16: no padding column, no padding
// channel) and the activation image is shared by the three column taps as in conv_h2s_kernel:
//   DMA bytes per three K-steps: 48.4 KB of image + 3 x 18 KB of weights for 14.2 MFLOP = 139 flop per byte.
// 192-byte rows put rows r and r + 4 on the same banks; the 16-byte chunk c of row R is therefore stored at chunk
// (c + ((R >> 2) & 3)) mod 12 - any 16 consecutive rows then cover the 16 bank quads once, at every row offset (the shifted
// reads of the column taps included).  The LDS-DMA writes whole 1 KB pieces (5 1/3 rows), so each lane derives (row, chunk) from
// its linear chunk index once, before the loop.  The weight image is the 32-channel-chunk image the layer already has: a lane's
// 16 bytes lie in ONE 32-channel chunk, the three chunks of a tap are RS * 64 bytes apart.
// Reduction order (96-chunk, tap row, tap col, channel): not the order of conv_h2_kernel<.., 32, ..> - tested against float64.
constexpr int H96_ROWB = 192, H96_NCH = 12;                      // bytes and 16-byte chunks per LDS row
constexpr int H96_APIECES = 49, H96_AEXT = H96_APIECES * 1024;   // 258 rows x 192 B = 49,536 B in whole 1 KB pieces
constexpr int H96_ASLOTS = 7;                                    // pieces p = wave + 8 j, j = 0 .. 6 (p < 49)
constexpr int H96_BN = 96, H96_BPIECES = 18, H96_BSZ = H96_BPIECES * 1024, H96_BSLOTS = 3;
// THREE weight stages: a K-step here is 36 MFMAs per wave (0.5 us per SIMD), shorter than an L2 round trip under load - with two
// stages (tile t + 1 requested at the start of K-step t, awaited at its end) the kernel ran at 0.75 PF/s waiting for it.  Tile
// t + 2 is requested instead and the wait at the end of a K-step leaves its three pieces in flight (counted wait).
constexpr int H96_NSB = 3;

template <int S>
__device__ __forceinline__ void conv_h2s96_body(const void *__restrict__ act_v, const void *__restrict__ wgt_v, void *__restrict__ out,
                                                const H2Geom &g, const H2Epi &epi, size_t act_group_stride, char *smem)
{
    static_assert(S == 3, "column taps");
    // pieces beyond the image / the tile are issued out of range (zeros) into ONE shared 1 KB patch: every wave writes the same zeros
    char *smemA = smem, *smemB = smem + 2 * H96_AEXT, *patch = smemB + H96_NSB * H96_BSZ;
    const char *act = reinterpret_cast<const char *>(act_v);
    const char *wgt = reinterpret_cast<const char *>(wgt_v);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int grp = blockIdx.y;
    const int OWp = g.OW + S - 1;
    const int Mp = g.ng * g.OH * OWp;
    const int tile = xcd_remap(blockIdx.x, g.nMt * g.nNt);
    const int mt = tile / g.nNt, nt = tile - mt * g.nNt;
    const int m0 = mt * H2_BM, n0 = nt * H96_BN;
    const int step_h = __builtin_amdgcn_readfirstlane(g.step_h);
    const int smin = g.step_w < 0 ? (S - 1) * g.step_w : 0;
    const int nR = __builtin_amdgcn_readfirstlane(g.nR);
    const int nch = g.IC / 96;
    const int RS = g.R * g.S;

    // ---- DMA slots: lane L of piece p holds the linear chunk n = 64 p + L = 12 row + physical chunk
    const int img0 = m0 / (g.OH * OWp);
    const size_t img_elems = (size_t)g.IH * g.IW * g.IC;
    const char *abase = act + ((size_t)grp * act_group_stride + (size_t)img0 * img_elems) * 2;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(abase, (act_group_stride - (size_t)img0 * img_elems) * 2);
    const char *wbase = wgt + ((size_t)grp * g.Ncols + n0) * g.Kp * 2;
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(wbase, (size_t)(g.Ncols - n0 < H96_BN ? g.Ncols - n0 : H96_BN) * g.Kp * 2);
    int bh_[H96_ASLOTS], rowbase_[H96_ASLOTS];
    bool rv_[H96_ASLOTS];
#pragma unroll
    for (int j = 0; j < H96_ASLOTS; ++j) {
        const int pc = wave + 8 * j;
        const int n = 64 * pc + lane;
        const int e = n / H96_NCH, phys = n - e * H96_NCH;
        int lc = phys - ((e >> 2) & 3);
        lc += lc < 0 ? H96_NCH : 0;
        const int q = m0 + e;
        bool v = pc < H96_APIECES && e < H2_BM + S - 1 && q < Mp;
        const int qq = v ? q : m0;
        const int owp = qq % OWp, t = qq / OWp;
        const int oh = t % g.OH, nimg = t / g.OH - img0;
        const int col = owp + g.add_w + smin;
        v = v && (unsigned)col < (unsigned)g.IW;
        const int bh = oh * g.mul_h + g.add_h;
        rv_[j] = v;
        bh_[j] = bh;
        rowbase_[j] = (((nimg * g.IH + bh) * g.IW + (v ? col : 0)) * g.IC) * 2 + lc * 16;
    }
    unsigned rowoffB[H96_BSLOTS];
#pragma unroll
    for (int j = 0; j < H96_BSLOTS; ++j) {
        const int pc = wave + 8 * j;
        const int n = 64 * pc + lane;
        const int row = n / H96_NCH, phys = n - row * H96_NCH;
        int lc = phys - ((row >> 2) & 3);
        lc += lc < 0 ? H96_NCH : 0;
        const bool ok = pc < H96_BPIECES && n0 + row < g.Ncols;
        // 32-channel chunk lc >> 2 of the tap (RS * 64 bytes further per chunk in the weight image), 16-byte piece lc & 3 of it
        rowoffB[j] = ok ? (unsigned)row * (unsigned)g.Kp * 2u + (unsigned)(lc >> 2) * (unsigned)RS * 64u + (unsigned)(lc & 3) * 16u : OOB;
    }
    // ---- scalar walks (as conv_h2s_body): weight tile per K-step over (96-chunk, tap row, tap col), image per (96-chunk, tap row)
    int b_js = 0, b_jr = 0, b_ch = nR > 0 ? 0 : nch;
    int a_jr = 0, a_ch = b_ch;
    int bcol = 0, hoffA = 0, aoffA = 0;
    bool kvB = false, kvA = false;
    auto walk_b = [&]() {
        bcol = ((3 * b_ch * RS + b_jr * g.S + b_js) * 32) * 2;
        kvB = b_ch < nch;
        ++b_js;
        const bool c2 = b_js >= S;
        b_js = c2 ? 0 : b_js;
        b_jr += c2 ? 1 : 0;
        const bool c3 = b_jr >= nR;
        b_jr = c3 ? 0 : b_jr;
        b_ch += c3 ? 1 : 0;
    };
    auto walk_a = [&]() {
        hoffA = a_jr * step_h;
        aoffA = ((hoffA * g.IW) * g.IC + a_ch * 96) * 2;
        kvA = a_ch < nch;
        ++a_jr;
        const bool c3 = a_jr >= nR;
        a_jr = c3 ? 0 : a_jr;
        a_ch += c3 ? 1 : 0;
    };
    // branch-free (see conv_h2s_body): pieces beyond the image / the tile go out of range into the wave's own 1 KB patch
    auto dma_ax = [&](int stage, auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        const bool ok = rv_[j] & kvA & ((unsigned)(bh_[j] + hoffA) < (unsigned)g.IH);
        const unsigned vo = ok ? (unsigned)(rowbase_[j] + aoffA) : OOB;
        char *dst = smemA + stage * H96_AEXT + (wave + 8 * j) * 1024;
        if constexpr (8 * j + 7 >= H96_APIECES) dst = wave + 8 * j < H96_APIECES ? dst : patch;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, lds_ptr(dst), 16, vo, 0, 0, 0);
    };
    auto dma_b = [&](int stage, auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        const unsigned vo = ((rowoffB[j] != OOB) & kvB) ? rowoffB[j] + (unsigned)bcol : OOB;
        char *dst = smemB + stage * H96_BSZ + (wave + 8 * j) * 1024;
        if constexpr (8 * j + 7 >= H96_BPIECES) dst = wave + 8 * j < H96_BPIECES ? dst : patch;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, lds_ptr(dst), 16, vo, 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>;

    f32x16 acc_unused[1][1];
    f32x4v acc16[4][3];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[a][b][r] = 0.f;

    // ---- fragment addresses: lane (r = l & 15, q = l >> 4) holds k = 8 q .. 8 q + 7 of substep sx = logical chunk 4 sx + q of its row
    const int r16 = lane & 15, g4 = lane >> 4;
    auto chunk_off = [](int row, int c) { return row * H96_ROWB + (((c + ((row >> 2) & 3)) % H96_NCH) << 4); };
    int a16t[S][3], b16[3];
#pragma unroll
    for (int js = 0; js < S; ++js) {
        const int rr = wm * 64 + r16 + (g.step_w < 0 ? S - 1 - js : js);      // + 16 tt: the rotation does not change
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) a16t[js][sx] = chunk_off(rr, 4 * sx + g4);
    }
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) b16[sx] = chunk_off(wn * 48 + r16, 4 * sx + g4);
    bf16x8 fa[3][4], fb[3][3];                                      // one fragment set per substep
    auto ld_a = [&](const char *base, auto js_tag, auto sx_tag) {
        constexpr int js = decltype(js_tag)::value, sx = decltype(sx_tag)::value;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) fa[sx][tt] = *reinterpret_cast<const bf16x8 *>(base + a16t[js][sx] + tt * 16 * H96_ROWB);
    };
    auto ld_b = [&](const char *base, auto sx_tag) {
        constexpr int sx = decltype(sx_tag)::value;
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) fb[sx][tt] = *reinterpret_cast<const bf16x8 *>(base + b16[sx] + tt * 16 * H96_ROWB);
    };
    auto mm_row = [&](auto sx_tag, auto tm_tag) {
        constexpr int sx = decltype(sx_tag)::value, tm = decltype(tm_tag)::value;
#pragma unroll
        for (int tn = 0; tn < 3; ++tn) acc16[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[sx][tm], fb[sx][tn], acc16[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: image 0 and weight tile 0
    walk_a();
    dma_ax(0, I0{}); dma_ax(0, I1{}); dma_ax(0, I2{}); dma_ax(0, I3{}); dma_ax(0, I4{}); dma_ax(0, I5{}); dma_ax(0, I6{});
    walk_b();
    dma_b(0, I0{}); dma_b(0, I1{}); dma_b(0, I2{});
    walk_b();                                                      // tile 1 is on its way before the loop starts
    dma_b(1, I0{}); dma_b(1, I1{}); dma_b(1, I2{});
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int sA = 0, sB = 0;
    ld_a(smemA, I0{}, I0{});
    ld_b(smemB, I0{});
    walk_a();                                                      // image 1: staged behind the K-steps js = 0, 1 of image 0
    const int nsuper = nch * nR;
    // one K-step = column tap js x 96 channels.  On entry the fragments of substep 0 are in registers; one barrier per K-step
    auto k_step = [&](auto js_tag) {
        constexpr int js = decltype(js_tag)::value;
        constexpr int jsn = js == S - 1 ? 0 : js + 1;
        const char *A = smemA + sA * H96_AEXT, *B = smemB + sB * H96_BSZ;
        ld_a(A, js_tag, I1{});
        ld_b(B, I1{});
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I0{}, I0{});
        const int sBi = sB == 0 ? 2 : sB - 1;                      // weight tile t + 2 into the stage tile t - 1 was read from
        walk_b();
        dma_b(sBi, I0{});
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I0{}, I1{});
        dma_b(sBi, I1{});
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I0{}, I2{});
        dma_b(sBi, I2{});
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I0{}, I3{});
        ld_a(A, js_tag, I2{});
        ld_b(B, I2{});
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I1{}, I0{}); mm_row(I1{}, I1{}); mm_row(I1{}, I2{}); mm_row(I1{}, I3{});
        // the weight tile of K-step t + 1 and the image pieces issued one K-step ago have landed (the three pieces of tile t + 2,
        // issued last, may still be in flight); every fragment read is back
        asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int sAn = js == S - 1 ? sA ^ 1 : sA;
        sB = sB == 2 ? 0 : sB + 1;
        ld_a(smemA + sAn * H96_AEXT, std::integral_constant<int, jsn>{}, I0{});
        ld_b(smemB + sB * H96_BSZ, I0{});
        __builtin_amdgcn_sched_barrier(0);
        // the next image goes into the stage the PREVIOUS image was read from: slots 0 .. 3 behind tap 0, 4 .. 6 behind tap 1, so
        // that the last of them has the whole K-step of tap 2 to land
        mm_row(I2{}, I0{});
        if constexpr (js == 0) dma_ax(sA ^ 1, I0{});
        if constexpr (js == 1) dma_ax(sA ^ 1, I4{});
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I2{}, I1{});
        if constexpr (js == 0) dma_ax(sA ^ 1, I1{});
        if constexpr (js == 1) dma_ax(sA ^ 1, I5{});
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I2{}, I2{});
        if constexpr (js == 0) dma_ax(sA ^ 1, I2{});
        if constexpr (js == 1) dma_ax(sA ^ 1, I6{});
        __builtin_amdgcn_sched_barrier(0);
        mm_row(I2{}, I3{});
        if constexpr (js == 0) dma_ax(sA ^ 1, I3{});
        if constexpr (js == S - 1) walk_a();
        __builtin_amdgcn_sched_barrier(0);
        sA = sAn;
    };
    for (int u = 0; u < nsuper; ++u) {
        k_step(I0{});
        k_step(I1{});
        k_step(I2{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // trailing (zero-filling) DMAs must not outlive the block's LDS

    __builtin_amdgcn_s_barrier();
    int *s_pix = reinterpret_cast<int *>(smem + 8 * 32 * 32 * 3 + 256);      // behind the epilogue's 8 staging patches + 256 flags
    if (tid < H2_BM) {
        const int m = m0 + tid;
        const int owp = m % OWp, t = m / OWp;
        s_pix[tid] = (m < Mp && owp < g.OW) ? t * g.OW + owp : -1;
    }
    const int npix = g.ng * g.OH * g.OW;
    h2_epilogue<ElemBF16, 1, true, 3, 8>(acc_unused, acc16, g, epi, out, smem, m0, n0, grp,
                                      [&](int m) { return s_pix[m - m0] >= 0; },
                                      [&](int m) -> size_t { return ((size_t)grp * npix + (size_t)s_pix[m - m0]) * g.Ncols; });
}

template <int S>
__global__ __launch_bounds__(512, 2) void conv_h2s96_kernel(const unsigned short *__restrict__ act, const unsigned short *__restrict__ wgt,
                                                            void *__restrict__ out, H2Geom g, H2Epi epi, size_t act_group_stride)
{
    __shared__ __attribute__((aligned(1024))) char smem[2 * H96_AEXT + H96_NSB * H96_BSZ + 1024];   // + the shared dummy patch
    conv_h2s96_body<S>(act, wgt, out, g, epi, act_group_stride, smem);
}

// The same layers on 128-row tiles without any ring: one image of 130 rows (25 KB), one weight tile (18 KB), four waves - 44 KB, so
// THREE blocks share a CU and hide each other's round trips, as in conv_h2_small_kernel.  conv_h2s96_kernel moves 139 flop per DMA
// byte but, at one block per CU, reached 0.78 PF/s on 5.6 TB/s of LDS-DMA traffic - half of what the L2 delivers: it waited.  Here a
// block stages twice the weight bytes per flop (90 flop per byte) and waits for every one of them - and nobody cares, because two
// other blocks are computing meanwhile.  Same reduction order and MFMAs as conv_h2s96_kernel: bit-identical
// (MQGAN_H2_96_BIG=1 selects the 256-row kernel).
constexpr int H96Q_BM = 128, H96Q_APIECES = 25, H96Q_AEXT = H96Q_APIECES * 1024;      // 130 rows x 192 B = 24,960 B

template <int S>
__device__ __forceinline__ void conv_h2s96q_body(const void *__restrict__ act_v, const void *__restrict__ wgt_v, void *__restrict__ out,
                                                 const H2Geom &g, const H2Epi &epi, size_t act_group_stride, char *smem)
{
    static_assert(S == 3, "column taps");
    char *smemA = smem, *smemB = smem + H96Q_AEXT, *patch = smemB + H96_BSZ;
    const char *act = reinterpret_cast<const char *>(act_v);
    const char *wgt = reinterpret_cast<const char *>(wgt_v);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int grp = blockIdx.y;
    const int OWp = g.OW + S - 1;
    const int Mp = g.ng * g.OH * OWp;
    const int tile = xcd_remap(blockIdx.x, g.nMt * g.nNt);
    const int mt = tile / g.nNt, nt = tile - mt * g.nNt;
    const int m0 = mt * H96Q_BM, n0 = nt * H96_BN;
    const int step_h = __builtin_amdgcn_readfirstlane(g.step_h);
    const int smin = g.step_w < 0 ? (S - 1) * g.step_w : 0;
    const int nR = __builtin_amdgcn_readfirstlane(g.nR);
    const int nch = g.IC / 96;
    const int RS = g.R * g.S;
    const int img0 = m0 / (g.OH * OWp);
    const size_t img_elems = (size_t)g.IH * g.IW * g.IC;
    const char *abase = act + ((size_t)grp * act_group_stride + (size_t)img0 * img_elems) * 2;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(abase, (act_group_stride - (size_t)img0 * img_elems) * 2);
    const char *wbase = wgt + ((size_t)grp * g.Ncols + n0) * g.Kp * 2;
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(wbase, (size_t)(g.Ncols - n0 < H96_BN ? g.Ncols - n0 : H96_BN) * g.Kp * 2);
    // image pieces p = wave + 4 j, j = 0 .. 6 (p < 25); weight pieces p = wave + 4 j, j = 0 .. 4 (p < 18); see conv_h2s96_body
    constexpr int ASL = 7, BSL = 5;
    int bh_[ASL], rowbase_[ASL];
    bool rv_[ASL];
#pragma unroll
    for (int j = 0; j < ASL; ++j) {
        const int pc = wave + 4 * j;
        const int n = 64 * pc + lane;
        const int e = n / H96_NCH, phys = n - e * H96_NCH;
        int lc = phys - ((e >> 2) & 3);
        lc += lc < 0 ? H96_NCH : 0;
        const int q = m0 + e;
        bool v = pc < H96Q_APIECES && e < H96Q_BM + S - 1 && q < Mp;
        const int qq = v ? q : m0;
        const int owp = qq % OWp, t = qq / OWp;
        const int oh = t % g.OH, nimg = t / g.OH - img0;
        const int col = owp + g.add_w + smin;
        v = v && (unsigned)col < (unsigned)g.IW;
        rv_[j] = v;
        bh_[j] = oh * g.mul_h + g.add_h;
        rowbase_[j] = (((nimg * g.IH + bh_[j]) * g.IW + (v ? col : 0)) * g.IC) * 2 + lc * 16;
    }
    unsigned rowoffB[BSL];
#pragma unroll
    for (int j = 0; j < BSL; ++j) {
        const int pc = wave + 4 * j;
        const int n = 64 * pc + lane;
        const int row = n / H96_NCH, phys = n - row * H96_NCH;
        int lc = phys - ((row >> 2) & 3);
        lc += lc < 0 ? H96_NCH : 0;
        const bool ok = pc < H96_BPIECES && n0 + row < g.Ncols;
        rowoffB[j] = ok ? (unsigned)row * (unsigned)g.Kp * 2u + (unsigned)(lc >> 2) * (unsigned)RS * 64u + (unsigned)(lc & 3) * 16u : OOB;
    }
    f32x16 acc_unused[1][1];
    f32x4v acc16[4][3];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[a][b][r] = 0.f;
    const int r16 = lane & 15, g4 = lane >> 4;
    auto chunk_off = [](int row, int c) { return row * H96_ROWB + (((c + ((row >> 2) & 3)) % H96_NCH) << 4); };
    int a16t[S][3], b16[3];
#pragma unroll
    for (int js = 0; js < S; ++js) {
        const int rr = wm * 64 + r16 + (g.step_w < 0 ? S - 1 - js : js);
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) a16t[js][sx] = chunk_off(rr, 4 * sx + g4);
    }
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) b16[sx] = chunk_off(wn * 48 + r16, 4 * sx + g4);

    for (int ch = 0; ch < nch; ++ch)
        for (int jr = 0; jr < nR; ++jr) {
            const int hoff = jr * step_h;
            const int aoff = ((hoff * g.IW) * g.IC + ch * 96) * 2;
#pragma unroll
            for (int j = 0; j < ASL; ++j) {
                const bool ok = rv_[j] & ((unsigned)(bh_[j] + hoff) < (unsigned)g.IH);
                const unsigned vo = ok ? (unsigned)(rowbase_[j] + aoff) : OOB;
                char *dst = wave + 4 * j < H96Q_APIECES ? smemA + (wave + 4 * j) * 1024 : patch;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, lds_ptr(dst), 16, vo, 0, 0, 0);
            }
#pragma unroll
            for (int js = 0; js < S; ++js) {
                const unsigned bcol = (unsigned)(((3 * ch * RS + jr * g.S + js) * 32) * 2);
#pragma unroll
                for (int j = 0; j < BSL; ++j) {
                    const unsigned vo = rowoffB[j] != OOB ? rowoffB[j] + bcol : OOB;
                    char *dst = wave + 4 * j < H96_BPIECES ? smemB + (wave + 4 * j) * 1024 : patch;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, lds_ptr(dst), 16, vo, 0, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    bf16x8 fa[4], fb[3];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) fa[tt] = *reinterpret_cast<const bf16x8 *>(smemA + a16t[js][sx] + tt * 16 * H96_ROWB);
#pragma unroll
                    for (int tt = 0; tt < 3; ++tt) fb[tt] = *reinterpret_cast<const bf16x8 *>(smemB + b16[sx] + tt * 16 * H96_ROWB);
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 3; ++tn)
                            acc16[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm], fb[tn], acc16[tm][tn], 0, 0, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                      // reads are back before the next weight tile (or image) lands
            }
        }
    int *s_pix = reinterpret_cast<int *>(smem + 4 * 32 * 32 * 3 + 256);      // behind the epilogue's 4 staging patches + flags
    if (tid < H96Q_BM) {
        const int m = m0 + tid;
        const int owp = m % OWp, t = m / OWp;
        s_pix[tid] = (m < Mp && owp < g.OW) ? t * g.OW + owp : -1;
    }
    const int npix = g.ng * g.OH * g.OW;
    h2_epilogue<ElemBF16, 1, true, 3, 4>(acc_unused, acc16, g, epi, out, smem, m0, n0, grp,
                                         [&](int m) { return s_pix[m - m0] >= 0; },
                                         [&](int m) -> size_t { return ((size_t)grp * npix + (size_t)s_pix[m - m0]) * g.Ncols; });
}

template <int S>
__global__ __launch_bounds__(256, 3) void conv_h2s96q_kernel(const unsigned short *__restrict__ act, const unsigned short *__restrict__ wgt,
                                                             void *__restrict__ out, H2Geom g, H2Epi epi, size_t act_group_stride)
{
    __shared__ __attribute__((aligned(1024))) char smem[H96Q_AEXT + H96_BSZ + 1024];
    conv_h2s96q_body<S>(act, wgt, out, g, epi, act_group_stride, smem);
}

template <int TN, int NSTAGE, int CBK, bool PARITY, bool MF16 = false>
__global__ __launch_bounds__(512, 2) void conv_h2_kernel(const unsigned short *__restrict__ act, const unsigned short *__restrict__ wgt,
                                                         void *__restrict__ out, H2Geom g, H2Parity pi, H2Epi epi,
                                                         size_t act_group_stride)
{
    __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * (H2_BM + 64 * TN) * H2_ROWB];
    conv_h2_body<ElemBF16, TN, NSTAGE, CBK, PARITY, MF16>(act, wgt, out, g, pi, epi, act_group_stride, smem);
}
// exact f32 products on the same LDS-DMA structure (MQ_PRECISION_F32, C % 16 == 0): f32 tensors, the existing KRSC / CRSK
// weight layouts (channel blocks of 16 = 64-byte runs, two (chunk, tap) positions per K-step)
template <int TN, int NSTAGE, bool PARITY>
__global__ __launch_bounds__(512, 2) void conv_d32_kernel(const float *__restrict__ act, const float *__restrict__ wgt,
                                                          float *__restrict__ out, H2Geom g, H2Parity pi, H2Epi epi,
                                                          size_t act_group_stride)
{
    __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * (H2_BM + 64 * TN) * H2_ROWB];
    conv_h2_body<ElemF32, TN, NSTAGE, 16, PARITY>(act, wgt, out, g, pi, epi, act_group_stride, smem);
}

// ------------------------------------------------------------------------------------------ wgrad
// dw[k][kk] = sum_m dy[m][k] * x[pixel(m) + tap(kk)][channel(kk)], kk in plain (tap, channel) order, reduction over the
// pixels m of one group, split over `splits` pixel ranges (slabs reduced in a fixed order afterwards).
// Both operands are pixel-major in HBM, so the LDS images are [64 pixels][256 channels] (512-byte rows) filled by
// LDS-DMA; the MFMA wants 8 consecutive PIXELS of one channel per lane, i.e. the transpose:
// ds_read_b64_tr_b16 (lane 4q+p of a 16-lane group addresses pixel P+q, channels 4p..4p+3; lane i receives channel i of
// the 4 pixels), two reads per 32x32x16 operand.  Bank rule for that read: the 32 lanes of a half touch 4 pixel rows x
// 64 bytes, which must differ mod 256: the 16-byte chunk index of a row is XORed with (pixel & 3) << 2 (on the DMA's
// source side and on the reads).
// Block tile (WM * 32 * TM) output channels x 256 filter columns, 8 waves as WM x (8 / WM):
//   <4, 2> 256 x 256 (K >= 256);   <2, 3> 192 x 256 (K = 192, 384, ...);   <2, 2> 128 x 256 (the rest).
struct H2Wgrad {
    int OH, OW, IH, IW, IC;
    int S, sh, sw, ph, pw, dh, dw;
    int Mtot, K, RSC, RS;
    int nKt, nCt;
    int splits, chunk;         // pixels per split (multiple of 64)
    int pipe;                  // conv_wgrad_h2_kernel<2, 2, *>: fragments of substep s + 1 requested before the MFMAs of substep s
};
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;


// ---- bias gradient inside the weight-gradient kernels (r5).  db[k] = sum over the pixels of dy[pixel][k]: the dy tile of every
// K-step is in LDS anyway (64 pixels x 256 channels, 512-byte rows, 16-byte chunk index XORed with the row's swizzle), so the
// blocks add it up on the vector unit.  r2-r4 read dy a second time from HBM for this (h2_colsum_*: 4.0 ms of the config-3 step,
// 67 launches).  The work is SPREAD over the column tiles of a (split, group, row tile): column tile ct owns the 8-channel chunks
// ct, ct + nCt, ... of the 32 - one to three chunks of 64 pixels per K-step, a 16-byte read for some of its threads.  (First
// version: column tile 0 summed everything, 4 reads + 32 adds per thread and K-step - those blocks became the stragglers of every
// launch and the step got 1.5 ms SLOWER than with the separate passes.)  The reads are inline asm: in front of a C++ load from
// LDS the compiler waits for every LDS-DMA in flight (vmcnt(0)), i.e. for the NEXT tile.
constexpr int H2_BS_MAX = 4;                      // items per thread: 32 chunks x 64 pixels / 512 threads (one column tile owns all)
template <bool MF16>
__device__ __forceinline__ void h2_dy_colsum_step(const char *img, int ct, int nct, float (&bs)[H2_BS_MAX][8])
{
#pragma unroll
    for (int r = 0; r < H2_BS_MAX; ++r) {
        const int item = threadIdx.x + 512 * r;
        const int c16 = ct + nct * (item >> 6), p = item & 63;
        if (c16 < 32) {                                            // wave-uniform: a wave's 64 items share the chunk
            const int swz = ((p & 3) << 2) | (MF16 ? (((p >> 3) & 1) << 1) : 0);
            const unsigned a = (unsigned)reinterpret_cast<uintptr_t>(img + p * 512 + ((c16 ^ swz) << 4));
            u32x4 v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bs[r][2 * e] += __builtin_bit_cast(float, v[e] << 16);
                bs[r][2 * e + 1] += __builtin_bit_cast(float, v[e] & 0xFFFF0000u);
            }
        }
    }
}
// after the K loop (every wave past its last barrier, LDS free): the 64 pixels' sums of a channel are added in a fixed order and go
// to the (split, group) row of the bias slab
__device__ __forceinline__ void h2_dy_colsum_store(const float (&bs)[H2_BS_MAX][8], char *smem, float *__restrict__ bias_row, int ct,
                                                   int nct, int k0, int bmk, int K)
{
    float *sp = reinterpret_cast<float *>(smem);                   // [owned chunk][8 channels][64 pixels]
    __syncthreads();
#pragma unroll
    for (int r = 0; r < H2_BS_MAX; ++r) {
        const int item = threadIdx.x + 512 * r;
        const int slot = item >> 6, p = item & 63;
        if (ct + nct * slot < 32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sp[(slot * 8 + e) * 64 + p] = bs[r][e];
        }
    }
    __syncthreads();
    const int slot = threadIdx.x >> 3, e = threadIdx.x & 7;       // 64 slots x 8 channels >= 32 chunks
    const int c16 = ct + nct * slot;
    const int ch = c16 * 8 + e;
    if (c16 < 32 && ch < bmk && k0 + ch < K) {
        float t = 0.f;
        for (int q = 0; q < 64; ++q) t += sp[(slot * 8 + e) * 64 + q];
        bias_row[k0 + ch] = t;
    }
}
// db[i] = sum over the splits of the bias slab, fixed order
__global__ void h2_bias_reduce_kernel(const float *__restrict__ slab, int splits, int n, float *__restrict__ db)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float t = 0.f;
    for (int k = 0; k < splits; ++k) t += slab[(size_t)k * n + i];
    db[i] = t;
}

template <int WM, int TM, int WRAPS, bool MF16>
__device__ __forceinline__ void conv_wgrad_h2_body(const unsigned short *__restrict__ x, const unsigned short *__restrict__ dy,
                                                   float *__restrict__ out, float *__restrict__ bias_slab, const H2Wgrad &g, int groups,
                                                   char *smem)
{
    constexpr int WN = 8 / WM, TNW = WM;           // wave grid and 32-column tiles per wave (WN * 32 * TNW = 256)
    constexpr int BMK = WM * 32 * TM;
    constexpr int IMG = 64 * 512;                  // one operand image: 64 pixels x 512 bytes
    constexpr int STAGE = 2 * IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    // all tiles of one (split, group) share a pixel range: keep them on one XCD (see igemm.hip)
    const int ntile = gridDim.x;
    const int lin = blockIdx.x + ntile * (blockIdx.y + gridDim.y * blockIdx.z);
    const int virt = xcd_remap(lin, ntile * gridDim.y * gridDim.z);
    const int tile = virt % ntile, sg = virt / ntile;
    const int grp = sg % gridDim.y, split = sg / gridDim.y;
    const int kt = tile / g.nCt, ct = tile - kt * g.nCt;
    const int k0 = kt * BMK, c0 = ct * 256;
    const int m_begin = split * g.chunk;
    const int m_end = min(g.Mtot, m_begin + g.chunk);
    const int ng = g.Mtot / (g.OH * g.OW);
    const int img0 = m_begin / (g.OH * g.OW);

    // ---- DMA slots: piece p = wave + 8 j = pixels 2p, 2p+1 of the step; lane -> pixel 2p + (lane >> 5), physical chunk lane & 31
    const int pixl = 2 * wave + (lane >> 5);                       // pixel of slot 0; slot j adds 16 j
    // logical chunk (8 channels / columns) this lane fetches = physical chunk XOR the read-side swizzle of its pixel row:
    // (pixel & 3) << 2 spreads the four pixel rows of a transposed read over the four 64-byte windows of the bank space;
    // the 16x16x32 shape reads the SAME channels of two row quads 8 pixels apart in one half-wave: ((pixel >> 3) & 1) << 1
    // puts those on opposite 32-byte halves of the window
    const int lc = (lane & 31) ^ (((pixl & 3) << 2) | (MF16 ? (((pixl >> 3) & 1) << 1) : 0));
    const unsigned short *dyg = dy + ((size_t)grp * g.Mtot + m_begin) * g.K;
    const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(dyg, (size_t)(m_end - m_begin) * g.K * 2);   // rows >= m_end: zeros
    const unsigned short *xg = x + ((size_t)grp * ng + img0) * g.IH * g.IW * g.IC;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(xg, (size_t)(ng - img0) * g.IH * g.IW * g.IC * 2);
    const int chA = 8 * lc;
    const bool okA = chA < BMK && k0 + chA < g.K;
    unsigned voA[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) voA[j] = okA ? (unsigned)(((pixl + 16 * j) * g.K + k0 + chA) * 2) : OOB;
    const int kkB = c0 + 8 * lc;
    const bool okB = kkB < g.RSC;
    const int tapB = okB ? kkB / g.IC : 0, cB = okB ? kkB - tapB * g.IC : 0;
    const int tap_h = (tapB / g.S) * g.dh - g.ph, tap_w = (tapB % g.S) * g.dw - g.pw;
    const int tap_off = ((tap_h * g.IW + tap_w) * g.IC + cB) * 2;
    int b_ow[4], b_ohs[4], b_base[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m_begin + pixl + 16 * j;
        const int mm = m < g.Mtot ? m : m_begin;
        const int ow = mm % g.OW, t2 = mm / g.OW;
        const int oh = t2 % g.OH, n = t2 / g.OH - img0;
        b_ow[j] = ow;
        b_ohs[j] = oh * g.sh;
        b_base[j] = (((n * g.IH + oh * g.sh) * g.IW + ow * g.sw) * g.IC) * 2;
    }
    int left = m_end - m_begin;                                    // pixels of this split from the staged step's first one on
    unsigned soffA = 0;
    auto issue = [&](int stage) {
        char *sa = smem + stage * STAGE + wave * 1024;
        char *sb = sa + IMG;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#ifdef MQ_H2_ABLATE
            if ((MQ_H2_ABLATE & 32) && j > 0) continue;             // ablation (tools/h2_ablate.sh): a quarter of the dy DMA
#endif
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, lds_ptr(sa + j * 8192), 16, voA[j], soffA, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#ifdef MQ_H2_ABLATE
            if ((MQ_H2_ABLATE & 64) && j > 0) continue;             // ... a quarter of the x DMA
#endif
            const bool ok = okB && (pixl + 16 * j) < left && (unsigned)(b_ohs[j] + tap_h) < (unsigned)g.IH &&
                            (unsigned)(b_ow[j] * g.sw + tap_w) < (unsigned)g.IW;
            const unsigned vo = ok ? (unsigned)(b_base[j] + tap_off) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, lds_ptr(sb + j * 8192), 16, vo, 0, 0, 0);
        }
        // advance by 64 pixels
        soffA += 64u * (unsigned)g.K * 2u;
        left -= 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b_ow[j] += 64;
            b_base[j] += 64 * g.sw * g.IC * 2;
#pragma unroll
            for (int i = 0; i < WRAPS; ++i) {
                const bool w1 = b_ow[j] >= g.OW;
                b_ow[j] -= w1 ? g.OW : 0;
                b_ohs[j] += w1 ? g.sh : 0;
                b_base[j] += w1 ? (g.sh * g.IW - g.OW * g.sw) * g.IC * 2 : 0;
                const bool w2 = b_ohs[j] >= g.OH * g.sh;
                b_ohs[j] -= w2 ? g.OH * g.sh : 0;
                b_base[j] += w2 ? (g.IH - g.OH * g.sh) * g.IW * g.IC * 2 : 0;
            }
        }
    };

    f32x16 acc[MF16 ? 1 : TM][MF16 ? 1 : TNW];
    f32x4v acc16[MF16 ? 2 * TM : 1][MF16 ? 2 * TNW : 1];
#pragma unroll
    for (int a = 0; a < (MF16 ? 1 : TM); ++a)
#pragma unroll
        for (int b = 0; b < (MF16 ? 1 : TNW); ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
    for (int a = 0; a < (MF16 ? 2 * TM : 1); ++a)
#pragma unroll
        for (int b = 0; b < (MF16 ? 2 * TNW : 1); ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[a][b][r] = 0.f;

    // transposed-read lane constants: lane 4q+p of a 16-lane group addresses pixel row q, channels 4p..4p+3 and receives
    // channel (lane & 15) of the four rows.  32x32x16: group gsel = (lane >> 4) & 1 -> channels +16 gsel, lane half lh ->
    // pixels +8 lh.  16x16x32: all four groups read the same 16 channels, group g4 = lane >> 4 -> pixels +8 g4.
    const int trq = (lane & 15) >> 2, trp = lane & 3, gsel = (lane >> 4) & 1, g4 = lane >> 4;
    constexpr int NTA = MF16 ? 2 * TM : TM, NTB = MF16 ? 2 * TNW : TNW;
    int aad[NTA], bad[NTB];
#pragma unroll
    for (int t = 0; t < NTA; ++t) {
        const int chan = MF16 ? wm * 32 * TM + t * 16 + 4 * trp : wm * 32 * TM + t * 32 + 16 * gsel + 4 * trp;
        const int swz = MF16 ? ((trq << 2) | ((g4 & 1) << 1)) : (trq << 2);
        aad[t] = (8 * (MF16 ? g4 : lh) + trq) * 512 + ((((chan * 2) >> 4) ^ swz) << 4) + ((chan * 2) & 15);
    }
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
        const int chan = MF16 ? wn * 32 * TNW + t * 16 + 4 * trp : wn * 32 * TNW + t * 32 + 16 * gsel + 4 * trp;
        const int swz = MF16 ? ((trq << 2) | ((g4 & 1) << 1)) : (trq << 2);
        bad[t] = IMG + (8 * (MF16 ? g4 : lh) + trq) * 512 + ((((chan * 2) >> 4) ^ swz) << 4) + ((chan * 2) & 15);
    }
    const int nsteps = (m_end - m_begin + 63) / 64;
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    auto tr8 = [&](const char *p0, int row_off) {                      // 8 pixels of one channel per lane: two transposed reads
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)lds_ptr(p0 + row_off * 512));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)lds_ptr(p0 + (row_off + 4) * 512));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto compute = [&](int stage) {
        const char *base = smem + stage * STAGE;
        if constexpr (MF16) {
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {                            // 32 pixels per substep
                bf16x8 a[NTA], b[NTB];
#pragma unroll
                for (int t = 0; t < NTA; ++t) a[t] = tr8(base + aad[t], 32 * sx);
#pragma unroll
                for (int t = 0; t < NTB; ++t) b[t] = tr8(base + bad[t], 32 * sx);
#pragma unroll
                for (int tm = 0; tm < NTA; ++tm)
#pragma unroll
                    for (int tn = 0; tn < NTB; ++tn)
                        acc16[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[tm], b[tn], acc16[tm][tn], 0, 0, 0);
            }
        } else if (WM == 2 && TM == 2 && g.pipe) {
            // Two fragment sets - only where the registers are: <2, 2> runs at 234 with them, <2, 3> spilled 93 registers and
            // quadrupled its time when it got them, <4, 2> sits at 252 without.  The compiler's own order
            // requests the fragments of substep s + 1 behind the last-but-one MFMA of substep s - an LDS round trip exposed four
            // times per K-step, and the two waves of a SIMD reach it together.  Here the next set is requested first.
            bf16x8 a[2][NTA], b[2][NTB];
#pragma unroll
            for (int t = 0; t < NTA; ++t) a[0][t] = tr8(base + aad[t], 0);
#pragma unroll
            for (int t = 0; t < NTB; ++t) b[0][t] = tr8(base + bad[t], 0);
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                if (sx < 3) {
#pragma unroll
                    for (int t = 0; t < NTA; ++t) a[(sx + 1) & 1][t] = tr8(base + aad[t], 16 * (sx + 1));
#pragma unroll
                    for (int t = 0; t < NTB; ++t) b[(sx + 1) & 1][t] = tr8(base + bad[t], 16 * (sx + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tm = 0; tm < NTA; ++tm)
#pragma unroll
                    for (int tn = 0; tn < NTB; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sx & 1][tm], b[sx & 1][tn], acc[tm][tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                bf16x8 a[NTA], b[NTB];
#pragma unroll
                for (int t = 0; t < NTA; ++t) {
#ifdef MQ_H2_ABLATE
                    if ((MQ_H2_ABLATE & 16) && t > 0) { a[t] = a[0]; continue; }      // ... one fragment read per operand
#endif
                    a[t] = tr8(base + aad[t], 16 * sx);
                }
#pragma unroll
                for (int t = 0; t < NTB; ++t) {
#ifdef MQ_H2_ABLATE
                    if ((MQ_H2_ABLATE & 16) && t > 0) { b[t] = b[0]; continue; }
#endif
                    b[t] = tr8(base + bad[t], 16 * sx);
                }
#pragma unroll
                for (int tm = 0; tm < NTA; ++tm)
#pragma unroll
                    for (int tn = 0; tn < NTB; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
            }
        }
    };
    int st_c = 0;
    const bool sum_dy = bias_slab != nullptr && ct < 32;           // block-uniform: column tile ct owns the chunks ct, ct + nCt, ...
    float bs[H2_BS_MAX][8] = {};
    for (int t = 0; t < nsteps; ++t) {
        issue(st_c ^ 1);                 // past the end of the split: every lane is out of range, the DMA writes zeros
        compute(st_c);
        if (sum_dy) h2_dy_colsum_step<MF16>(smem + st_c * STAGE, ct, g.nCt, bs);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        st_c ^= 1;
    }
    if (sum_dy) h2_dy_colsum_store(bs, smem, bias_slab + ((size_t)split * groups + grp) * g.K, ct, g.nCt, k0, BMK, g.K);

    // slab [split][group][K][RSC], columns scattered into the channel-blocked filter layout (common.h: kidx)
    float *o = out + ((size_t)split * groups + grp) * (size_t)g.K * g.RSC;
    const int cbs = chan_block(g.IC);
#pragma unroll
    for (int tn = 0; tn < NTB; ++tn) {
        const int colp = c0 + wn * 32 * TNW + (MF16 ? tn * 16 + (lane & 15) : tn * 32 + li);
        if (colp >= g.RSC) continue;
        const int tap = colp / g.IC;
        const int col = kidx(tap, colp - tap * g.IC, g.RS, cbs);
#pragma unroll
        for (int tm = 0; tm < NTA; ++tm) {
            if constexpr (MF16) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = k0 + wm * 32 * TM + tm * 16 + (lane >> 4) * 4 + j;
                    if (row < g.K) o[(size_t)row * g.RSC + col] = acc16[tm][tn][j];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = k0 + wm * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < g.K) o[(size_t)row * g.RSC + col] = acc[tm][tn][r];
                }
            }
        }
    }
}

template <int WM, int TM, int WRAPS, bool MF16>
__global__ __launch_bounds__(512, 2) void conv_wgrad_h2_kernel(const unsigned short *__restrict__ x, const unsigned short *__restrict__ dy,
                                                               float *__restrict__ out, float *__restrict__ bias_slab, H2Wgrad g, int groups)
{
    __shared__ __attribute__((aligned(1024))) char smem[2 * 2 * 64 * 512];
    conv_wgrad_h2_body<WM, TM, WRAPS, MF16>(x, dy, out, bias_slab, g, groups, smem);
}

// ------------------------------------------------------------------------------------------ shared-row weight gradient (r4)
// conv_wgrad_h2_kernel is bound by its LDS-DMA bytes too (tools/h2_ablate.sh, bits 4-6: fragment reads -1 %, three quarters of
// the dy DMA -10 %, of the x DMA -12..-20 %, both -22..-30 %).  For three stride-1 column taps the gathered x tiles of (tap row,
// tap col 0..2) are one image shifted by a pixel - when the pixels are enumerated over a width padded by two, as in
// conv_h2s_kernel; dy is zero at the two extra positions of an image row.  The block's columns are re-tiled accordingly: one tap
// row r, one chunk of CHK = 64 CH2 channels, all three column taps = 3 CHK columns (the epilogue scatters them into the (tap,
// channel) order of the slab).  Per 64-pixel K-step the block stages the dy tile (64 x 256 channels, 32 KB, as before) and ONE x
// image of 66 pixels x CHK channels (17 KB at CHK = 128) instead of 64 pixels x 256 gathered columns (32 KB):
//     64 KB per 256 x 256 x 64 products  ->  49 KB per 256 x 384 x 64:  -49 % DMA bytes per product (CHK = 128)
// The three taps read their B fragments at pixel offsets 0, 1, 2 of the image (transposed reads address pixel rows per lane; the
// 64-byte bank window of a row is XOR-ed with the row's low bits, on the DMA's source side and on the reads).
// Block tile 192 output channels x (3 taps x 128 channels), 8 waves as 2 (96 rows each) x 4 (32 channels each, all taps); 32x32x16
// (K = 192, 384, 768 at config 3: whole tiles; against the 192 x 256 tile of conv_wgrad_h2_kernel<2, 3>: 41 KB per 192 x 384 x 64
// products instead of 56 KB per 192 x 256 x 64: -51 %).  S = 3, stride 1, column dilation 1, C % 128 == 0; pixel sums are formed in the same pixel order as conv_wgrad_h2_kernel but
// in other groups of 16 (the zero positions shift the groups): equal within rounding, not bit for bit.
template <int WRAPS>
__device__ __forceinline__ void conv_wgrad_h2s_body(const unsigned short *__restrict__ x, const unsigned short *__restrict__ dy,
                                                    float *__restrict__ out, float *__restrict__ bias_slab, const H2Wgrad &g, int groups,
                                                    char *smem)
{
    constexpr int CH2 = 2, CHK = 64 * CH2;         // channels per chunk: 128
    constexpr int BMK = 192, TMR = 3;              // rows of the block tile; 32-row tiles per wave (2 waves x 96 rows)
    constexpr int RB = CHK * 2;                    // bytes of one x image row
    constexpr int RPP = 1024 / RB;                 // image rows per 1 KB DMA piece
    constexpr int NP = (66 + RPP - 1) / RPP;       // pieces of one image (17 / 9)
    constexpr int BJ = (NP - 1) / 8;               // whole rounds of 8 pieces (2 / 1); piece 8 BJ belongs to wave 0
    constexpr int CPR = RB / 16;                   // 16-byte chunks per image row
    constexpr int IMGA = 64 * 512;                 // dy image: 64 pixels x 256 channels
    constexpr int IMGB = NP * 1024;
    constexpr int STAGE = IMGA + IMGB;
    constexpr int NH = 1;                          // 32-channel column tiles per tap and wave (4 waves x 32 channels)
    constexpr int NTB = 3 * NH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;
    const int ntile = gridDim.x;
    const int lin = blockIdx.x + ntile * (blockIdx.y + gridDim.y * blockIdx.z);
    const int virt = xcd_remap(lin, ntile * gridDim.y * gridDim.z);
    const int tile = virt % ntile, sg = virt / ntile;
    const int grp = sg % gridDim.y, split = sg / gridDim.y;
    const int kt = tile / g.nCt, ct = tile - kt * g.nCt;
    const int nchunk = g.IC / CHK;
    const int tap_r = ct / nchunk, chunk = ct - tap_r * nchunk;
    const int k0 = kt * BMK;
    const int OWp = g.OW + 2;
    const int Mp = g.Mtot;                                         // pixels of a group over the padded width (host: ng OH (OW + 2))
    const int m_begin = split * g.chunk;
    const int m_end = min(Mp, m_begin + g.chunk);
    const int ng = Mp / (g.OH * OWp);
    const int row0 = m_begin / OWp;                                // first (image, output row) of the split: offsets are relative to it
    const int img0 = row0 / g.OH;

    // ---- dy slots (as conv_wgrad_h2_body: piece = wave + 8 j = pixels 2p, 2p + 1 of the step), walked over the padded width
    const int pixl = 2 * wave + (lane >> 5);
    const int lcA = (lane & 31) ^ ((pixl & 3) << 2);
    const unsigned short *dyg = dy + ((size_t)grp * ng * g.OH + row0) * g.OW * g.K;
    const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(dyg, ((size_t)ng * g.OH - row0) * g.OW * g.K * 2);
    const int chA = 8 * lcA;
    const bool okA = chA < BMK && k0 + chA < g.K;
    int a_owp[4], a_idx[4];                                        // padded column; real pixel index relative to row0
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = m_begin + pixl + 16 * j;
        const int row = q / OWp;
        a_owp[j] = q - row * OWp;
        a_idx[j] = (row - row0) * g.OW + a_owp[j];
    }
    // ---- x slots: piece = wave + 8 j (j < BJ) and piece 8 BJ (wave 0 only); lane -> image row RPP piece + lane / CPR, chunk lane % CPR
    const unsigned short *xg = x + ((size_t)grp * ng + img0) * g.IH * g.IW * g.IC;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(xg, (size_t)(ng - img0) * g.IH * g.IW * g.IC * 2);
    const int ripB = lane / CPR;
    auto swzB = [](int row) { return RB == 256 ? (row & 3) << 2 : ((row >> 1) & 1) << 2; };
    const int lcB = (lane % CPR) ^ swzB(ripB);                     // RPP pieces start on multiples of 4 (8) rows
    const int tap_h = tap_r * g.dh - g.ph;
    const int x_const = ((tap_h * g.IW - g.pw) * g.IC + chunk * CHK + 8 * lcB) * 2;
    int b_owp[BJ + 1], b_oh[BJ + 1], b_base[BJ + 1];
    bool b_row_ok[BJ + 1];
#pragma unroll
    for (int j = 0; j <= BJ; ++j) {
        const int piece = j < BJ ? wave + 8 * j : 8 * BJ;
        const int e = RPP * piece + ripB;                          // image row
        b_row_ok[j] = e < 66 && (j < BJ || wave == 0);
        const int q = m_begin + e;
        const int row = q / OWp;
        b_owp[j] = q - row * OWp;
        const int n = row / g.OH;
        b_oh[j] = row - n * g.OH;
        b_base[j] = ((((n - img0) * g.IH + b_oh[j]) * g.IW + b_owp[j]) * g.IC) * 2;
    }
    int left = m_end - m_begin;                                    // pixels of this split from the staged step's first one on
    auto issue = [&](int stage) {
        char *sa = smem + stage * STAGE + wave * 1024;
        char *sb = smem + stage * STAGE + IMGA;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = okA && (pixl + 16 * j) < left && a_owp[j] < g.OW;
            const unsigned vo = ok ? (unsigned)((a_idx[j] * g.K + k0 + chA) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, lds_ptr(sa + j * 8192), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j <= BJ; ++j) {
            const bool ok = b_row_ok[j] && (unsigned)(b_oh[j] + tap_h) < (unsigned)g.IH && (unsigned)(b_owp[j] - g.pw) < (unsigned)g.IW;
            const unsigned vo = ok ? (unsigned)(b_base[j] + x_const) : OOB;
            // the last piece is wave 0's; the other waves issue the same instruction out of range into a patch of their own (no branch)
            char *dst = j < BJ ? sb + (wave + 8 * j) * 1024 : (wave == 0 ? sb + 8 * BJ * 1024 : smem + 2 * STAGE + wave * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, lds_ptr(dst), 16, vo, 0, 0, 0);
        }
        // advance by 64 positions of the padded enumeration
        left -= 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a_owp[j] += 64;
            a_idx[j] += 64;
#pragma unroll
            for (int i = 0; i < WRAPS; ++i) {
                const bool w1 = a_owp[j] >= OWp;
                a_owp[j] -= w1 ? OWp : 0;
                a_idx[j] -= w1 ? 2 : 0;                              // next row: + OW, column - (OW + 2)
            }
        }
#pragma unroll
        for (int j = 0; j <= BJ; ++j) {
            b_owp[j] += 64;
            b_base[j] += 64 * g.IC * 2;
#pragma unroll
            for (int i = 0; i < WRAPS; ++i) {
                const bool w1 = b_owp[j] >= OWp;
                b_owp[j] -= w1 ? OWp : 0;
                b_oh[j] += w1 ? 1 : 0;
                b_base[j] += w1 ? (g.IW - OWp) * g.IC * 2 : 0;
                const bool w2 = b_oh[j] >= g.OH;
                b_oh[j] -= w2 ? g.OH : 0;
                b_base[j] += w2 ? (g.IH - g.OH) * g.IW * g.IC * 2 : 0;
            }
        }
    };

    f32x16 acc[TMR][NTB];
#pragma unroll
    for (int a = 0; a < TMR; ++a)
#pragma unroll
        for (int b = 0; b < NTB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transposed-read lane constants (conv_wgrad_h2_body, 32x32x16): lane 4 q + p of a 16-lane group addresses pixel row q,
    // channels 4 p .. 4 p + 3; group gsel = (lane >> 4) & 1 -> channels + 16, lane half lh -> pixels + 8
    const int trq = (lane & 15) >> 2, trp = lane & 3, gsel = (lane >> 4) & 1;
    int aad[TMR], bad[3][NH];
#pragma unroll
    for (int t = 0; t < TMR; ++t) {
        const int chan = wm * 96 + t * 32 + 16 * gsel + 4 * trp;
        aad[t] = (8 * lh + trq) * 512 + ((((chan * 2) >> 4) ^ (trq << 2)) << 4) + ((chan * 2) & 15);
    }
#pragma unroll
    for (int sft = 0; sft < 3; ++sft)
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int chan = wn * 32 + h * 32 + 16 * gsel + 4 * trp;
            const int row = 8 * lh + trq + sft;                      // + 16 sx, + 4: multiples of 4 leave the swizzle alone
            bad[sft][h] = IMGA + row * RB + ((((chan * 2) >> 4) ^ swzB(row)) << 4) + ((chan * 2) & 15);
        }
    const int nsteps = (m_end - m_begin + 63) / 64;
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    auto tr8 = [&](const char *p0, int row_off, int rowb) {         // 8 pixels of one channel per lane: two transposed reads
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)lds_ptr(p0 + row_off * rowb));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)lds_ptr(p0 + (row_off + 4) * rowb));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto compute = [&](int stage) {
        const char *base = smem + stage * STAGE;
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
            bf16x8 a[TMR], b[NTB];
#pragma unroll
            for (int t = 0; t < TMR; ++t) a[t] = tr8(base + aad[t], 16 * sx, 512);
#pragma unroll
            for (int sft = 0; sft < 3; ++sft)
#pragma unroll
                for (int h = 0; h < NH; ++h) b[sft * NH + h] = tr8(base + bad[sft][h], 16 * sx, RB);
#pragma unroll
            for (int tm = 0; tm < TMR; ++tm)
#pragma unroll
                for (int tn = 0; tn < NTB; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
    };
    int st_c = 0;
    const bool sum_dy = bias_slab != nullptr && ct < 32;           // block-uniform; dy is zero at the padded positions
    float bs[H2_BS_MAX][8] = {};
    for (int t = 0; t < nsteps; ++t) {
        issue(st_c ^ 1);                 // past the end of the split: dy out of range (zeros); the x image is whatever follows
        compute(st_c);
        if (sum_dy) h2_dy_colsum_step<false>(smem + st_c * STAGE, ct, g.nCt, bs);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        st_c ^= 1;
    }
    if (sum_dy) h2_dy_colsum_store(bs, smem, bias_slab + ((size_t)split * groups + grp) * g.K, ct, g.nCt, k0, BMK, g.K);

    // slab [split][group][K][RSC]; the tile's columns (tap col s, channel) go to (tap, channel) of the channel-blocked filter layout
    float *o = out + ((size_t)split * groups + grp) * (size_t)g.K * g.RSC;
    const int cbs = chan_block(g.IC);
#pragma unroll
    for (int sft = 0; sft < 3; ++sft)
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int tap = tap_r * 3 + sft;
            const int col = kidx(tap, chunk * CHK + wn * 32 + h * 32 + li, g.RS, cbs);
#pragma unroll
            for (int tm = 0; tm < TMR; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = k0 + wm * 96 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < g.K) o[(size_t)row * g.RSC + col] = acc[tm][sft * NH + h][r];
                }
        }
}

template <int WRAPS>
__global__ __launch_bounds__(512, 2) void conv_wgrad_h2s_kernel(const unsigned short *__restrict__ x, const unsigned short *__restrict__ dy,
                                                                float *__restrict__ out, float *__restrict__ bias_slab, H2Wgrad g,
                                                                int groups)
{
    constexpr int NP = 17;                                          // 66 image rows of 256 bytes in 1 KB pieces
    __shared__ __attribute__((aligned(1024))) char smem[2 * (64 * 512 + NP * 1024) + 8 * 1024];
    conv_wgrad_h2s_body<WRAPS>(x, dy, out, bias_slab, g, groups, smem);
}

// ------------------------------------------------------------------------------------------ shared-row weight gradient, 96-channel chunks (r6)
// The refiner's outer levels at config 3 have 96 / 192 / 288 / 576 input channels and 96 / 192 / 384 filters (hifimusic: base 96).
// No multiple of 128, so their weight gradients ran on conv_wgrad_h2_kernel<2, 2> / <2, 3> with gathered 256-column tiles: K = 96
// filled 96 of the 128 tile rows and 96 of the 256 channels of every staged dy row - 49 useful flop per DMA byte, 0.16 of the bf16
// peak on 5 ms of the step (the r5 review's "kernel furthest below its roofline with >= 4 ms").  Here, as in conv_h2s96_kernel on
// the forward side, an LDS row is one pixel's 96-channel chunk (192 bytes):
//   block tile  96 RG filters x (one tap row, one 96-channel chunk, the three column taps = 288 columns), 3 RG waves, each 96 x 96
//               (3 x 3 MFMA tiles of 32 x 32 x 16: the same per-wave work as conv_wgrad_h2s_kernel)
//   per 64-pixel K-step: RG dy images of 64 pixels x 96 filters (12 KB each) + ONE x image of 66 pixels x 96 channels (12.4 KB),
//               shared by the three column taps at pixel offsets 0, 1, 2 over the width padded by two (dy is zero at the two
//               extra positions of a row): 141 (RG = 1) / 191 (RG = 2) flop per DMA byte
// 192-byte rows need no bank swizzle for the transposed reads: the four pixel rows a half-wave addresses start 192 bytes apart,
// i.e. at 0, 192, 128, 64 mod 256 - four disjoint 64-byte windows at every row offset (column-tap shifts included).  The LDS-DMA
// writes whole 1 KB pieces (5 1/3 rows), so each lane derives (row, 16-byte chunk) from its linear chunk index once, before the
// loop.  Two (RG = 1: 51 KB) / one (RG = 2: 75 KB) blocks per CU by registers (144 accumulators).
// S = 3, stride 1, column dilation 1, C % 96 == 0, K % 96 == 0 (RG = 2 when K % 192 == 0), Q >= 14.  Pixel sums in the order of
// conv_wgrad_h2s_kernel (groups of 16 over the padded width): equal to conv_wgrad_h2_kernel within rounding, not bit for bit.
constexpr int W96_ROWB = 192, W96_DYP = 12, W96_XP = 13;          // bytes per LDS row; 1 KB pieces of a dy image (64 rows) / the x image (66 rows)
constexpr int W96_BSR = 4;                                        // bias-sum items per thread: 12 RG chunks x 64 pixels / (192 RG threads)

template <int RG, int WRAPS>
__device__ __forceinline__ void conv_wgrad_h2s96_body(const unsigned short *__restrict__ x, const unsigned short *__restrict__ dy,
                                                      float *__restrict__ out, float *__restrict__ bias_slab, const H2Wgrad &g, int groups,
                                                      char *smem)
{
    constexpr int NW = 3 * RG, NT = 64 * NW;
    constexpr int BMK = 96 * RG;
    constexpr int IMGA = RG * W96_DYP * 1024;                      // the dy images of a stage
    constexpr int STAGE = IMGA + W96_XP * 1024;
    constexpr int XS = (W96_XP + NW - 1) / NW;                     // x slots per wave: 5 (three waves) / 3 (six)
    char *const dummy = smem + 2 * STAGE;                          // 1 KB every out-of-range x slot zero-fills (zeros over zeros)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / 3, wn = wave - 3 * wm;
    const int li = lane & 31, lh = lane >> 5;
    const int ntile = gridDim.x;
    const int lin = blockIdx.x + ntile * (blockIdx.y + gridDim.y * blockIdx.z);
    const int virt = xcd_remap(lin, ntile * gridDim.y * gridDim.z);
    const int tile = virt % ntile, sg = virt / ntile;
    const int grp = sg % gridDim.y, split = sg / gridDim.y;
    const int kt = tile / g.nCt, ct = tile - kt * g.nCt;
    const int nchunk = g.IC / 96;
    const int tap_r = ct / nchunk, chunk = ct - tap_r * nchunk;
    const int k0 = kt * BMK;
    const int OWp = g.OW + 2;
    const int Mp = g.Mtot;                                         // pixels of a group over the padded width (host: ng OH (OW + 2))
    const int m_begin = split * g.chunk;
    const int m_end = min(Mp, m_begin + g.chunk);
    const int ng = Mp / (g.OH * OWp);
    const int row0 = m_begin / OWp;                                // first (image, output row) of the split: offsets are relative to it
    const int img0 = row0 / g.OH;

    // ---- dy slots: piece p = wave + NW j, j = 0 .. 3, of the 12 RG pieces; piece p is image p / 12 (filters 96 (p / 12) ...), linear
    // 16-byte chunks 64 (p % 12) + lane of that image's 64 rows x 12 chunks.  The descriptor ends at the split's last real pixel:
    // positions past m_end read zeros without a test (a valid position q < m_end has a real index below that of m_end)
    const unsigned short *dyg = dy + ((size_t)grp * ng * g.OH + row0) * g.OW * g.K;
    const int row_e = m_end / OWp, owp_e = m_end - row_e * OWp;
    const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(dyg, ((size_t)(row_e - row0) * g.OW + (owp_e < g.OW ? owp_e : g.OW)) * g.K * 2);
    int a_owp[4], a_off[4];                                        // padded column; byte offset of (real pixel, filter chunk)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = wave + NW * j;
        const int rgi = p / W96_DYP;
        const int L = 64 * (p - rgi * W96_DYP) + lane;
        const int pix = L / 12, c = L - 12 * pix;
        const int q = m_begin + pix;
        const int row = q / OWp;
        const int ch = k0 + rgi * 96 + 8 * c;
        a_owp[j] = ch < g.K ? q - row * OWp : (1 << 30);           // a filter chunk beyond K: never a valid column
        a_off[j] = (((row - row0) * g.OW + q - row * OWp) * g.K + ch) * 2;
    }
    // ---- x slots: piece p = wave + NW j of the 13; linear chunk 64 p + lane of the image's 66 rows x 12 chunks
    const unsigned short *xg = x + ((size_t)grp * ng + img0) * g.IH * g.IW * g.IC;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(xg, (size_t)(ng - img0) * g.IH * g.IW * g.IC * 2);
    const int tap_h = tap_r * g.dh - g.ph;
    int b_owp[XS], b_oh[XS], b_base[XS];
#pragma unroll
    for (int j = 0; j < XS; ++j) {
        const int p = wave + NW * j;
        const int L = 64 * p + lane;
        const int e = L / 12, c = L - 12 * e;                      // image row, chunk
        const int q = m_begin + e;
        const int row = q / OWp;
        b_owp[j] = (p < W96_XP && e < 66) ? q - row * OWp : -(1 << 30);      // rows the image does not have: never a valid column
        const int n = row / g.OH;
        b_oh[j] = row - n * g.OH;
        b_base[j] = ((((n - img0) * g.IH + b_oh[j] + tap_h) * g.IW + q - row * OWp - g.pw) * g.IC + chunk * 96 + 8 * c) * 2;
    }
    auto issue = [&](int stage) {
        char *st = smem + stage * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned vo = (unsigned)a_owp[j] < (unsigned)g.OW ? (unsigned)a_off[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, lds_ptr(st + (wave + NW * j) * 1024), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < XS; ++j) {
            const bool ok = (unsigned)(b_oh[j] + tap_h) < (unsigned)g.IH && (unsigned)(b_owp[j] - g.pw) < (unsigned)g.IW;
            const unsigned vo = ok ? (unsigned)b_base[j] : OOB;
            // pieces past the 13th do not exist: those waves issue the same instruction out of range into the shared patch (no branch)
            char *dst = wave + NW * j < W96_XP ? st + IMGA + (wave + NW * j) * 1024 : dummy;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, lds_ptr(dst), 16, vo, 0, 0, 0);
        }
        // advance by 64 positions of the padded enumeration
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a_owp[j] += 64;
            a_off[j] += 64 * g.K * 2;
#pragma unroll
            for (int i = 0; i < WRAPS; ++i) {
                const bool w1 = a_owp[j] >= OWp;
                a_owp[j] -= w1 ? OWp : 0;
                a_off[j] -= w1 ? 2 * g.K * 2 : 0;                    // next row: + OW, column - (OW + 2)
            }
        }
#pragma unroll
        for (int j = 0; j < XS; ++j) {
            b_owp[j] += 64;
            b_base[j] += 64 * g.IC * 2;
#pragma unroll
            for (int i = 0; i < WRAPS; ++i) {
                const bool w1 = b_owp[j] >= OWp;
                b_owp[j] -= w1 ? OWp : 0;
                b_oh[j] += w1 ? 1 : 0;
                b_base[j] += w1 ? (g.IW - OWp) * g.IC * 2 : 0;
                const bool w2 = b_oh[j] >= g.OH;
                b_oh[j] -= w2 ? g.OH : 0;
                b_base[j] += w2 ? (g.IH - g.OH) * g.IW * g.IC * 2 : 0;
            }
        }
    };

    f32x16 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transposed-read lane constants (conv_wgrad_h2_body, 32x32x16): lane 4 q + p of a 16-lane group addresses pixel row q,
    // channels 4 p .. 4 p + 3; group gsel = (lane >> 4) & 1 -> channels + 16, lane half lh -> pixels + 8.  No swizzle (192-byte rows)
    const int trq = (lane & 15) >> 2, trp = lane & 3, gsel = (lane >> 4) & 1;
    int aad[3], bad[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) aad[t] = wm * W96_DYP * 1024 + (8 * lh + trq) * W96_ROWB + (t * 32 + 16 * gsel + 4 * trp) * 2;
#pragma unroll
    for (int sft = 0; sft < 3; ++sft) bad[sft] = IMGA + (8 * lh + trq + sft) * W96_ROWB + (wn * 32 + 16 * gsel + 4 * trp) * 2;
    const int nsteps = (m_end - m_begin + 63) / 64;
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    auto tr8 = [&](const char *p0, int row_off) {                   // 8 pixels of one channel per lane: two transposed reads
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)lds_ptr(p0 + row_off * W96_ROWB));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)lds_ptr(p0 + (row_off + 4) * W96_ROWB));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto compute = [&](int stage) {
        const char *base = smem + stage * STAGE;
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
            bf16x8 a[3], b[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) a[t] = tr8(base + aad[t], 16 * sx);
#pragma unroll
            for (int sft = 0; sft < 3; ++sft) b[sft] = tr8(base + bad[sft], 16 * sx);
#pragma unroll
            for (int tm = 0; tm < 3; ++tm)
#pragma unroll
                for (int tn = 0; tn < 3; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
    };
    // bias gradient (see h2_dy_colsum_step): column tile ct owns the 8-filter chunks ct, ct + nCt, ... of the 12 RG; item = (owned
    // chunk, pixel).  Inline asm: in front of a C++ LDS load the compiler would drain the DMA of the NEXT stage
    const bool sum_dy = bias_slab != nullptr && ct < 12 * RG;      // block-uniform; dy is zero at the padded positions
    float bs[W96_BSR][8] = {};
    auto colsum_step = [&](const char *img) {
#pragma unroll
        for (int r = 0; r < W96_BSR; ++r) {
            const int item = tid + NT * r;
            const int c16 = ct + g.nCt * (item >> 6), p = item & 63;
            if (c16 < 12 * RG) {                                   // wave-uniform: a wave's 64 items share the chunk
                const int rgi = c16 / 12;
                const unsigned a = (unsigned)reinterpret_cast<uintptr_t>(img + rgi * W96_DYP * 1024 + p * W96_ROWB + ((c16 - 12 * rgi) << 4));
                u32x4 v;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bs[r][2 * e] += __builtin_bit_cast(float, v[e] << 16);
                    bs[r][2 * e + 1] += __builtin_bit_cast(float, v[e] & 0xFFFF0000u);
                }
            }
        }
    };
    int st_c = 0;
    for (int t = 0; t < nsteps; ++t) {
        issue(st_c ^ 1);                 // past the end of the split: dy out of range (zeros); the x image is whatever follows
        compute(st_c);
        if (sum_dy) colsum_step(smem + st_c * STAGE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        st_c ^= 1;
    }
    if (sum_dy) {                        // the 64 pixels' sums of a filter are added in a fixed order (h2_dy_colsum_store)
        float *sp = reinterpret_cast<float *>(smem);               // [owned chunk][8 filters][64 pixels]: <= 24 KB RG
        float *bias_row = bias_slab + ((size_t)split * groups + grp) * g.K;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < W96_BSR; ++r) {
            const int item = tid + NT * r;
            const int slot = item >> 6, p = item & 63;
            if (ct + g.nCt * slot < 12 * RG) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sp[(slot * 8 + e) * 64 + p] = bs[r][e];
            }
        }
        __syncthreads();
        const int slot = tid >> 3, e = tid & 7;                    // 24 RG slots x 8 filters = NT threads >= the 12 RG chunks
        const int c16 = ct + g.nCt * slot;
        const int ch = c16 * 8 + e;
        if (c16 < 12 * RG && k0 + ch < g.K) {
            float tsum = 0.f;
            for (int q = 0; q < 64; ++q) tsum += sp[(slot * 8 + e) * 64 + q];
            bias_row[k0 + ch] = tsum;
        }
    }

    // slab [split][group][K][RSC]; the tile's columns (tap col s, channel) go to (tap, channel) of the channel-blocked filter layout
    float *o = out + ((size_t)split * groups + grp) * (size_t)g.K * g.RSC;
    const int cbs = chan_block(g.IC);
#pragma unroll
    for (int sft = 0; sft < 3; ++sft) {
        const int tap = tap_r * 3 + sft;
        const int col = kidx(tap, chunk * 96 + wn * 32 + li, g.RS, cbs);
#pragma unroll
        for (int tm = 0; tm < 3; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = k0 + wm * 96 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.K) o[(size_t)row * g.RSC + col] = acc[tm][sft][r];
            }
    }
}

template <int RG, int WRAPS>
__global__ __launch_bounds__(192 * RG, 2) void conv_wgrad_h2s96_kernel(const unsigned short *__restrict__ x, const unsigned short *__restrict__ dy,
                                                                       float *__restrict__ out, float *__restrict__ bias_slab, H2Wgrad g,
                                                                       int groups)
{
    __shared__ __attribute__((aligned(1024))) char smem[2 * (RG * W96_DYP + W96_XP) * 1024 + 1024];
    conv_wgrad_h2s96_body<RG, WRAPS>(x, dy, out, bias_slab, g, groups, smem);
}

// out[i] = sum_k slabs[k][i] in the fixed order k = 0, 1, ... (deterministic)
__global__ void h2_slab_reduce_kernel(const float *__restrict__ slabs, float *__restrict__ out, size_t n4, int splits)
{
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
}

// bias gradient of a bf16 dy (rows x cols): partial column sums per row chunk, then a fixed-order combine.
// grid (ceil(cols / 128), nchunks), block (16, 16): thread (cx, ry) sums 8 columns (one 16-byte load) of rows ry, ry + 16, ...
// of its chunk - 16 lanes cover 256 contiguous bytes of a row, so narrow layers (96 columns = 12 lanes) keep most lanes busy
__global__ __launch_bounds__(256) void h2_colsum_partial_kernel(const unsigned short *__restrict__ x, int64_t rows, int cols,
                                                                float *__restrict__ part, int nchunks)
{
    __shared__ float red[16][16][8];
    const int c8 = (blockIdx.x * 16 + threadIdx.x) * 8;
    const int64_t per = (rows + nchunks - 1) / nchunks;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(rows, r0 + per);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c8 < cols) {
        int64_t r = r0 + threadIdx.y;
        for (; r + 16 < r1; r += 32) {                 // two independent loads in flight
            const uint4 v = *reinterpret_cast<const uint4 *>(x + r * cols + c8);
            const uint4 u = *reinterpret_cast<const uint4 *>(x + (r + 16) * cols + c8);
            const unsigned w[4] = {v.x, v.y, v.z, v.w}, z[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[2 * i] += __builtin_bit_cast(float, w[i] << 16);
                s[2 * i + 1] += __builtin_bit_cast(float, w[i] & 0xFFFF0000u);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[2 * i] += __builtin_bit_cast(float, z[i] << 16);
                s[2 * i + 1] += __builtin_bit_cast(float, z[i] & 0xFFFF0000u);
            }
        }
        for (; r < r1; r += 16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(x + r * cols + c8);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[2 * i] += __builtin_bit_cast(float, w[i] << 16);
                s[2 * i + 1] += __builtin_bit_cast(float, w[i] & 0xFFFF0000u);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.y][threadIdx.x][i] = s[i];
    __syncthreads();
    if (threadIdx.y == 0 && c8 < cols)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x][i];
            part[(size_t)blockIdx.y * cols + c8 + i] = t;
        }
}
// grid (ceil(cols / 64)), block (64, 16): thread (c, ty) sums chunks ty, ty + 16, ...; fixed-order LDS combine
__global__ void h2_colsum_final_kernel(const float *__restrict__ part, int cols, int nchunks, float *__restrict__ out)
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

// ------------------------------------------------------------------------------------------ weight images
// src: f32 filter bank rows [rows][RS * Cin] in the channel-blocked order of common.h (kidx with CB = chan_block(Cin));
// dst: bf16 [rows][Kp], index (chunk * RS + tap) * CBdst + ci, zero beyond RS * Cin.
__global__ void h2_weight_image_kernel(const float *__restrict__ src, int64_t rows, int RS, int Cin, int CBsrc, int CBdst, int Kp,
                                       unsigned short *__restrict__ dst)
{
    const int64_t total = rows * Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / Kp;
        const int k = (int)(i - row * Kp);
        float v = 0.f;
        if (k < RS * Cin) {
            const int ch = k / (RS * CBdst), rem = k - ch * RS * CBdst;
            const int tap = rem / CBdst, ci = rem - tap * CBdst;
            v = src[row * (int64_t)(RS * Cin) + kidx(tap, ch * CBdst + ci, RS, CBsrc)];
        }
        dst[i] = f2bf(v);
    }
}

// ------------------------------------------------------------------------------------------ host side
static int h2_check(const mqConvDesc *d, const char *who)
{
    MQ_REQUIRE(d != nullptr, "%s: null descriptor", who);
    MQ_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0 && d->P > 0 && d->Q > 0,
               "%s: non-positive dimension", who);
    MQ_REQUIRE(d->stride_h > 0 && d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0 && d->pad_h >= 0 && d->pad_w >= 0,
               "%s: bad stride/dilation/padding", who);
    MQ_REQUIRE(d->groups > 0 && d->N % d->groups == 0, "%s: N=%d not divisible by groups=%d", who, d->N, d->groups);
    MQ_REQUIRE((int64_t)(d->P - 1) * d->stride_h - d->pad_h < d->H && (int64_t)(d->Q - 1) * d->stride_w - d->pad_w < d->W,
               "%s: output extent (%d,%d) inconsistent with input (%d,%d)", who, d->P, d->Q, d->H, d->W);
    MQ_REQUIRE((int64_t)d->N * d->H * d->W * d->C < (int64_t)1 << 40 && (int64_t)d->N * d->P * d->Q < INT32_MAX &&
               (int64_t)d->N * d->H * d->W < INT32_MAX, "%s: tensor too large", who);
    return 0;
}

static bool h2_shape_ok(const mqConvDesc *d)
{
    if (d->C % 32 != 0 || d->K % 32 != 0) return false;
    if (d->stride_h > 2 || d->stride_w > 2) return false;
    // 32-bit byte offsets inside the images one 256-row tile can touch (both passes)
    const int64_t img_x = (int64_t)d->H * d->W * d->C * 2, img_y = (int64_t)d->P * d->Q * d->K * 2;
    const int64_t t_x = cdiv(H2_BM, (int64_t)d->P * d->Q) + 2, t_y = cdiv(H2_BM, (int64_t)d->H * d->W) + 2;
    if (img_x * t_x >= 0x7FFFFFF0ll || img_y * t_y >= 0x7FFFFFF0ll) return false;
    if ((int64_t)std::max(d->K, d->C) * (int64_t)(cdiv((int64_t)d->R * d->S * std::max(d->C, d->K), 64) * 64) * 2 >= 0xFFFFFFF0ll) return false;
    return true;
}

static int h2_kp(int RS, int Cin) { return (int)(cdiv((int64_t)RS * Cin, 64) * 64); }

static int pick_tn(int ncols)
{
    // column tile 64 * TN that wastes the fewest MFMAs on padding, weighted by the relative efficiency of the tile
    // shapes (wider tiles reuse the A fragments more and need fewer bytes per flop)
    const int cand[4] = {4, 3, 2, 1};
    const double eff[4] = {1.0, 0.93, 0.85, 0.65};
    double best = 1e30;
    int tn = 4;
    for (int i = 0; i < 4; ++i) {
        const int bn = 64 * cand[i];
        const double cost = (double)cdiv(ncols, bn) * bn / eff[i];
        if (cost < best - 1e-9) { best = cost; tn = cand[i]; }
    }
    return tn;
}

// f32 engine on the same structure: reduction channels in blocks of 16 (the layout mq_weight_prep writes)
static bool d32_shape_ok(const mqConvDesc *d, bool dgrad)
{
    const int ic = dgrad ? d->K : d->C;
    if (ic % 16 != 0) return false;
    if (d->stride_h > 2 || d->stride_w > 2) return false;
    const int64_t img_x = (int64_t)d->H * d->W * d->C * 4, img_y = (int64_t)d->P * d->Q * d->K * 4;
    const int64_t t_x = cdiv(H2_BM, (int64_t)d->P * d->Q) + 2, t_y = cdiv(H2_BM, (int64_t)d->H * d->W) + 2;
    if ((dgrad ? img_y * t_y : img_x * t_x) >= 0x7FFFFFF0ll) return false;
    if ((int64_t)(dgrad ? d->C : d->K) * d->R * d->S * ic * 4 >= 0xFFFFFFF0ll) return false;
    if ((int64_t)(d->N / d->groups) * (dgrad ? (int64_t)d->H * d->W : (int64_t)d->P * d->Q) < 4 * H2_BM) return false;   // small problems: 128-row tiles
    return true;
}

// A/B switches of measured-and-rejected variants exist in the test-variant build only (make variants, -DMQ_VARIANTS=1): in the
// product library the variant's code is not compiled and its switch reads as "not set".
static inline bool variant_env(const char *name)
{
#if MQ_VARIANTS
    return getenv(name) != nullptr;
#else
    (void)name;
    return false;
#endif
}

template <class ET, bool DGRAD>
static int launch_h2(const mqConvDesc *d, const void *act, const void *w, void *out, const H2Epi &epi_in,
                     hipStream_t st, const char *who)
{
    if constexpr (ET::F32)
        MQ_REQUIRE(d32_shape_ok(d, DGRAD), "%s: shape not supported by the LDS-DMA f32 kernels", who);
    else
        MQ_REQUIRE(h2_shape_ok(d), "%s: shape not supported by the bf16-storage kernels (need C %% 32 == 0, K %% 32 == 0, strides <= 2; "
                   "got C=%d K=%d stride=(%d,%d)); ask mq_h2_supported first", who, d->C, d->K, d->stride_h, d->stride_w);
    MQ_REQUIRE((uintptr_t)act % 16 == 0 && (uintptr_t)w % 16 == 0, "%s: operands must be 16-byte aligned", who);
    H2Geom g;
    memset(&g, 0, sizeof(g));
    H2Parity pi;
    memset(&pi, 0, sizeof(pi));
    const int ng = d->N / d->groups;
    g.ng = ng;
    g.R = d->R; g.S = d->S; g.sh = d->stride_h; g.sw = d->stride_w; g.ph = d->pad_h; g.pw = d->pad_w; g.dh = d->dil_h; g.dw = d->dil_w;
    if (!DGRAD) {
        g.OH = d->P; g.OW = d->Q; g.IH = d->H; g.IW = d->W; g.IC = d->C; g.Ncols = d->K;
        g.mul_h = d->stride_h; g.mul_w = d->stride_w; g.add_h = -d->pad_h; g.add_w = -d->pad_w;
        g.step_h = d->dil_h; g.step_w = d->dil_w;
    } else {
        g.OH = d->H; g.OW = d->W; g.IH = d->P; g.IW = d->Q; g.IC = d->K; g.Ncols = d->C;
        g.mul_h = 1; g.mul_w = 1; g.add_h = d->pad_h; g.add_w = d->pad_w;
        g.step_h = -d->dil_h; g.step_w = -d->dil_w;
    }
    g.nR = g.R; g.nS = g.S;
    g.tap_r0 = 0; g.tap_rstep = 1; g.tap_s0 = 0; g.tap_sstep = 1;
    g.osh = 1; g.osw = 1; g.oa = 0; g.ob = 0; g.OHf = g.OH; g.OWf = g.OW;
    g.Mtot = ng * g.OH * g.OW;
    g.CB = ET::F32 ? 16 : (g.IC % 64 == 0 ? 64 : 32);
    g.nchunks = g.IC / g.CB;
    g.Kp = ET::F32 ? d->R * d->S * g.IC : h2_kp(d->R * d->S, g.IC);
    // MFMA shape of the bf16 kernels: 16x16x32 by default - same LDS bytes and cycles per flop as 32x32x16, but the chip
    // holds a higher clock on it (measured +3..13 % on the 384-1152-channel layers, tools/h2_micro.py); MQGAN_H2_MF32=1
    // switches back for A/B runs
    static const bool mf16 = !variant_env("MQGAN_H2_MF32");
    (void)mf16;
    int TN = pick_tn(g.Ncols);
#if MQ_VARIANTS
    if (const char *e = getenv("MQGAN_H2_TN")) {                 // experiments only
        const int v = atoi(e);
        if (v >= 1 && v <= 4) TN = v;
    }
#endif
    const int BN = 64 * TN;
    g.nMt = (int)cdiv(g.Mtot, H2_BM);
    g.nNt = (int)cdiv(g.Ncols, BN);
    const size_t act_gs = (size_t)ng * g.IH * g.IW * g.IC;
    const bool parity = DGRAD && (g.sh > 1 || g.sw > 1);
    H2Epi epi = epi_in;
    dim3 grid((unsigned)(g.nMt * g.nNt), d->groups), block(512);
    if (parity) {
        int t0 = 0;
        for (int a = 0; a < g.sh; ++a)
            for (int b = 0; b < g.sw; ++b) {
                const int c = pi.ncls++;
                pi.a[c] = a; pi.b[c] = b;
                pi.Ha[c] = g.OH > a ? (g.OH - a + g.sh - 1) / g.sh : 0;
                pi.Wb[c] = g.OW > b ? (g.OW - b + g.sw - 1) / g.sw : 0;
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
                t0 += (int)cdiv((int64_t)ng * pi.Ha[c] * pi.Wb[c], H2_BM);
            }
        pi.tile0[pi.ncls] = t0;
        grid = dim3((unsigned)(t0 * g.nNt), d->groups);
    }
#if MQ_VARIANTS
#define MQ_H2(TN_, NS_, CB_, PAR_) do { kname("conv_h2_kernel<%d, %d, %d, %s, %s>", TN_, NS_, CB_, tf(PAR_), tf(mf16)); if (mf16) hipLaunchKernelGGL((conv_h2_kernel<TN_, NS_, CB_, PAR_, true>), grid, block, 0, st, reinterpret_cast<const unsigned short *>(act), \
                                                      reinterpret_cast<const unsigned short *>(w), out, g, pi, epi, act_gs); \
                                        else hipLaunchKernelGGL((conv_h2_kernel<TN_, NS_, CB_, PAR_, false>), grid, block, 0, st, reinterpret_cast<const unsigned short *>(act), \
                                                      reinterpret_cast<const unsigned short *>(w), out, g, pi, epi, act_gs); } while (0)
#define MQ_D32(TN_, NS_, PAR_) do { kname("conv_d32_kernel<%d, %d, %s>", TN_, NS_, tf(PAR_)); hipLaunchKernelGGL((conv_d32_kernel<TN_, NS_, PAR_>), grid, block, 0, st, reinterpret_cast<const float *>(act), \
                                                  reinterpret_cast<const float *>(w), reinterpret_cast<float *>(out), g, pi, epi, act_gs); } while (0)
#else          // product: the 16x16x32 instantiations only, no f32 LDS-DMA kernels
#define MQ_H2(TN_, NS_, CB_, PAR_) do { kname("conv_h2_kernel<%d, %d, %d, %s, %s>", TN_, NS_, CB_, tf(PAR_), tf(true)); hipLaunchKernelGGL((conv_h2_kernel<TN_, NS_, CB_, PAR_, true>), grid, block, 0, st, reinterpret_cast<const unsigned short *>(act), \
                                                      reinterpret_cast<const unsigned short *>(w), out, g, pi, epi, act_gs); } while (0)
#define MQ_D32(TN_, NS_, PAR_) do { return ::mq::fail("%s: the LDS-DMA f32 kernels are in the test-variant library only", who); } while (0)
#endif
#define MQ_H2_CB(TN_, NS_, PAR_) do { if constexpr (ET::F32) MQ_D32(TN_, NS_, PAR_); else if (g.CB == 64) MQ_H2(TN_, NS_, 64, PAR_); else MQ_H2(TN_, NS_, 32, PAR_); } while (0)
#if MQ_VARIANTS
    static const bool ns2 = getenv("MQGAN_H2_NS2") != nullptr;      // experiment: TN = 2 on a two-stage ring (A/B of the ring depth)
#define MQ_H2_TN(PAR_) do { if (TN == 4) MQ_H2_CB(4, 2, PAR_); else if (TN == 3) MQ_H2_CB(3, 2, PAR_); \
                            else if (TN == 2) { if (ns2) MQ_H2_CB(2, 2, PAR_); else MQ_H2_CB(2, 3, PAR_); } else MQ_H2_CB(1, 2, PAR_); } while (0)
#else
#define MQ_H2_TN(PAR_) do { if (TN == 4) MQ_H2_CB(4, 2, PAR_); else if (TN == 3) MQ_H2_CB(3, 2, PAR_); \
                            else if (TN == 2) MQ_H2_CB(2, 3, PAR_); else MQ_H2_CB(1, 2, PAR_); } while (0)
#endif
    // three stride-1 column taps on 64-channel chunks: the shared-row kernel (one activation image per (chunk, tap row))
    const bool no_share = getenv("MQGAN_H2_NO_SHARE") != nullptr;      // read per call: tests switch it
    bool shared_rows = false;
    if constexpr (!ET::F32)
        shared_rows = mf16 && !no_share && !parity && g.CB == 64 && (g.S == 3 || g.S == 5 || g.S == 7) && g.sw == 1 && g.dw == 1 &&
                      g.sh == 1 && TN >= 2 && (int64_t)ng * g.OH * (g.OW + g.S - 1) < 0x7FFFFF00ll;
    // 96-channel chunks: the shared-row kernel on 192-byte rows with a 96-column tile (reduction channels a multiple of 96 but not
    // of 64, output columns that 96-wide tiles pad no more than the 64 TN ones)
    bool shared96 = false;
    if constexpr (!ET::F32)
        shared96 = mf16 && !no_share && getenv("MQGAN_H2_NO_96") == nullptr && !parity && g.CB == 32 && g.IC % 96 == 0 && g.S == 3 &&
                   g.sw == 1 && g.dw == 1 && g.sh == 1 && cdiv(g.Ncols, 96) * 96 <= cdiv(g.Ncols, BN) * BN &&
                   (int64_t)ng * g.OH * (g.OW + g.S - 1) < 0x7FFFFF00ll;
    // one to four K-steps (the 1x1 layers at the networks' edges): no ring to fill - 128-row tiles, one stage, three blocks per CU
    bool small_k = false;
    if constexpr (!ET::F32)
        small_k = mf16 && !parity && !shared_rows && !shared96 && getenv("MQGAN_H2_NO_SMALL") == nullptr &&
                  cdiv((int64_t)g.nchunks * g.R * g.S, g.CB == 64 ? 1 : 2) <= 4;
    // strided input gradients: inside one stride-parity class the taps are a stride-1 walk over dy - every class with two or more
    // column taps goes through the shared-row kernel on its own (class geometry in H2Geom), the others through the per-tap kernel
    bool shared_classes = false;
    if constexpr (!ET::F32)
        shared_classes = mf16 && !no_share && getenv("MQGAN_H2_NO_SHARE_PARITY") == nullptr && parity && g.CB == 64 && TN >= 2 &&
                         (int64_t)ng * g.OH * (g.OW + 8) < 0x7FFFFF00ll;
    if (shared_classes) {
        if constexpr (!ET::F32) {
            const int shift_h = g.sh == 2 ? 1 : 0, shift_w = g.sw == 2 ? 1 : 0;
            for (int c = 0; c < pi.ncls; ++c) {
                const int64_t Mc = (int64_t)ng * pi.Ha[c] * pi.Wb[c];
                if (Mc == 0) continue;
                const int ns = pi.ns[c], nr = pi.nr[c];
                const int stepw = -((pi.sstep[c] * g.dw) >> shift_w);
                if (ns >= 2 && ns <= 4 && nr >= 1 && stepw == -1) {
                    H2Geom gc = g;
                    gc.OH = pi.Ha[c]; gc.OW = pi.Wb[c];
                    gc.mul_h = 1; gc.mul_w = 1;
                    gc.add_h = (pi.a[c] + g.ph - pi.r0[c] * g.dh) >> shift_h;
                    gc.add_w = (pi.b[c] + g.pw - pi.s0[c] * g.dw) >> shift_w;
                    gc.step_h = -((pi.rstep[c] * g.dh) >> shift_h);
                    gc.step_w = stepw;
                    gc.nR = nr; gc.nS = ns;
                    gc.tap_r0 = pi.r0[c]; gc.tap_rstep = pi.rstep[c]; gc.tap_s0 = pi.s0[c]; gc.tap_sstep = pi.sstep[c];
                    gc.osh = g.sh; gc.osw = g.sw; gc.oa = pi.a[c]; gc.ob = pi.b[c]; gc.OHf = g.OH; gc.OWf = g.OW;
                    gc.nMt = (int)cdiv((int64_t)ng * gc.OH * (gc.OW + ns - 1), H2_BM);
                    const dim3 cgrid((unsigned)(gc.nMt * gc.nNt), d->groups);
#define MQ_H2SC(TN_, S_) do { kname("conv_h2s_kernel<%d, %d, true>", TN_, S_); hipLaunchKernelGGL((conv_h2s_kernel<TN_, S_, true>), cgrid, block, 0, st, reinterpret_cast<const unsigned short *>(act), \
                                        reinterpret_cast<const unsigned short *>(w), out, gc, epi, act_gs); } while (0)
#define MQ_H2SC_TN(S_) do { if (TN == 4) MQ_H2SC(4, S_); else if (TN == 3) MQ_H2SC(3, S_); else MQ_H2SC(2, S_); } while (0)
                    if (ns == 2) MQ_H2SC_TN(2); else if (ns == 3) MQ_H2SC_TN(3); else MQ_H2SC_TN(4);
#undef MQ_H2SC_TN
#undef MQ_H2SC
                } else {
                    H2Parity p1;
                    memset(&p1, 0, sizeof(p1));
                    p1.ncls = 1;
                    p1.a[0] = pi.a[c]; p1.b[0] = pi.b[c]; p1.Ha[0] = pi.Ha[c]; p1.Wb[0] = pi.Wb[c];
                    p1.nr[0] = pi.nr[c]; p1.ns[0] = pi.ns[c]; p1.r0[0] = pi.r0[c]; p1.s0[0] = pi.s0[c];
                    p1.rstep[0] = pi.rstep[c]; p1.sstep[0] = pi.sstep[c];
                    p1.tile0[0] = 0; p1.tile0[1] = (int)cdiv(Mc, H2_BM);
                    const dim3 cgrid((unsigned)(p1.tile0[1] * g.nNt), d->groups);
#define MQ_H2P1(TN_, NS_) do { kname("conv_h2_kernel<%d, %d, 64, true, true>", TN_, NS_); hipLaunchKernelGGL((conv_h2_kernel<TN_, NS_, 64, true, true>), cgrid, block, 0, st, \
                                        reinterpret_cast<const unsigned short *>(act), reinterpret_cast<const unsigned short *>(w), out, g, p1, epi, act_gs); } while (0)
                    if (TN == 4) MQ_H2P1(4, 2); else if (TN == 3) MQ_H2P1(3, 2); else MQ_H2P1(2, 3);
#undef MQ_H2P1
                }
            }
        }
    } else if (shared96) {
        if constexpr (!ET::F32) {
            const bool big = variant_env("MQGAN_H2_96_BIG");              // A/B runs and the identity test: the 256-row kernel
            g.nMt = (int)cdiv((int64_t)ng * g.OH * (g.OW + g.S - 1), big ? H2_BM : H96Q_BM);
            g.nNt = (int)cdiv(g.Ncols, H96_BN);
            grid = dim3((unsigned)(g.nMt * g.nNt), d->groups);
            if (big) {
#if MQ_VARIANTS
                kname("conv_h2s96_kernel<%d>", 3);
                hipLaunchKernelGGL((conv_h2s96_kernel<3>), grid, block, 0, st, reinterpret_cast<const unsigned short *>(act),
                                   reinterpret_cast<const unsigned short *>(w), out, g, epi, act_gs);
#endif
            } else {
                kname("conv_h2s96q_kernel<%d>", 3);
                hipLaunchKernelGGL((conv_h2s96q_kernel<3>), grid, dim3(256), 0, st, reinterpret_cast<const unsigned short *>(act),
                                   reinterpret_cast<const unsigned short *>(w), out, g, epi, act_gs);
            }
        }
    } else if (small_k) {
        if constexpr (!ET::F32) {
            g.nMt = (int)cdiv(g.Mtot, H2Q_BM);
            grid = dim3((unsigned)(g.nMt * g.nNt), d->groups);
#define MQ_H2Q(TN_, CB_) do { kname("conv_h2_small_kernel<%d, %d>", TN_, CB_); hipLaunchKernelGGL((conv_h2_small_kernel<TN_, CB_>), grid, dim3(256), 0, st, \
                                        reinterpret_cast<const unsigned short *>(act), reinterpret_cast<const unsigned short *>(w), out, g, epi, act_gs); } while (0)
#define MQ_H2Q_TN(CB_) do { if (TN == 4) MQ_H2Q(4, CB_); else if (TN == 3) MQ_H2Q(3, CB_); else if (TN == 2) MQ_H2Q(2, CB_); else MQ_H2Q(1, CB_); } while (0)
            if (g.CB == 64) MQ_H2Q_TN(64); else MQ_H2Q_TN(32);
#undef MQ_H2Q_TN
#undef MQ_H2Q
        }
    } else if (shared_rows) {
        if constexpr (!ET::F32) {
            g.nMt = (int)cdiv((int64_t)ng * g.OH * (g.OW + g.S - 1), H2_BM);
            grid = dim3((unsigned)(g.nMt * g.nNt), d->groups);
#define MQ_H2S(TN_, S_) do { kname("conv_h2s_kernel<%d, %d, false>", TN_, S_); hipLaunchKernelGGL((conv_h2s_kernel<TN_, S_>), grid, block, 0, st, reinterpret_cast<const unsigned short *>(act), \
                                        reinterpret_cast<const unsigned short *>(w), out, g, epi, act_gs); } while (0)
#define MQ_H2S_TN(S_) do { if (TN == 4) MQ_H2S(4, S_); else if (TN == 3) MQ_H2S(3, S_); else MQ_H2S(2, S_); } while (0)
            if (g.S == 3) MQ_H2S_TN(3); else if (g.S == 5) MQ_H2S_TN(5); else MQ_H2S_TN(7);
#undef MQ_H2S_TN
#undef MQ_H2S
        }
    } else if (parity) MQ_H2_TN(true);
    else MQ_H2_TN(false);
#undef MQ_H2_TN
#undef MQ_H2_CB
#undef MQ_D32
#undef MQ_H2
    MQ_LAUNCH_CHECK(who);
    return 0;
}

// hooks for igemm.hip's dispatcher (MQ_PRECISION_F32)
bool d32_nt_ok(const mqConvDesc *d, bool dgrad) { return d32_shape_ok(d, dgrad); }
#if MQ_VARIANTS
int d32_nt_launch(const mqConvDesc *d, bool dgrad, const float *act, const float *w, const float *bias, const uint8_t *mask, int d1,
                  int d2, int d3, int act_kind, float slope, float *out, hipStream_t st)
{
    H2Epi epi{bias, act_kind ? mask : nullptr, mask ? d1 : 1, mask ? d2 : 0, mask ? d3 : 1, act_kind, slope, 1};
    if (dgrad) return launch_h2<ElemF32, true>(d, act, w, out, epi, st, "mq_conv_dgrad(lds-dma)");
    return launch_h2<ElemF32, false>(d, act, w, out, epi, st, "mq_conv_fprop(lds-dma)");
}
#endif

}  // namespace mq

using namespace mq;

extern "C" int mq_h2_supported(const mqConvDesc *d)
{
    if (d == nullptr || d->precision != MQ_PRECISION_BF16) return 0;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0 || d->R <= 0 || d->S <= 0 || d->P <= 0 || d->Q <= 0 || d->groups <= 0) return 0;
    return h2_shape_ok(d) ? 1 : 0;
}

extern "C" size_t mq_h2_weight_elems(const mqConvDesc *d, int dgrad)
{
    if (d == nullptr || d->C <= 0 || d->K <= 0 || d->R <= 0 || d->S <= 0 || d->groups <= 0) return 0;
    const int rows = dgrad ? d->C : d->K, cin = dgrad ? d->K : d->C;
    return (size_t)d->groups * rows * h2_kp(d->R * d->S, cin);
}

extern "C" int mq_h2_weight_image(const mqConvDesc *d, int dgrad, const float *w, uint16_t *w_h2, mqStream stream)
{
    if (h2_check(d, "mq_h2_weight_image")) return 1;
    MQ_REQUIRE(w && w_h2, "mq_h2_weight_image: null pointer");
    const int rows = dgrad ? d->C : d->K, cin = dgrad ? d->K : d->C;
    MQ_REQUIRE(cin % 32 == 0, "mq_h2_weight_image: reduction channel count %d is not a multiple of 32", cin);
    const int RS = d->R * d->S, Kp = h2_kp(RS, cin);
    const int64_t nrows = (int64_t)d->groups * rows;
    hipLaunchKernelGGL(h2_weight_image_kernel, dim3(ew_grid(nrows * Kp, 256)), dim3(256), 0, S(stream), w, nrows, RS, cin,
                       chan_block(cin), cin % 64 == 0 ? 64 : 32, Kp, w_h2);
    MQ_LAUNCH_CHECK("mq_h2_weight_image");
    return 0;
}

extern "C" int mq_conv_fprop_h2(const mqConvDesc *d, const uint16_t *x, const uint16_t *w_h2, const float *bias,
                                const uint8_t *mask, int64_t d1, int64_t d2, int64_t d3, int act, float slope, int out_f32, void *y,
                                mqStream stream)
{
    if (h2_check(d, "mq_conv_fprop_h2")) return 1;
    MQ_REQUIRE(x && w_h2 && y, "mq_conv_fprop_h2: null pointer");
    MQ_REQUIRE(act == 0 || act == 2, "mq_conv_fprop_h2: act must be 0 (none) or 2 (LeakyReLU + mask)");
    MQ_REQUIRE(mask == nullptr || (d1 > 0 && d3 > 0 && d2 >= 0 && d1 < INT32_MAX && d2 < INT32_MAX && d3 < INT32_MAX),
               "mq_conv_fprop_h2: bad mask map");
    H2Epi epi{bias, act ? mask : nullptr, (int)(mask ? d1 : 1), (int)(mask ? d2 : 0), (int)(mask ? d3 : 1), act, slope, out_f32, nullptr};
    return launch_h2<ElemBF16, false>(d, x, w_h2, y, epi, S(stream), "mq_conv_fprop_h2");
}

extern "C" int mq_conv_dgrad_h2(const mqConvDesc *d, const uint16_t *dy, const uint16_t *wt_h2, int out_f32, void *dx, mqStream stream)
{
    if (h2_check(d, "mq_conv_dgrad_h2")) return 1;
    MQ_REQUIRE(dy && wt_h2 && dx, "mq_conv_dgrad_h2: null pointer");
    H2Epi epi{nullptr, nullptr, 1, 0, 1, 0, 0.f, out_f32, nullptr};
    return launch_h2<ElemBF16, true>(d, dy, wt_h2, dx, epi, S(stream), "mq_conv_dgrad_h2");
}

// mq_conv_dgrad_h2 for a conv whose bf16 input x came out of a LeakyReLU + masked_fill layer: returns the gradient with
// respect to that layer's pre-activation (see mq_conv_dgrad_gated); bf16 results are bit-identical to mq_conv_dgrad_h2
// followed by the pointwise LeakyReLU backward.
extern "C" int mq_conv_dgrad_h2_gated(const mqConvDesc *d, const uint16_t *dy, const uint16_t *wt_h2, const uint16_t *x,
                                      const uint8_t *mask, int64_t d1, int64_t d2, int64_t d3, float slope, int out_f32, void *dx,
                                      mqStream stream)
{
    if (h2_check(d, "mq_conv_dgrad_h2_gated")) return 1;
    MQ_REQUIRE(dy && wt_h2 && x && dx, "mq_conv_dgrad_h2_gated: null pointer");
    MQ_REQUIRE((uintptr_t)x % 16 == 0, "mq_conv_dgrad_h2_gated: x must be 16-byte aligned");
    MQ_REQUIRE(mask == nullptr || (d1 > 0 && d3 > 0 && d2 >= 0 && d1 < INT32_MAX && d2 < INT32_MAX && d3 < INT32_MAX),
               "mq_conv_dgrad_h2_gated: bad mask map");
    H2Epi epi{nullptr, mask, (int)(mask ? d1 : 1), (int)(mask ? d2 : 0), (int)(mask ? d3 : 1), 3, slope, out_f32, x};
    return launch_h2<ElemBF16, true>(d, dy, wt_h2, dx, epi, S(stream), "mq_conv_dgrad_h2_gated");
}

namespace mq {
static bool h2_wgrad_ok(const mqConvDesc *d)
{
    if (!h2_shape_ok(d)) return false;
    if (d->Q < 16) return false;                      // the pixel walk handles at most 4 row wraps per 64-pixel step
    return true;
}
static void h2_wgrad_plan(const mqConvDesc *d, H2Wgrad &g, int &wm, int &tm)
{
    const int ng = d->N / d->groups;
    g.OH = d->P; g.OW = d->Q; g.IH = d->H; g.IW = d->W; g.IC = d->C;
    g.S = d->S; g.sh = d->stride_h; g.sw = d->stride_w; g.ph = d->pad_h; g.pw = d->pad_w; g.dh = d->dil_h; g.dw = d->dil_w;
    g.Mtot = ng * d->P * d->Q;
    g.K = d->K;
    g.RS = d->R * d->S;
    g.RSC = g.RS * d->C;
    // row tile: 256 when K fills it, 192 for multiples of 192, else 128
    if (g.K % 256 == 0 || (g.K > 256 && g.K % 192 != 0)) { wm = 4; tm = 2; }
    else if (g.K % 192 == 0) { wm = 2; tm = 3; }
    else { wm = 2; tm = 2; }
    const int bmk = wm * 32 * tm;
    g.nKt = (int)cdiv(g.K, bmk);
    g.nCt = (int)cdiv(g.RSC, 256);
    const int64_t tiles = (int64_t)g.nKt * g.nCt * d->groups;
    const int64_t slots = 256;                        // one 512-thread block per CU
    int64_t max_splits = cdiv(g.Mtot, 64 * 16);       // >= 16 K-steps per split
    if (max_splits > 512) max_splits = 512;
    if (max_splits < 1) max_splits = 1;
    int64_t lo = slots / tiles, hi = 4 * slots / tiles + 1;
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
    {   // 32-bit byte offsets inside one split
        const int64_t pq = (int64_t)d->P * d->Q;
        const int64_t img = 2 * std::max((int64_t)d->H * d->W * d->C, pq * (int64_t)d->K);
        while (splits < max_splits && (cdiv(cdiv(g.Mtot, splits), pq) + 2) * img >= 0x7FFFFFF0ll) ++splits;
    }
    const int64_t chunk = cdiv(cdiv(g.Mtot, splits), 64) * 64;
    g.chunk = (int)chunk;
    g.splits = (int)cdiv(g.Mtot, chunk);
}
}  // namespace mq

extern "C" int mq_h2_wgrad_supported(const mqConvDesc *d)
{
    if (!mq_h2_supported(d)) return 0;
    return h2_wgrad_ok(d) ? 1 : 0;
}

// the shared-row kernel's layers and plan: pixels over the padded width, tiles of 192 rows x (tap row, 128-channel chunk)
static bool h2_wgrad_shared_shape(const mqConvDesc *d)
{
    return d->S == 3 && d->stride_w == 1 && d->stride_h == 1 && d->dil_w == 1 && d->C % 128 == 0 && d->K % 192 == 0 && d->Q >= 14;
}
// the 96-channel-chunk form (conv_wgrad_h2s96_kernel): what the 128-channel form does not take of the layers with C % 96 == 0 and
// K % 96 == 0; tiles of 96 RG rows x (tap row, 96-channel chunk), RG = 2 when K % 192 == 0
static bool h2_wgrad_shared96_shape(const mqConvDesc *d)
{
    return d->S == 3 && d->stride_w == 1 && d->stride_h == 1 && d->dil_w == 1 && d->C % 96 == 0 && d->K % 96 == 0 && d->Q >= 14 &&
           !h2_wgrad_shared_shape(d);
}
static void h2_wgrad_plan_shared(const mqConvDesc *d, H2Wgrad &g, int bmk = 192, int chk = 128, int64_t slots = 256)
{
    const int ng = d->N / d->groups;
    g.OH = d->P; g.OW = d->Q; g.IH = d->H; g.IW = d->W; g.IC = d->C;
    g.S = d->S; g.sh = d->stride_h; g.sw = d->stride_w; g.ph = d->pad_h; g.pw = d->pad_w; g.dh = d->dil_h; g.dw = d->dil_w;
    g.Mtot = ng * d->P * (d->Q + 2);
    g.K = d->K;
    g.RS = d->R * d->S;
    g.RSC = g.RS * d->C;
    g.nKt = (int)cdiv(g.K, bmk);
    g.nCt = d->R * (d->C / chk);
    const int64_t tiles = (int64_t)g.nKt * g.nCt * d->groups;
    int64_t max_splits = cdiv(g.Mtot, 64 * 16);
    if (max_splits > 512) max_splits = 512;
    if (max_splits < 1) max_splits = 1;
    int64_t lo = slots / tiles, hi = 4 * slots / tiles + 1;
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
    {   // 32-bit byte offsets inside one split (dy relative to the split's first row, x to its first image)
        const int64_t pq = (int64_t)d->P * (d->Q + 2);
        const int64_t img = 2 * std::max((int64_t)d->H * d->W * d->C, (int64_t)d->P * d->Q * (int64_t)d->K);
        while (splits < max_splits && (cdiv(cdiv(g.Mtot, splits), pq) + 2) * img >= 0x7FFFFFF0ll) ++splits;
    }
    const int64_t chunk = cdiv(cdiv(g.Mtot, splits), 64) * 64;
    g.chunk = (int)chunk;
    g.splits = (int)cdiv(g.Mtot, chunk);
}

extern "C" size_t mq_conv_wgrad_h2_workspace(const mqConvDesc *d)
{
    if (!mq_h2_wgrad_supported(d)) return 0;
    H2Wgrad g;
    int wm, tm;
    h2_wgrad_plan(d, g, wm, tm);
    if (h2_wgrad_shared_shape(d)) {
        H2Wgrad gs;                                  // either kernel may run (MQGAN_H2_NO_SHARE[_WGRAD]): room for the larger slab set
        h2_wgrad_plan_shared(d, gs);
        if (gs.splits > g.splits) g.splits = gs.splits;
    }
    if (h2_wgrad_shared96_shape(d)) {
        H2Wgrad gs;
        const int rg = d->K % 192 == 0 ? 2 : 1;
        h2_wgrad_plan_shared(d, gs, 96 * rg, 96, rg == 1 ? 512 : 256);
        if (gs.splits > g.splits) g.splits = gs.splits;
    }
    const size_t slabs = g.splits > 1 ? (size_t)g.splits * d->groups * g.K * g.RSC * sizeof(float) : 0;
    return slabs + (size_t)1024 * d->groups * g.K * sizeof(float);          // + column-sum partials of the bias gradient
}

extern "C" int mq_conv_wgrad_h2(const mqConvDesc *d, const uint16_t *x, const uint16_t *dy, float *dw, float *dbias, void *workspace,
                                size_t workspace_bytes, mqStream stream)
{
    if (h2_check(d, "mq_conv_wgrad_h2")) return 1;
    MQ_REQUIRE(x && dy && dw && workspace, "mq_conv_wgrad_h2: null pointer");
    MQ_REQUIRE(h2_wgrad_ok(d), "mq_conv_wgrad_h2: shape not supported (ask mq_h2_wgrad_supported first)");
    MQ_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)dw % 16 == 0, "mq_conv_wgrad_h2: 16-byte alignment");
    MQ_REQUIRE(workspace_bytes >= mq_conv_wgrad_h2_workspace(d), "mq_conv_wgrad_h2: workspace too small");
    H2Wgrad g;
    int wm = 0, tm = 0;
    const bool shared = h2_wgrad_shared_shape(d) && getenv("MQGAN_H2_NO_SHARE") == nullptr && getenv("MQGAN_H2_NO_SHARE_WGRAD") == nullptr;
    // MQGAN_H2_NO_WGRAD96=1: the 96-channel layers back on the gathered tiles of conv_wgrad_h2_kernel (A/B runs; read per call)
    const bool shared96 = h2_wgrad_shared96_shape(d) && getenv("MQGAN_H2_NO_SHARE") == nullptr && getenv("MQGAN_H2_NO_WGRAD96") == nullptr;
    const int rg96 = d->K % 192 == 0 ? 2 : 1;
    if (shared) h2_wgrad_plan_shared(d, g);
    else if (shared96) h2_wgrad_plan_shared(d, g, 96 * rg96, 96, rg96 == 1 ? 512 : 256);
    else h2_wgrad_plan(d, g, wm, tm);
    hipStream_t st = S(stream);
    const size_t n = (size_t)d->groups * g.K * g.RSC;
    float *target = g.splits > 1 ? reinterpret_cast<float *>(workspace) : dw;
    float *bias_ws = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + (g.splits > 1 ? (size_t)g.splits * n * sizeof(float) : 0));
    dim3 grid(g.nKt * g.nCt, d->groups, g.splits), block(512);
    // the bias gradient rides on the dy tiles the kernel stages anyway: per-split column sums into bias_ws [splits][groups][K]
    // (MQGAN_H2_BIAS_PASS=1: the separate column-sum passes over dy of r2-r4, for A/B runs)
    static const bool bias_pass = variant_env("MQGAN_H2_BIAS_PASS");
    float *bias_slab = (dbias && !bias_pass) ? bias_ws : nullptr;
    g.pipe = !variant_env("MQGAN_H2_WGRAD_NOPIPE");                  // read per call: A/B runs
    const int wraps = d->Q >= 64 ? 1 : 4;
    // the weight gradient stays on 32x32x16: its 16x16x32 instantiation (conflict-free too) measured 3-5 % slower
    // (twice the transposed LDS reads per flop); MQGAN_H2_WG16=1 selects it for A/B runs
#if MQ_VARIANTS
    static const bool mf16 = getenv("MQGAN_H2_WG16") != nullptr;
#define MQ_WH2(WM_, TM_, WR_) do { kname("conv_wgrad_h2_kernel<%d, %d, %d, %s>", WM_, TM_, WR_, tf(mf16)); if (mf16) hipLaunchKernelGGL((conv_wgrad_h2_kernel<WM_, TM_, WR_, true>), grid, block, 0, st, x, dy, target, bias_slab, g, d->groups); \
                                   else hipLaunchKernelGGL((conv_wgrad_h2_kernel<WM_, TM_, WR_, false>), grid, block, 0, st, x, dy, target, bias_slab, g, d->groups); } while (0)
#else
#define MQ_WH2(WM_, TM_, WR_) do { kname("conv_wgrad_h2_kernel<%d, %d, %d, %s>", WM_, TM_, WR_, tf(false)); hipLaunchKernelGGL((conv_wgrad_h2_kernel<WM_, TM_, WR_, false>), grid, block, 0, st, x, dy, target, bias_slab, g, d->groups); } while (0)
#endif
#define MQ_WH(WM_, TM_) do { if (wraps == 1) MQ_WH2(WM_, TM_, 1); else MQ_WH2(WM_, TM_, 4); } while (0)
    if (shared96) {
        const dim3 blk(192 * rg96);
#define MQ_W96(RG_, WR_) do { kname("conv_wgrad_h2s96_kernel<%d, %d>", RG_, WR_); hipLaunchKernelGGL((conv_wgrad_h2s96_kernel<RG_, WR_>), grid, blk, 0, st, x, dy, target, bias_slab, g, d->groups); } while (0)
        if (rg96 == 1) { if (d->Q + 2 >= 64) MQ_W96(1, 1); else MQ_W96(1, 4); }
        else { if (d->Q + 2 >= 64) MQ_W96(2, 1); else MQ_W96(2, 4); }
#undef MQ_W96
    } else if (shared) {
        if (d->Q + 2 >= 64) { kname("conv_wgrad_h2s_kernel<%d>", 1); hipLaunchKernelGGL((conv_wgrad_h2s_kernel<1>), grid, block, 0, st, x, dy, target, bias_slab, g, d->groups); }
        else { kname("conv_wgrad_h2s_kernel<%d>", 4); hipLaunchKernelGGL((conv_wgrad_h2s_kernel<4>), grid, block, 0, st, x, dy, target, bias_slab, g, d->groups); }
    } else if (wm == 4) MQ_WH(4, 2);
    else if (tm == 3) MQ_WH(2, 3);
    else MQ_WH(2, 2);
#undef MQ_WH
#undef MQ_WH2
    MQ_LAUNCH_CHECK("mq_conv_wgrad_h2");
    if (g.splits > 1) {
        MQ_REQUIRE(n % 4 == 0, "mq_conv_wgrad_h2: filter bank size not a multiple of 4");
        hipLaunchKernelGGL(h2_slab_reduce_kernel, dim3(ew_grid((int64_t)n / 4, 256)), dim3(256), 0, st, target, dw, n / 4, g.splits);
        MQ_LAUNCH_CHECK("mq_conv_wgrad_h2(reduce)");
    }
    if (bias_slab) {
        const int nb = d->groups * g.K;
        hipLaunchKernelGGL(h2_bias_reduce_kernel, dim3((unsigned)cdiv(nb, 256)), dim3(256), 0, st, bias_slab, g.splits, nb, dbias);
        MQ_LAUNCH_CHECK("mq_conv_wgrad_h2(bias)");
    } else if (dbias) {
        for (int grp = 0; grp < d->groups; ++grp) {
            const int64_t rows = (int64_t)(d->N / d->groups) * d->P * d->Q;      // real pixels (the shared-row plan counts padded ones)
            const int cols = g.K;
            int64_t want = cdiv(4096, cdiv(cols, 128));              // ~4096 blocks in total: 16 per CU
            if (want > rows / 64) want = rows / 64;
            if (want > 1024) want = 1024;
            if (want < 1) want = 1;
            float *ws = bias_ws + (size_t)grp * 1024 * cols;
            hipLaunchKernelGGL(h2_colsum_partial_kernel, dim3((unsigned)cdiv(cols, 128), (unsigned)want), dim3(16, 16), 0, st,
                               dy + (size_t)grp * rows * cols, rows, cols, ws, (int)want);
            hipLaunchKernelGGL(h2_colsum_final_kernel, dim3((unsigned)cdiv(cols, 64)), dim3(64, 16), 0, st, ws, cols, (int)want,
                               dbias + (size_t)grp * cols);
        }
        MQ_LAUNCH_CHECK("mq_conv_wgrad_h2(bias)");
    }
    return 0;
}
