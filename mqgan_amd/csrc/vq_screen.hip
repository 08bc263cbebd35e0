// L2 nearest-neighbour search past the f32-MFMA roof: bf16 candidate SCREENING + exact f32 re-rank
// ("bf16 MFMA for candidate scoring + exact fp32 re-rank of near-ties to keep indices exact").
// Results are bit-identical to mq_l2_argmin / oracle/vq_ref.c: same indices, same best distances.
//
// Definition (oracle/vq_ref.c):  dist(n,k) = fmaf(-2, dot(n,k), cn(k)),  dot = sequential f32 fma chain over d,
//                                idx(n) = argmin_k dist(n,k), lowest k on ties, NaN counts as -inf.
//
// Round 5: every bf16 score is formed ONCE and the search is THREE launches (r3-r4: five launches, every score formed twice -
// one pass for the row minima, one to emit the candidates against them).
//
// 1. prep   (l2_screen_prep_kernel)   x -> bf16 copy xh (RNE) and, per row, the bound E(n) (below); NaN marks rows without one.
// 2. screen (l2_screen_kernel)   a(n,k) = -cn(k)/2 + <xh_n, ch_k> on the bf16 MFMA, the norm term being the accumulator's initial
//              value (the MFMA's C operand: no distance epilogue), so that score(n,k) = -2 a(n,k) and the best code has the LARGEST
//              a.  The GEMM is formed TRANSPOSED (A = 128-code tile, B = 128-row tile, both staged by LDS-DMA into 128-byte
//              swizzled rows, two stages, two blocks per CU: 64 KB of loads in flight per CU is what the L2 -> LDS path needs to
//              run at its rate): a lane holds 32 codes of ONE row per wave tile, so row-wise work is lane-local.  After every
//              code tile a lane takes the running maximum of its row (its own codes, the other lane half by one DPP exchange, the
//              other code half of the block through an LDS word that is read one tile late - any value ever written there is a
//              lower bound of the final maximum) and appends every code with  a >= running maximum - E(n)  to a PRIVATE list
//              (row, code split, wave column, lane half): plain stores, a register counter, no atomics.  A single pass
//              over-collects - the first tiles are compared against a maximum that still rises: ~c H(tiles) entries per row
//              instead of c, c = 2-4 - and the re-rank drops what the final maximum excludes before it spends a chain on it.
//              (Tried first and measured: the block's x rows converted in-kernel and kept resident in LDS, 256-code tiles of the
//              codebook streamed past them - one launch fewer, half the DMA bytes per tile; but at 64-128 KB of LDS for the image
//              only ONE stage of 32 KB is in flight per CU and every K-step waits out the DMA latency: 303 us at 8192 x 512 against
//              ~130 us for this structure.  Bytes in flight decide, not bytes moved.)
// 3. re-rank (l2_rerank_kernel)  a block per 64 rows: the lists' entries are filtered against the FINAL maximum, the survivors
//              (2-4 per row) get their exact distance (the oracle's sequential fma chain, one lane per survivor, survivors
//              compacted so that the lanes of a wave are all busy) and meet in a packed (distance, index) minimum per row in LDS -
//              minimum distance, lowest index on ties; the block then writes idx / best_dist itself (no unpack launch).  Rows that
//              have no error bound (NaN / inf / huge), a list that ran over or more survivors than the block's pair table holds
//              are searched exhaustively with the exact chain by the same block - slow, rare, exact.
//
// Why the winner is always a candidate.  Let k* be the oracle's answer for row n and E(n) a bound with
// |score(n,k) - dist(n,k)| <= E(n) for EVERY k.  Then score(k*) <= dist(k*) + E <= dist(k) + E <= score(k) + 2E for all k, i.e.
// (score = -2 a)  a(k*) >= a(k) - E for all k, so a(k*) >= g - E for the final maximum g and for every running maximum (each is
// <= g): appended, and kept by the filter.  The threshold in a-units is therefore the WHOLE score bound E, not half of it: r5
// stored E / 2 (the 2 of "2E" and the 2 of "score = -2a" cancel; they were applied twice) and passed every random test because
// Cauchy-Schwarz leaves random data 10-30x inside the bound - tests/vq_adversarial.py builds the rows that sit at 0.6-0.67 E.  Every code with dist == dist(k*) satisfies the same inequalities, so the lowest such index is among
// the re-ranked survivors too; all other survivors have a larger exact distance.
//
// The bound (r6: from the rounding residuals the prep kernels measure, not from bf16's worst case).  With ex = x - xh and
// ec = c - ch (both exact in f32: xh is x with its low significand bits rounded away),
//   <x, c> - <xh, ch> = <ex, c> + <xh, ec>   exactly, so   |...| <= |ex_n| |c_k| + |xh_n| |ec_k|   (Cauchy-Schwarz, twice).
// bf16 products are exact in f32; the MFMA's f32 accumulation of [-cn/2, p_1 .. p_D] errs by at most D 2^-23 (cn/2 + sum|xh ch|),
// the oracle's chain by at most D 2^-24 sum|x c| (any summation order; the MFMA is given twice the IEEE figure), its final fmaf
// by 2^-24 (cn + 2 |x||c|).  Doubled for the score:
//   E(n) = 2.02 (|ex_n| cmax + |xh_n| ecmax) + 2.02 D 2^-21 |x_n| cmax + (2^-20 + D 2^-23) cmax^2 + 1e-30
// with cmax = max_k |c_k|, ecmax = max_k |ec_k| (1e-30: flushed denormals; every norm is taken 0.1 % high to cover its own
// rounding, |xh_n| as 1.004 |x_n|).  r3-r5 used the worst case |ex| <= 2^-8 |x|, |ec| <= 2^-8 |c|, i.e. 2.02 * 1.02 * 2^-7 |x| cmax:
// random data rounds with an rms residual of 0.41 * 2^-8, so the measured norms give a bound 2.2x smaller (1.5-1.9 survivors per
// row where the worst-case bound has 2.4-4) and lose nothing where the worst case is real (tests/vq_adversarial.py: residuals at
// half an ulp in every component - there the two bounds agree and the winner trails the best score by 0.69 E).  Rows with |x_n| or
// cmax above 1e18 or not finite have no bound (products may overflow): exhaustive search.
#include <stdlib.h>

#include "gemm_core.h"

namespace mq {

typedef __attribute__((address_space(3))) void lds_void_s;
__device__ __forceinline__ lds_void_s *lds_ptr_s(const void *p) { return reinterpret_cast<lds_void_s *>(reinterpret_cast<uintptr_t>(p)); }

constexpr int SC_CT = 128;            // codes per block tile (2 waves x 64)
constexpr int SC_RT = 128;            // rows per block tile (2 waves x 64)
constexpr int SC_ROWB = 128;          // bytes per LDS row: 64 bf16 k's
constexpr int SC_OP = 128 * SC_ROWB;  // one operand tile of a stage
constexpr int SC_STAGE = 2 * SC_OP;
constexpr int SC_LISTS = 4;           // private lists per (row, code split): 2 wave columns x 2 lane halves
constexpr int SC_PL = 16;             // entries per private list (64 bytes would overflow once per ~10^4 rows at 8192 codes); a longer list sends its row to the exhaustive search
constexpr int RR_ROWS = 64;           // rows per re-rank block
constexpr int RR_CAP = 1024;          // survivors a re-rank block holds
constexpr int RR_STAGE = 2 * 64 * 64;   // one stage of a wave's staging ring: 64 pairs x 16 f32 of the code row + of the x row
constexpr int RR_NS = 4;                // ring depth (a power of two)
constexpr int RR_PER_ROW = 32;        // ... and per row; rows with more are searched exhaustively

__device__ __forceinline__ unsigned ord_f32(float v)
{
    unsigned u = __builtin_bit_cast(unsigned, v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(unsigned u)
{
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __builtin_bit_cast(float, u);
}
__device__ __forceinline__ unsigned long long pack_di(float dist, int idx)
{
    return ((unsigned long long)ord_f32(dist) << 32) | (unsigned)idx;
}

// |v - bf16(v)|^2 of four elements (the difference is exact in f32; inf - inf and NaN give NaN, which poisons the bound as it must)
__device__ __forceinline__ float residual_sq(float4 v, uint2 h)
{
    const float e0 = v.x - __builtin_bit_cast(float, h.x << 16), e1 = v.y - __builtin_bit_cast(float, h.x & 0xffff0000u);
    const float e2 = v.z - __builtin_bit_cast(float, h.y << 16), e3 = v.w - __builtin_bit_cast(float, h.y & 0xffff0000u);
    return fmaf(e0, e0, fmaf(e1, e1, fmaf(e2, e2, e3 * e3)));
}

// ------------------------------------------------------------------------------------------ codebook side (once per codebook)
// one wave per code: bf16 copy, max |c|^2 over the codes
__global__ __launch_bounds__(256) void l2_screen_cbprep_kernel(const float *__restrict__ src, int rows, int d,
                                                               unsigned short *__restrict__ dst_h, unsigned *__restrict__ cmax2_bits)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *s = src + (size_t)row * d;
    unsigned short *o = dst_h + (size_t)row * d;
    float ss = 0.f, se = 0.f;
    for (int t = lane * 4; t < d; t += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(s + t);
        ss = fmaf(v.x, v.x, ss);
        ss = fmaf(v.y, v.y, ss);
        ss = fmaf(v.z, v.z, ss);
        ss = fmaf(v.w, v.w, ss);
        const uint2 h = pack_bf16x4(v);
        se += residual_sq(v, h);
        *reinterpret_cast<uint2 *>(o + t) = h;
    }
    ss = wave_sum(ss);
    se = wave_sum(se);
    // NaN orders above +inf as an unsigned pattern: a non-finite code poisons the bound of every row (-> exhaustive search)
    if (lane == 0) {
        atomicMax(cmax2_bits, __builtin_bit_cast(unsigned, ss));             // max |c|^2
        atomicMax(cmax2_bits + 1, __builtin_bit_cast(unsigned, se));         // max |c - bf16(c)|^2
    }
}

// sequential-chain code norms (the oracle's cn) and the accumulator seeds -cn / 2, padded to a multiple of the code tile with
// -inf (a padded code never rises above anything)
__global__ void l2_screen_norms_kernel(const float *__restrict__ c, int k, int k_pad, int d, float *__restrict__ cn,
                                       float *__restrict__ cnh)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k_pad) return;
    if (j >= k) {
        cn[j] = INFINITY;
        cnh[j] = -INFINITY;
        return;
    }
    float acc = 0.f;
    for (int t = 0; t < d; ++t) acc = __fmaf_rn(c[(size_t)j * d + t], c[(size_t)j * d + t], acc);
    cn[j] = acc;
    cnh[j] = -0.5f * acc;
}

// E(n) of the header comment from |x_n|^2, |x_n - xh_n|^2, max |c|^2, max |c - ch|^2; NaN = no bound
__device__ __forceinline__ float screen_bound(float ss, float se, float cmax2, float ecmax2, int d)
{
    const float nx = sqrtf(ss) * 1.001f, ex = sqrtf(se) * 1.001f;
    const float cm = sqrtf(cmax2) * 1.001f, ecm = sqrtf(ecmax2) * 1.001f;
    const float fd = (float)d;
    const float e = 2.02f * (ex * cm + 1.004f * nx * ecm) + 2.02f * (fd * 4.76837158203125e-07f) * nx * cm +     // 2^-21
                    (9.5367431640625e-07f + fd * 1.1920928955078125e-07f) * cm * cm + 1e-30f;                      // 2^-20, 2^-23
    return (nx < 1e18f && cm < 1e18f) ? e : __builtin_nanf("");
}

// ------------------------------------------------------------------------------------------ 1. prep (x side, every call)
// one wave per row: bf16 copy, |x_n|^2 -> E(n), the bound on |score - dist| = the threshold on a = -score / 2 (header comment);
// NaN = no bound
__global__ __launch_bounds__(256) void l2_screen_prep_kernel(const float *__restrict__ x, int64_t rows, int d,
                                                             unsigned short *__restrict__ xh, const float *__restrict__ cmax2,
                                                             float *__restrict__ bound, unsigned *__restrict__ gmax,
                                                             unsigned *__restrict__ exhaustive_rows)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) *exhaustive_rows = 0u;          // the re-rank launch counts them
    if (row >= rows) return;
    const float *s = x + row * d;
    unsigned short *o = xh + row * d;
    float ss = 0.f, se = 0.f;
    for (int t = lane * 4; t < d; t += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(s + t);
        ss = fmaf(v.x, v.x, ss);
        ss = fmaf(v.y, v.y, ss);
        ss = fmaf(v.z, v.z, ss);
        ss = fmaf(v.w, v.w, ss);
        const uint2 h = pack_bf16x4(v);
        se += residual_sq(v, h);
        *reinterpret_cast<uint2 *>(o + t) = h;
    }
    ss = wave_sum(ss);
    se = wave_sum(se);
    if (lane != 0) return;
    const float e2 = screen_bound(ss, se, cmax2[0], cmax2[1], d);
    const bool ok = e2 < 1e30f;                                                       // false for NaN too
    bound[row] = ok ? e2 : __builtin_nanf("");
    gmax[row] = ord_f32(-INFINITY);                    // the row's running maximum, shared by the code splits (atomicMax)
}

// ------------------------------------------------------------------------------------------ 2. screen
// grid (row tiles, code splits), 256 threads = 4 waves: wave (wc, wr) owns codes [64 wc, +64) x rows [64 wr, +64) of the
// block tile as 2 x 2 MFMA tiles of 32 x 32 (v_mfma_f32_32x32x16_bf16, A = codes, B = rows).  Both operand tiles
// (128 x 64 k's, 128-byte rows, 16-byte chunks XOR-swizzled with (row >> 1) & 7) are staged by LDS-DMA into a ring of two
// stages (lane-linear destination: the swizzle is applied to the per-lane SOURCE address); the x tile is re-streamed from L2
// for every code tile (D <= 512: 128 KB per block, it stays in L2; longer rows work and are tested, at whatever rate L2 misses leave).
__global__ __launch_bounds__(256, 2) void l2_screen_kernel(const unsigned short *__restrict__ xh, const unsigned short *__restrict__ ch,
                                                           const float *__restrict__ cnh, const float *__restrict__ bound, int64_t n, int k,
                                                           int d, int tiles_per_split, int splits, uint2 *__restrict__ cand,
                                                           unsigned char *__restrict__ cntb, unsigned *__restrict__ gmax)
{
    __shared__ __attribute__((aligned(1024))) char smem[2 * SC_STAGE];
    __shared__ float s_rm[2][SC_RT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wr = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * SC_RT;
    const int split = blockIdx.y;
    const int ntiles = (k + SC_CT - 1) / SC_CT;
    const int t_begin = split * tiles_per_split;
    const int t_end = min(ntiles, t_begin + tiles_per_split);
    const int nk = (d + 63) >> 6;
    const int total = nk * (t_end - t_begin);
    if (total <= 0) return;                                   // block-uniform (the layout gives every split a tile)

    // ---- DMA slots: piece p = wave + 4 j covers tile rows 8p .. 8p+7; lane -> row 8p + (lane >> 3), physical chunk lane & 7
    const int lc = (lane & 7) ^ ((((wave & 1) << 2) | (lane >> 4)) & 7);        // logical chunk this lane has to fetch
    const int64_t rows_here = n - m0 < SC_RT ? n - m0 : SC_RT;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xh + m0 * d, (size_t)rows_here * d * 2);
    const int64_t code0 = (int64_t)t_begin * SC_CT;
    const int64_t codes_here = min((int64_t)k, (int64_t)t_end * SC_CT) - code0;
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(ch + code0 * d, (size_t)codes_here * d * 2);
    unsigned rowoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rowoff[j] = (unsigned)(8 * (wave + 4 * j) + (lane >> 3)) * (unsigned)d * 2u + (unsigned)lc * 16u;
    auto issue = [&](int it) {
        const int tile = it / nk, ks = it - tile * nk;
        const bool kv = ks * 64 + lc * 8 < d;
        const unsigned koff = (unsigned)ks * 128u;
        const unsigned toff = (unsigned)tile * (unsigned)(SC_CT * 2) * (unsigned)d;
        char *st = smem + (it & 1) * SC_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, lds_ptr_s(st + (wave + 4 * j) * 1024), 16,
                                                     kv ? rowoff[j] + koff + toff : OOB, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, lds_ptr_s(st + SC_OP + (wave + 4 * j) * 1024), 16,
                                                     kv ? rowoff[j] + koff : OOB, 0, 0, 0);
    };
    issue(0);
    if (tid < SC_RT) { s_rm[0][tid] = -INFINITY; s_rm[1][tid] = -INFINITY; }

    // ---- fragment addresses: lane (li, lh) reads chunk (2 s + lh) ^ ((li >> 1) & 7) of row li of a 32-row MFMA tile
    const int sw = (li >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = li * SC_ROWB + ((((2 * s) ^ (lh ^ sw)) & 7) << 4);
    const int a_base = (wc * 64) * SC_ROWB, b_base = SC_OP + (wr * 64) * SC_ROWB;

    // accumulator seeds of a code tile: lane (li, lh) holds codes cbase + tm*32 + 8 (r >> 2) + 4 lh + (r & 3) -> float4 per (tm, r >> 2)
    float4 seed[2][4];
    auto load_seeds = [&](int tile) {
        const int cbase = (t_begin + tile) * SC_CT + wc * 64;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int q = 0; q < 4; ++q) seed[tm][q] = *reinterpret_cast<const float4 *>(cnh + cbase + tm * 32 + 8 * q + 4 * lh);
    };
    f32x16 acc[2][2];
    auto seed_acc = [&]() {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    acc[tm][tn][4 * q + 0] = seed[tm][q].x;
                    acc[tm][tn][4 * q + 1] = seed[tm][q].y;
                    acc[tm][tn][4 * q + 2] = seed[tm][q].z;
                    acc[tm][tn][4 * q + 3] = seed[tm][q].w;
                }
    };
    load_seeds(0);
    float rm[2], ebound[2], seen[2];
    int cnt[2];
    bool rvalid[2];
    int64_t lbase[2], grow[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int64_t row = m0 + wr * 64 + tn * 32 + li;
        rvalid[tn] = row < n;
        rm[tn] = -INFINITY;
        cnt[tn] = 0;
        lbase[tn] = ((row * splits + split) * SC_LISTS + wc * 2 + lh) * SC_PL;
        ebound[tn] = rvalid[tn] ? bound[row] : __builtin_nanf("");                // NaN: nothing compares >= against it
        grow[tn] = rvalid[tn] ? row : m0;
        seen[tn] = -INFINITY;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    seed_acc();
    if (nk < total) load_seeds(1);                            // the next tile's seeds travel while this tile multiplies
    __syncthreads();
    int tile = 0, ks = 0;
    for (int it = 0; it < total; ++it) {
        if (it + 1 < total) issue(it + 1);
        const char *st = smem + (it & 1) * SC_STAGE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = *reinterpret_cast<const bf16x8 *>(st + a_base + t * 32 * SC_ROWB + foff[s]);
                fb[t] = *reinterpret_cast<const bf16x8 *>(st + b_base + t * 32 * SC_ROWB + foff[s]);
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
        }
        if (++ks == nk) {
            // ---- fold the finished code tile: running maximum of the row, candidates into the lane's private list
            const int cbase = (t_begin + tile) * SC_CT + wc * 64;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                float tmax = -INFINITY;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) tmax = fmaxf(tmax, fmaxf(acc[tm][tn][r], acc[tm][tn][r + 1]));
                const float m = fmaxf(tmax, __shfl_xor(tmax, 32, 64));           // all 64 codes of this wave's slice
                const int rl = wr * 64 + tn * 32 + li;
                // what the other code half of this block had reached a tile ago, and what the other code splits had published in
                // the row's global word (any value ever written to either is <= the final maximum, stale reads included)
                rm[tn] = fmaxf(fmaxf(rm[tn], m), fmaxf(s_rm[wc ^ 1][rl], seen[tn]));
                if (lh == 0) {
                    s_rm[wc][rl] = rm[tn];
                    if (splits > 1 && rvalid[tn] && m > seen[tn]) atomicMax(&gmax[grow[tn]], ord_f32(rm[tn]));
                }
                const float thr = rm[tn] - ebound[tn];                            // NaN for rows without a bound
                if (__any(tmax >= thr)) {                                         // wave-uniform
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float a0 = acc[tm][tn][4 * q], a1 = acc[tm][tn][4 * q + 1], a2 = acc[tm][tn][4 * q + 2],
                                        a3 = acc[tm][tn][4 * q + 3];
                            if (fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)) >= thr) {
                                const float av[4] = {a0, a1, a2, a3};
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (av[e] >= thr && av[e] > -INFINITY) {      // -inf: padded codes
                                        if (cnt[tn] < SC_PL)
                                            cand[lbase[tn] + cnt[tn]] = make_uint2((unsigned)(cbase + tm * 32 + 8 * q + 4 * lh + e),
                                                                                   __builtin_bit_cast(unsigned, av[e]));
                                        ++cnt[tn];
                                    }
                            }
                        }
                }
            }
            seed_acc();
            ks = 0;
            ++tile;
            if ((tile + 1) * nk < total) load_seeds(tile + 1);
            if (splits > 1) {                                                     // what the other splits have reached: used at the next fold
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) seen[tn] = fmaxf(seen[tn], unord_f32(*reinterpret_cast<volatile const unsigned *>(gmax + grow[tn])));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // ---- list lengths (saturating) and the row maxima
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
        if (rvalid[tn]) cntb[lbase[tn] / SC_PL] = (unsigned char)(cnt[tn] > 255 ? 255 : cnt[tn]);
    __syncthreads();
    if (tid < SC_RT && m0 + tid < n) atomicMax(&gmax[m0 + tid], ord_f32(fmaxf(s_rm[0][tid], s_rm[1][tid])));
}

// ------------------------------------------------------------------------------------------ 3. re-rank, fallback
// dist(n,k) of the definition: the sequential f32 fma chain over d, then fmaf(-2, dot, cn).  One lane per pair.  The chain is
// a dependent fma per element and the rows come from L2 / HBM with microseconds of latency, so the two rows are streamed
// through a register ring of four stages of 16 floats each: 24 loads stay in flight while one stage's 16 chain steps run
// (measured on 40 k pairs of 256 elements: 64 us with one stage ahead and a device-scope fence per pair, 26 us without the
// fence, 23 us with this ring.  Two restructurings that looked better on paper were slower and are not kept: eight lanes
// per pair with coalesced 128-byte lines and the sum hopping between lanes by DPP - 43 us, eight times the waves, each
// still a chain of memory latencies; code rows staged through LDS by the whole block, a lane per pair chaining out of
// LDS - 29 us, the staging loads and the 64 dependent steps per stage do not overlap without an asynchronous copy.)
__device__ __forceinline__ float exact_dist(const float *__restrict__ xr, const float *__restrict__ cr, int d, float cnk)
{
    float acc = 0.f;
    int t = 0;
    if (d >= 64) {
        float4 a[4][4], b[4][4];
#pragma unroll
        for (int st = 0; st < 3; ++st)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[st][j] = *reinterpret_cast<const float4 *>(xr + 16 * st + 4 * j);
                b[st][j] = *reinterpret_cast<const float4 *>(cr + 16 * st + 4 * j);
            }
        // stage s of the ring holds elements [t, t + 16) when (t / 16) % 4 == s; unrolled by four so that indices are static
        for (; t + 64 <= d; t += 64) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int nxt = t + 16 * s4 + 48;                       // the stage three ahead goes into the slot freed last
                if (nxt + 16 <= d) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        a[(s4 + 3) & 3][j] = *reinterpret_cast<const float4 *>(xr + nxt + 4 * j);
                        b[(s4 + 3) & 3][j] = *reinterpret_cast<const float4 *>(cr + nxt + 4 * j);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc = __fmaf_rn(a[s4][j].x, b[s4][j].x, acc);
                    acc = __fmaf_rn(a[s4][j].y, b[s4][j].y, acc);
                    acc = __fmaf_rn(a[s4][j].z, b[s4][j].z, acc);
                    acc = __fmaf_rn(a[s4][j].w, b[s4][j].w, acc);
                }
            }
        }
        // up to three stages (48 elements) are loaded but not consumed when d % 64 != 0: the tail loop below re-reads them
    }
    for (; t < d; t += 4) {
        const float4 a = *reinterpret_cast<const float4 *>(xr + t);
        const float4 b = *reinterpret_cast<const float4 *>(cr + t);
        acc = __fmaf_rn(a.x, b.x, acc);
        acc = __fmaf_rn(a.y, b.y, acc);
        acc = __fmaf_rn(a.z, b.z, acc);
        acc = __fmaf_rn(a.w, b.w, acc);
    }
    float dist = __fmaf_rn(-2.0f, acc, cnk);
    return dist != dist ? -INFINITY : dist;
}


// Phases (c)-(e) of the re-rank for the RR_ROWS rows from r0 on: survivors (s_pair, s_np) -> exact chains -> packed minima per row
// (s_best), exhaustive rows (s_exh / no survivor), idx / best_dist.  Shared by l2_rerank_kernel (r5) and l2_search_kernel (r6).
// s_ring: 4 * RR_NS * RR_STAGE bytes of LDS, 1 KB aligned.  The caller has filled the tables and synchronised the block.
struct RerankTables {
    unsigned long long *s_best;      // [RR_ROWS]
    unsigned *s_rowcnt, *s_exh;      // [RR_ROWS]
    unsigned *s_np;                  // [1]
    unsigned long long *s_todo;      // [1]
    uint2 *s_pair;                   // [RR_CAP]
    unsigned long long *s_red;       // [4]
    char *s_ring;
};
__device__ __forceinline__ void rerank_finish(const RerankTables &T, const float *__restrict__ x, const float *__restrict__ cb,
                                              const float *__restrict__ cn, int64_t n, int k, int d, int64_t r0, int32_t *__restrict__ idx,
                                              float *__restrict__ best_dist, unsigned *__restrict__ exhaustive_rows)
{
    unsigned long long *s_best = T.s_best;
    unsigned *s_rowcnt = T.s_rowcnt, *s_exh = T.s_exh;
    uint2 *s_pair = T.s_pair;
    unsigned long long *s_red = T.s_red;
    char *s_ring = T.s_ring;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- (c) exact chains, one lane per survivor, operands staged through LDS by LDS-DMA.
    // A lane that fetched its own two rows from global memory touched a different cache line per lane and instruction, each line
    // four times (16 bytes at a time): 23-33 us at 16,384 rows - the texture addresser, not the chain, was the cost.  Here a wave
    // stages 16 elements of the code row and of the x row of each of its 64 pairs per step - 4 + 4 instructions of 1 KB, four lanes
    // on each pair's 64 bytes - into a wave-private ring of FOUR stages, three of them in flight (the chain is a string of DMA
    // latencies: with one stage in flight every 32 elements cost a whole latency, 26 us; what is in flight decides); the lane of
    // pair p reads its 64-byte slots back (16-byte parts XOR-swizzled with (p >> 2) & 3, applied to the per-lane SOURCE part, as
    // the DMA writes lane-linear) and runs the 16 chain steps.  No block barrier: a wave waits for its own DMAs only.
    {
        const unsigned np = min(*T.s_np, (unsigned)RR_CAP);
        const __amdgpu_buffer_rsrc_t rcb = make_rsrc(cb, (size_t)k * d * 4);
        const int64_t rows_here = n - r0 < RR_ROWS ? n - r0 : RR_ROWS;
        const __amdgpu_buffer_rsrc_t rxb = make_rsrc(x + r0 * d, (size_t)rows_here * d * 4);
        char *ring = s_ring + wave * (RR_NS * RR_STAGE);
        const int nch = (d + 15) >> 4;
        const int sub = lane >> 2, part = (lane & 3) ^ ((lane >> 4) & 3);   // this lane's pair within a DMA instruction, logical 16-byte part
        for (unsigned b0 = 0; b0 < np; b0 += 256) {                    // block-uniform
            const unsigned pbase = b0 + wave * 64;
            if (pbase >= np) continue;                                 // wave-uniform
            // the 4 pairs this lane fetches for (instruction i covers pairs 16 i .. 16 i + 15 of the wave's 64)
            unsigned coff[4], xoff[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned q = pbase + 16 * i + sub;
                const uint2 pr = q < np ? s_pair[q] : make_uint2(0u, 0u);
                coff[i] = q < np ? pr.y * (unsigned)d * 4u + (unsigned)part * 16u : OOB;
                xoff[i] = q < np ? pr.x * (unsigned)d * 4u + (unsigned)part * 16u : OOB;
            }
            auto issue = [&](int ch) {
                char *st = ring + (ch & (RR_NS - 1)) * RR_STAGE;
                const bool pv = ch * 16 + part * 4 < d;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rcb, lds_ptr_s(st + i * 1024), 16,
                                                             (pv && coff[i] != OOB) ? coff[i] + (unsigned)ch * 64u : OOB, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rxb, lds_ptr_s(st + 4096 + i * 1024), 16,
                                                             (pv && xoff[i] != OOB) ? xoff[i] + (unsigned)ch * 64u : OOB, 0, 0, 0);
            };
            const unsigned me = pbase + lane;
            const uint2 mine = me < np ? s_pair[me] : make_uint2(0u, 0u);
            const float cnk = cn[mine.y];                           // travels while the chain runs
            float acc = 0.f;
            issue(0);
            issue(1);
            issue(2);
            // No branch inside the loop (at the join of a branch the compiler drains its wait counters): the DMA of stage ch + 3 is
            // issued unconditionally - past the end of the rows its offsets are out of range and it writes zeros into the free
            // stage - and all four 16-byte parts are chained unconditionally: parts past d are zeros on both sides, and
            // fma(0, 0, acc) == acc.
            for (int ch = 0; ch < nch; ++ch) {
                issue(ch + 3);
                asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                 // stage ch landed, three later ones in flight
                // LDS reads as inline asm: in front of a C++ load from LDS the compiler waits for EVERY LDS-DMA in flight
                // (vmcnt(0) - it cannot tell the landed stage from the ones just requested), which made every stage pay the
                // whole DMA latency.  The wait that matters is the counted one above.
                const unsigned sb = (unsigned)reinterpret_cast<uintptr_t>(ring + (ch & (RR_NS - 1)) * RR_STAGE + lane * 64);
                f32x4v c4[4], x4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned a = sb + (unsigned)((j ^ ((lane >> 2) & 3)) << 4);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(c4[j]) : "v"(a));
                    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(x4[j]) : "v"(a));
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(c4[0]), "+v"(c4[1]), "+v"(c4[2]), "+v"(c4[3]), "+v"(x4[0]), "+v"(x4[1]), "+v"(x4[2]), "+v"(x4[3])
                             :
                             : "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc = __fmaf_rn(x4[j][0], c4[j][0], acc);
                    acc = __fmaf_rn(x4[j][1], c4[j][1], acc);
                    acc = __fmaf_rn(x4[j][2], c4[j][2], acc);
                    acc = __fmaf_rn(x4[j][3], c4[j][3], acc);
                }
                // (the reads above are back - their values were just used - before the next iteration's DMA overwrites stage ch - 1 ... )
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the trailing zero-fill must not outlive the batch
            if (me < np) {
                float dist = __fmaf_rn(-2.0f, acc, cnk);
                dist = dist != dist ? -INFINITY : dist;
                atomicMin(&s_best[mine.x], pack_di(dist, (int)mine.y));
            }
        }
    }
    __syncthreads();
    // ---- (d) exhaustive rows: a 64-bit mask first (the common case is an empty one), then a block-uniform loop over its bits
    if (wave == 0) {
        const bool need = r0 + lane < n && (s_exh[lane] != 0u || s_rowcnt[lane] == 0u);
        const unsigned long long mask = __ballot(need);
        if (lane == 0) *T.s_todo = mask;
    }
    __syncthreads();
    unsigned long long todo = *T.s_todo;
    while (todo) {
        const int rl = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int64_t row = r0 + rl;
        unsigned long long best = ~0ull;
        for (int code = tid; code < k; code += 256) {
            const unsigned long long p = pack_di(exact_dist(x + row * d, cb + (int64_t)code * d, d, cn[code]), code);
            best = p < best ? p : best;
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_xor(best, off, 64);
            best = o < best ? o : best;
        }
        if (lane == 0) s_red[wave] = best;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w) best = s_red[w] < best ? s_red[w] : best;
            s_best[rl] = best;                       // the exhaustive answer covers whatever the survivors had found
            atomicAdd(exhaustive_rows, 1u);
        }
        __syncthreads();
    }
    // ---- (e)
    if (tid < RR_ROWS && r0 + tid < n) {
        const unsigned long long p = s_best[tid];
        idx[r0 + tid] = (int32_t)(p & 0xffffffffu);
        if (best_dist) best_dist[r0 + tid] = unord_f32((unsigned)(p >> 32));
    }
}

// A block per RR_ROWS rows.  (a) the rows' state: final maximum over the code splits, bound, list lengths; (b) every list entry
// against the final maximum, survivors compacted into an LDS table (LDS atomics: a counter per block and per row); (c) one lane
// per survivor runs the exact chain and puts (distance, index) into the row's packed minimum in LDS; (d) rows without a bound,
// with a list that ran over, with more survivors than the table takes or - defensively - with none are searched exhaustively
// by the whole block, one after the other; (e) idx / best_dist.
__global__ __launch_bounds__(256) void l2_rerank_kernel(const float *__restrict__ x, const float *__restrict__ cb, const float *__restrict__ cn,
                                                        const uint2 *__restrict__ cand, const unsigned char *__restrict__ cntb,
                                                        const unsigned *__restrict__ gmax, const float *__restrict__ bound, int64_t n, int k,
                                                        int d, int splits, int32_t *__restrict__ idx, float *__restrict__ best_dist,
                                                        unsigned *__restrict__ exhaustive_rows)
{
    __shared__ unsigned long long s_best[RR_ROWS];
    __shared__ unsigned s_rowcnt[RR_ROWS];
    __shared__ unsigned s_exh[RR_ROWS];
    __shared__ unsigned s_np;
    __shared__ unsigned long long s_todo;
    __shared__ uint2 s_pair[RR_CAP];
    __shared__ unsigned long long s_red[4];
    __shared__ __attribute__((aligned(1024))) char s_ring[4 * RR_NS * RR_STAGE];  // per wave: four stages of (64 pairs x 64 B) x {code, x}
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * RR_ROWS;
    if (tid < RR_ROWS) {
        s_best[tid] = ~0ull;
        s_rowcnt[tid] = 0u;
        s_exh[tid] = 0u;
    }
    if (tid == 0) s_np = 0u;
    __syncthreads();
    // ---- (a) + (b) lists -> survivors.  One thread per (row, list); everything a thread needs first - the row's final maximum and
    // bound, the list's length - is requested at once, then the first two entries of every list (lists hold 1-2 entries on
    // average), before anything is looked at: a kernel of 64-row blocks is a string of memory latencies, and r5's first version
    // spent 16 us of them one after the other.
    const int lists = splits * SC_LISTS;
    for (int i0 = tid; i0 < RR_ROWS * lists; i0 += 4 * 256) {
        int rl_[4], c_[4];
        int64_t L_[4];
        float thr_[4];
        uint2 e0_[4], e1_[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 256;
            const bool v = i < RR_ROWS * lists;
            rl_[u] = v ? i / lists : 0;
            const int64_t row = r0 + rl_[u];
            L_[u] = (v && row < n) ? row * lists + (i - rl_[u] * lists) : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c_[u] = L_[u] >= 0 ? (int)cntb[L_[u]] : 0;
            thr_[u] = L_[u] >= 0 ? unord_f32(gmax[r0 + rl_[u]]) - bound[r0 + rl_[u]] : 0.f;      // NaN: no bound
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            e0_[u] = c_[u] > 0 ? cand[L_[u] * SC_PL] : make_uint2(0u, 0u);
            e1_[u] = c_[u] > 1 ? cand[L_[u] * SC_PL + 1] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (L_[u] < 0) continue;
            const int rl = rl_[u];
            const float thr = thr_[u];
            if (!(thr == thr) || c_[u] > SC_PL) {                   // no bound, or the list ran over: exhaustive search
                s_exh[rl] = 1u;
                continue;
            }
            for (int j = 0; j < c_[u]; ++j) {
                const uint2 ent = j == 0 ? e0_[u] : j == 1 ? e1_[u] : cand[L_[u] * SC_PL + j];
                if (__builtin_bit_cast(float, ent.y) >= thr) {
                    const unsigned inrow = atomicAdd(&s_rowcnt[rl], 1u);
                    const unsigned pos = inrow < (unsigned)RR_PER_ROW ? atomicAdd(&s_np, 1u) : (unsigned)RR_CAP;
                    if (pos < (unsigned)RR_CAP) s_pair[pos] = make_uint2((unsigned)rl, ent.x);
                    else s_exh[rl] = 1u;
                }
            }
        }
    }
    __syncthreads();
    const RerankTables T{s_best, s_rowcnt, s_exh, &s_np, &s_todo, s_pair, s_red, s_ring};
    rerank_finish(T, x, cb, cn, n, k, d, r0, idx, best_dist, exhaustive_rows);
}


// ============================================================================================== round 6: two launches (d <= 256; built for d <= 512)
// l2_search_kernel + l2_rerank2_kernel.  What changed against the three launches above, and why:
//   * x never exists as a bf16 tensor in HBM.  A block owns RT rows for its whole life: it loads them as f32, rounds them to bf16
//     into an LDS image (nk K-steps x RT rows x 128 bytes, the 16-byte chunk swizzle of the MFMA fragment reads), and sums |x|^2
//     and |x - bf16(x)|^2 on the way - the bound E(n) of the header comment costs nothing extra.  RT = 128 for d <= 256, 64 for
//     d <= 512 (64 KB of image either way).  l2_screen_prep_kernel's launch, its 24 MB round trip and the per-tile re-streaming of
//     the x tile (half of the DMA bytes of l2_screen_kernel) are gone.
//   * only code tiles are streamed: 128 codes x 64 k's = 16 KB per K-step by LDS-DMA into a ring of FOUR stages, three in flight.
//     l2_screen_kernel had one 32 KB stage in flight per block (two blocks per CU): at 1024 x 256 every K-step cost one DMA
//     latency - 20 us for 3.4 us of MFMA work.  The waits are COUNTED (vmcnt), which only holds if nothing else a wave
//     issues to vector memory is in flight between them: the accumulator seeds (-cn / 2) therefore travel with the ks = 0 stage of
//     a tile as a 512-byte DMA piece of wave 0, and the candidates stay in REGISTERS until the block's last K-step:
//   * per (row, lane list) the LAST SR_CAP candidates (a >= running maximum - E at the time) and `lost` = the largest score ever
//     pushed out of that window.  A list is complete if lost < g - E for the FINAL maximum g (checked by the re-rank; otherwise
//     the row is searched exhaustively): the running maximum only rises, so what an over-collecting pass pushes out is almost
//     always obsolete - it matters only when nine candidates of one lane's share of the codes lie within E of each other.  No list
//     stores, no atomics, no cross-split traffic inside the loop; the row maxima are per (row, split) - plain stores, nothing to
//     initialise, so no memset in front of the kernel.
//   * 512 threads: one block per CU (132 KB of LDS), so the block itself has to bring two waves per SIMD - with four waves every
//     K-step was five DMA issues + twenty fragment reads + sixteen MFMAs one after the other on each SIMD (measured: 1.4 us per
//     K-step at 1024 x 256 where the MFMAs are 0.2 us).
// Tried first and measured (tools/vq_stamp.py): the re-rank INSIDE the score kernel, run by the block that closes a row tile (arrival
// counter behind a device-scope fence): the two fences cost 25-30 us EACH (on this chip a device-scope release is an L2 write-back,
// as r3 found for a fence per candidate), and with one block per CU every row tile closes at the same moment, so nothing overlaps
// anyway: 88 / 157 / 370 us against 47 / 84 / 200 for the three launches.  Removed.
constexpr int SR_CT = 128;                 // codes per tile
constexpr int SR_NST = 4;                  // code stages (SR_NST - 1 in flight)
constexpr int SR_STAGE = SR_CT * 128 + 1024;   // 16 KB of codes + 512 B of seeds (+ 512 B the seed piece's idle lanes zero)
constexpr int SR_CAP = 8;                  // entries per lane list (five candidates within E of each other in one list are a 1e-6 event per
                                           // list, i.e. one exhaustive row per call at 16,384 rows x 8 lists with a window of four: measured)
constexpr int SR_NW = 8;                   // waves per block

template <int WC, int WR, int TM, int TN>  // waves along codes / rows, 32x32 MFMA tiles per wave along codes / rows
__global__ __launch_bounds__(512) void l2_search_kernel(const float *__restrict__ x, const unsigned short *__restrict__ ch, const float *__restrict__ cnh,
                                                        const float *__restrict__ cmax2, int64_t n, int k, int d, int tiles_per_split, int splits,
                                                        uint2 *__restrict__ cand, uint2 *__restrict__ head, float *__restrict__ gsplit,
                                                        float *__restrict__ bound, unsigned *__restrict__ exhaustive_rows)
{
    static_assert(WC * WR == SR_NW && WC * TM * 32 == SR_CT, "eight waves, 128 codes per tile");
    constexpr int RT = WR * TN * 32;       // rows per block
    constexpr int NPJ = RT / 64;           // 8-row pieces of the x tile per wave and K-step
    constexpr int LPR = WC * 2;            // lists per (row, split)
    extern __shared__ __attribute__((aligned(1024))) char dsm[];
    __shared__ float s_rm[WC][RT];
    __shared__ float s_bound[RT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WR, wr = wave % WR;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * RT;
    const int split = blockIdx.y;
    const int ntiles = (k + SR_CT - 1) / SR_CT;
    const int t_begin = split * tiles_per_split;
    const int t_end = min(ntiles, t_begin + tiles_per_split);
    const int nk = (d + 63) >> 6;
    const int total = nk * (t_end - t_begin);
    char *img = dsm;                                        // [nk][RT][128 B]
    char *stages = dsm + (((size_t)nk * RT * 128 + 1023) & ~(size_t)1023);
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *exhaustive_rows = 0u;     // the re-rank launch counts them

    // ---- code tiles by LDS-DMA: piece p = wave + 8 j (j = 0, 1) covers tile rows 8 p .. 8 p + 7; lane -> row 8 p + (lane >> 3), physical chunk lane & 7
    const int lc = (lane & 7) ^ ((((wave & 1) << 2) | (lane >> 4)) & 7);
    const int64_t code0 = (int64_t)t_begin * SR_CT;
    const int64_t codes_here = min((int64_t)k, (int64_t)t_end * SR_CT) - code0;
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(ch + code0 * d, (size_t)(codes_here > 0 ? codes_here : 0) * d * 2);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(cnh + code0, (size_t)(t_end - t_begin > 0 ? t_end - t_begin : 0) * SR_CT * 4);   // cnh is padded to whole tiles
    unsigned rowoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) rowoff[j] = (unsigned)(8 * (wave + 8 * j) + (lane >> 3)) * (unsigned)d * 2u + (unsigned)lc * 16u;
    int is_tile = 0, is_ks = 0;                            // (tile, K-step) of the next stage to issue
    auto issue = [&](int it) {
        char *st = stages + (it & (SR_NST - 1)) * SR_STAGE;
        const bool live = it < total;                       // past the end: out-of-range pieces (zeros into a free stage) keep the counts uniform
        const bool kv = live && is_ks * 64 + lc * 8 < d;
        const unsigned koff = (unsigned)is_ks * 128u;
        const unsigned toff = (unsigned)is_tile * (unsigned)(SR_CT * 2) * (unsigned)d;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, lds_ptr_s(st + (wave + 8 * j) * 1024), 16, kv ? rowoff[j] + koff + toff : OOB, 0, 0, 0);
        if (wave == 0)                                      // the tile's accumulator seeds ride with its first K-step (lanes 32-63 write zeros behind them)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds_ptr_s(st + SR_CT * 128), 16,
                                                     (live && is_ks == 0 && lane < 32) ? (unsigned)is_tile * (unsigned)(SR_CT * 4) + (unsigned)lane * 16u : OOB, 0, 0, 0);
        if (++is_ks == nk) { is_ks = 0; ++is_tile; }
    };
    if (total > 0) {
#pragma unroll
        for (int p = 0; p < SR_NST - 1; ++p) issue(p);
    }

    // ---- phase A: the block's rows, f32 -> bf16 image in LDS, |x|^2 and |x - xh|^2 on the way
    {
        const int c8 = lane & 7;
        float ss[NPJ], se[NPJ];
#pragma unroll
        for (int j = 0; j < NPJ; ++j) ss[j] = se[j] = 0.f;
        for (int ks = 0; ks < nk; ++ks) {
            float4 v0[NPJ], v1[NPJ];
            const bool kv = ks * 64 + c8 * 8 < d;
#pragma unroll
            for (int j = 0; j < NPJ; ++j) {
                const int row = 8 * (wave + 8 * j) + (lane >> 3);
                const bool ok = kv && m0 + row < n;
                const float *src = x + (m0 + row) * d + ks * 64 + c8 * 8;
                v0[j] = ok ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
                v1[j] = ok ? *reinterpret_cast<const float4 *>(src + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < NPJ; ++j) {
                const int row = 8 * (wave + 8 * j) + (lane >> 3);
                ss[j] = fmaf(v0[j].x, v0[j].x, fmaf(v0[j].y, v0[j].y, fmaf(v0[j].z, v0[j].z, fmaf(v0[j].w, v0[j].w, ss[j]))));
                ss[j] = fmaf(v1[j].x, v1[j].x, fmaf(v1[j].y, v1[j].y, fmaf(v1[j].z, v1[j].z, fmaf(v1[j].w, v1[j].w, ss[j]))));
                const uint2 h0 = pack_bf16x4(v0[j]), h1 = pack_bf16x4(v1[j]);
                se[j] += residual_sq(v0[j], h0) + residual_sq(v1[j], h1);
                *reinterpret_cast<uint4 *>(img + ((size_t)ks * RT + row) * 128 + ((c8 ^ ((row >> 1) & 7)) << 4)) = make_uint4(h0.x, h0.y, h1.x, h1.y);
            }
        }
#pragma unroll
        for (int j = 0; j < NPJ; ++j) {
            float a = ss[j], b = se[j];
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                a += __shfl_xor(a, o, 64);
                b += __shfl_xor(b, o, 64);
            }
            const int row = 8 * (wave + 8 * j) + (lane >> 3);
            if (c8 == 0) {
                const float e = screen_bound(a, b, cmax2[0], cmax2[1], d);
                const float eb = (m0 + row < n && e < 1e30f) ? e : __builtin_nanf("");
                s_bound[row] = eb;
                if (split == 0 && m0 + row < n) bound[m0 + row] = eb;
            }
        }
        if (tid < RT) {
#pragma unroll
            for (int w = 0; w < WC; ++w) s_rm[w][tid] = -INFINITY;
        }
    }
    __syncthreads();

    // ---- phase B: scores
    const int sw = (li >> 1) & 7;
    unsigned foff[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) foff[s4] = (unsigned)(li * 128 + ((((2 * s4) ^ (lh ^ sw)) & 7) << 4));
    const unsigned a_base = (unsigned)reinterpret_cast<uintptr_t>(stages) + (unsigned)(wc * TM * 32) * 128u;
    const unsigned b_base = (unsigned)reinterpret_cast<uintptr_t>(img) + (unsigned)(wr * TN * 32) * 128u;
    const unsigned seed_base = (unsigned)reinterpret_cast<uintptr_t>(stages) + (unsigned)(SR_CT * 128) + (unsigned)(wc * TM * 32 + 4 * lh) * 4u;
    const unsigned rm_base = (unsigned)reinterpret_cast<uintptr_t>(&s_rm[0][0]) + (unsigned)(wr * TN * 32 + li) * 4u;
    f32x16 acc[TM][TN];
    float rm[TN], ebound[TN], lostv[TN], la[TN][SR_CAP];
    unsigned lcd[TN][SR_CAP], lcnt[TN];
    bool rvalid[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int rl = wr * TN * 32 + tn * 32 + li;
        rvalid[tn] = m0 + rl < n;
        rm[tn] = -INFINITY;
        lostv[tn] = -INFINITY;
        lcnt[tn] = 0u;
        ebound[tn] = s_bound[rl];
#pragma unroll
        for (int j = 0; j < SR_CAP; ++j) { la[tn][j] = -INFINITY; lcd[tn][j] = 0xFFFFFFFFu; }
    }
    int it = 0;
    for (int tile = 0; tile < t_end - t_begin; ++tile) {
        // The accumulators are touched by MFMAs only inside the K loop (zeroed here, read by the fold): seeding them element by
        // element at ks == 0 made the compiler carry all of them through VGPR <-> AGPR copies in EVERY K-step (128 moves per 16
        // MFMAs in the first version of this kernel).  The seeds -cn / 2 are read at ks == 0 into registers and ADDED by the fold -
        // another summation order for the same sum, inside the bound E.
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
        f32x4v sd[TM][4];
        for (int ks = 0; ks < nk; ++ks, ++it) {
            // stage `it` has landed (the SR_NST - 2 younger ones may be in flight: 2 pieces each, 3 for wave 0 - waiting for one more
            // than needed is safe), and every wave is done with stage it - 1, whose buffer the next issue overwrites
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (SR_NST - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            issue(it + SR_NST - 1);
            const unsigned st_off = (unsigned)((it & (SR_NST - 1)) * SR_STAGE);
            if (ks == 0) {
                // the tile's seeds: lane (li, lh) holds codes tm*32 + 8 (r >> 2) + 4 lh + (r & 3) -> one float4 per (tm, r >> 2)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        asm volatile("ds_read_b128 %0, %1" : "=v"(sd[tm][q]) : "v"(seed_base + st_off + (unsigned)(tm * 32 + 8 * q) * 4u));
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sd[tm][0]), "+v"(sd[tm][1]), "+v"(sd[tm][2]), "+v"(sd[tm][3])::"memory");
            }
            const unsigned a_st = a_base + st_off, b_st = b_base + (unsigned)(ks * RT) * 128u;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int t = 0; t < TM; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[t]) : "v"(a_st + (unsigned)(t * 32 * 128) + foff[s4]));
#pragma unroll
                for (int t = 0; t < TN; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(fb[t]) : "v"(b_st + (unsigned)(t * 32 * 128) + foff[s4]));
                // every fragment register is tied to the wait ONCE (the same variable twice would make the compiler copy a register
                // whose data has not landed yet)
                static_assert(TM == 1 && (TN == 1 || TN == 2), "fragment ties below");
                if constexpr (TN == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fb[0]), "+v"(fb[1])::"memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fb[0])::"memory");
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm], fb[tn], acc[tm][tn], 0, 0, 0);
            }
        }
        {
            // scores of the tile: accumulator + seed
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tm][tn][r] += sd[tm][r >> 2][r & 3];
            // ---- fold the finished code tile: running maximum of the row, candidates into the lane's register window
            const unsigned cbase = (unsigned)((t_begin + tile) * SR_CT + wc * TM * 32);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                float tmax = -INFINITY;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) tmax = fmaxf(tmax, fmaxf(acc[tm][tn][r], acc[tm][tn][r + 1]));
                const float m = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                // what the other code slices of the block had reached a tile ago (a lower bound of g).  LDS access by inline asm: in
                // front of a C++ load from LDS the compiler drains every LDS-DMA in flight (vmcnt(0)) - the ring's whole point
                float other = -INFINITY;
#pragma unroll
                for (int w = 0; w < WC; ++w) {
                    float o;
                    asm volatile("ds_read_b32 %0, %1" : "=v"(o) : "v"(rm_base + (unsigned)((w * RT + tn * 32) * 4)));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o)::"memory");
                    other = fmaxf(other, o);
                }
                rm[tn] = fmaxf(fmaxf(rm[tn], m), other);
                if (lh == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(rm_base + (unsigned)((wc * RT + tn * 32) * 4)), "v"(rm[tn]) : "memory");
                const float thr = rm[tn] - ebound[tn];        // NaN for rows without a bound: nothing compares >= against it
                if (__any(tmax >= thr)) {                    // wave-uniform
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float a0 = acc[tm][tn][4 * q], a1 = acc[tm][tn][4 * q + 1], a2 = acc[tm][tn][4 * q + 2], a3 = acc[tm][tn][4 * q + 3];
                            if (fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)) >= thr) {
                                const float av[4] = {a0, a1, a2, a3};
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (av[e] >= thr) {
                                        // the lane's list holds its LAST SR_CAP candidates; what it pushes out is remembered by value
                                        lostv[tn] = fmaxf(lostv[tn], la[tn][SR_CAP - 1]);
#pragma unroll
                                        for (int j = SR_CAP - 1; j > 0; --j) { la[tn][j] = la[tn][j - 1]; lcd[tn][j] = lcd[tn][j - 1]; }
                                        la[tn][0] = av[e];
                                        lcd[tn][0] = cbase + (unsigned)(tm * 32 + 8 * q + 4 * lh + e);
                                        ++lcnt[tn];
                                    }
                            }
                        }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the trailing out-of-range pieces must not outlive the ring
    // ---- the lists and the row maxima of this (row tile, split) -> global memory
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int rl = wr * TN * 32 + tn * 32 + li;
        if (rvalid[tn]) {
            const int64_t L = (((m0 + rl) * splits + split) * LPR + wc * 2 + lh);
            const unsigned c = lcnt[tn] < (unsigned)SR_CAP ? lcnt[tn] : (unsigned)SR_CAP;       // the window's newest c entries are slots 0 .. c - 1
#pragma unroll
            for (int j = 0; j < SR_CAP; ++j)
                if ((unsigned)j < c) cand[L * SR_CAP + j] = make_uint2(lcd[tn][j], __builtin_bit_cast(unsigned, la[tn][j]));
            head[L] = make_uint2(__builtin_bit_cast(unsigned, lostv[tn]), c);
        }
    }
    __syncthreads();
    if (tid < RT && m0 + tid < n) {
        float g = -INFINITY;
#pragma unroll
        for (int w = 0; w < WC; ++w) g = fmaxf(g, s_rm[w][tid]);
        gsplit[(m0 + tid) * splits + split] = g;
    }
}

// The re-rank of the two-launch path: a block per RR_ROWS rows.  Final maximum of a row = the largest of its splits' maxima; a list
// whose pushed-out score reaches g - E sends the row to the exhaustive search; the windows' entries against g - E -> survivors ->
// rerank_finish (exact chains, packed minima, idx / best_dist).
__global__ __launch_bounds__(256) void l2_rerank2_kernel(const float *__restrict__ x, const float *__restrict__ cb, const float *__restrict__ cn,
                                                         const uint2 *__restrict__ cand, const uint2 *__restrict__ head, const float *__restrict__ gsplit,
                                                         const float *__restrict__ bound, int64_t n, int k, int d, int splits, int lpr,
                                                         int32_t *__restrict__ idx, float *__restrict__ best_dist, unsigned *__restrict__ exhaustive_rows)
{
    __shared__ unsigned long long s_best[RR_ROWS];
    __shared__ unsigned s_rowcnt[RR_ROWS];
    __shared__ unsigned s_exh[RR_ROWS];
    __shared__ float s_thr[RR_ROWS];
    __shared__ unsigned s_np;
    __shared__ unsigned long long s_todo;
    __shared__ uint2 s_pair[RR_CAP];
    __shared__ unsigned long long s_red[4];
    __shared__ __attribute__((aligned(1024))) char s_ring[4 * RR_NS * RR_STAGE];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * RR_ROWS;
    if (tid < RR_ROWS) {
        s_best[tid] = ~0ull;
        s_rowcnt[tid] = 0u;
        s_exh[tid] = 0u;
        float thr = __builtin_nanf("");
        if (r0 + tid < n) {
            float g = -INFINITY;
            for (int sp = 0; sp < splits; ++sp) g = fmaxf(g, gsplit[(r0 + tid) * splits + sp]);
            thr = g - bound[r0 + tid];                          // NaN: no bound
        }
        s_thr[tid] = thr;
    }
    if (tid == 0) s_np = 0u;
    __syncthreads();
    const int lists = splits * lpr;
    for (int i = tid; i < RR_ROWS * lists; i += 256) {
        const int rl = i / lists;
        const int64_t row = r0 + rl;
        if (row >= n) continue;
        const int64_t L = row * lists + (i - rl * lists);
        const uint2 hd = head[L];
        const float thr = s_thr[rl];
        if (!(thr == thr) || __builtin_bit_cast(float, hd.x) >= thr) {       // no bound, or an entry that may matter fell out of this window
            s_exh[rl] = 1u;
            continue;
        }
        const int cnt = (int)hd.y;
        for (int j = 0; j < cnt; ++j) {
            const uint2 ent = cand[L * SR_CAP + j];
            if (__builtin_bit_cast(float, ent.y) >= thr) {
                const unsigned inrow = atomicAdd(&s_rowcnt[rl], 1u);
                const unsigned pos = inrow < (unsigned)RR_PER_ROW ? atomicAdd(&s_np, 1u) : (unsigned)RR_CAP;
                if (pos < (unsigned)RR_CAP) s_pair[pos] = make_uint2((unsigned)rl, ent.x);
                else s_exh[rl] = 1u;
            }
        }
    }
    __syncthreads();
    const RerankTables T{s_best, s_rowcnt, s_exh, &s_np, &s_todo, s_pair, s_red, s_ring};
    rerank_finish(T, x, cb, cn, n, k, d, r0, idx, best_dist, exhaustive_rows);
}

struct ScreenLayout {              // byte offsets inside the prepared-codebook blob and the workspace
    size_t cn, cnh, cmax2, ch, blob;
    size_t cand, cntb, gmax, bound, counters, xh, ws;
    int k_pad, splits, tiles_per_split;
};
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static ScreenLayout screen_layout(int64_t n, int k, int d)
{
    ScreenLayout L;
    L.k_pad = (int)(cdiv(k, SC_CT) * SC_CT);
    L.cn = 0;
    L.cnh = al256((size_t)L.k_pad * 4);
    L.cmax2 = L.cnh + al256((size_t)L.k_pad * 4);
    L.ch = L.cmax2 + 256;
    L.blob = L.ch + al256((size_t)k * d * 2);
    // code splits: one round of 2 blocks per CU (every block of a launch has the same work); fewer splits = longer private lists
    // of one running maximum each, i.e. less over-collection
    const int64_t row_tiles = cdiv(n > 0 ? n : 1, SC_RT);
    const int ntiles = (int)cdiv(k, SC_CT);
    int splits = (int)cdiv(512, row_tiles);
    if (splits > ntiles) splits = ntiles;
    if (splits < 1) splits = 1;
    L.tiles_per_split = (int)cdiv(ntiles, splits);
    L.splits = (int)cdiv(ntiles, L.tiles_per_split);
    size_t o = 0;
    L.cand = o; o += al256((size_t)n * L.splits * SC_LISTS * SC_PL * 8);
    L.cntb = o; o += al256((size_t)n * L.splits * SC_LISTS);
    L.gmax = o; o += al256((size_t)n * 4);
    L.bound = o; o += al256((size_t)n * 4);
    L.counters = o; o += 256;
    L.xh = o; o += al256((size_t)n * d * 2);
    L.ws = o;
    return L;
}

struct SearchLayout {              // workspace of the one-kernel path (l2_search_kernel)
    size_t cand, head, gsplit, bound, counters, ws, dyn_lds;
    int splits, tiles_per_split, rt, lpr;
};
static SearchLayout search_layout(int64_t n, int k, int d)
{
    SearchLayout L;
    L.rt = d <= 256 ? 128 : 64;
    L.lpr = 4 * 2;                                               // four waves along the codes, two lane halves
    const int64_t row_tiles = cdiv(n > 0 ? n : 1, L.rt);
    const int ntiles = (int)cdiv(k, SR_CT);
    // one block per CU (130 KB of LDS): enough code splits for one round of 256 blocks, no more - a split re-converts its rows
    int splits = (int)cdiv(256, row_tiles);
    if (splits > ntiles) splits = ntiles;
    if (splits < 1) splits = 1;
    L.tiles_per_split = (int)cdiv(ntiles, splits);
    L.splits = (int)cdiv(ntiles, L.tiles_per_split);
    size_t o = 0;
    L.cand = o; o += al256((size_t)n * L.splits * L.lpr * SR_CAP * 8);
    L.head = o; o += al256((size_t)n * L.splits * L.lpr * 8);            // per list: {largest score pushed out of the window, entries}
    L.gsplit = o; o += al256((size_t)n * L.splits * 4);          // per (row, split): the largest score
    L.bound = o; o += al256((size_t)n * 4);
    L.counters = o; o += 256;
    L.ws = o;
    const size_t nk = (size_t)(d + 63) / 64;
    const size_t main_lds = ((nk * L.rt * 128 + 1023) & ~(size_t)1023) + (size_t)SR_NST * SR_STAGE;
    L.dyn_lds = main_lds;
    return L;
}

}  // namespace mq

using namespace mq;

static bool screen_shape_ok(int64_t n, int k, int d)
{
    return n >= 0 && n < ((int64_t)1 << 31) && k > 0 && d > 0 && d % 8 == 0 && (int64_t)SC_RT * d * 2 < 0x7FFFFFF0ll &&
           (int64_t)k * d * 4 < 0xFFFFFFF0ll;
}

extern "C" int mq_l2_screen_supported(int64_t n, int k, int d) { return screen_shape_ok(n, k, d) ? 1 : 0; }

extern "C" size_t mq_l2_screen_codebook_bytes(int k, int d) { return screen_shape_ok(0, k, d) ? screen_layout(0, k, d).blob : 0; }

// d <= 256: the two-launch search (round 6); longer rows keep the three launches of round 5 (MQGAN_VQ_3LAUNCH=1 forces them: A/B runs and
// the bit-identity test of the two paths)
// Measured (16,384 rows, same box, profiles/r6_vq_*): 1024 x 256  42.8 us against 46.8 (score kernel 24 us where prep + screen took 26, and one
// launch fewer); 2048 x 512  92.7 against 83.5, 8192 x 512  254 against 200 - with 64-row tiles (the 64 KB image limit at d = 512) a
// wave's 32 x 32 tile reads two LDS fragments per MFMA and the kernel is LDS-bound where l2_screen_kernel's 64 x 64 wave tiles are
// not.  So: d <= 256 takes the two launches, longer rows keep the three.  (MQGAN_VQ_2LAUNCH_MAXD overrides the limit for A/B runs, up
// to 512.)
static int search_maxd()
{
    const char *e = getenv("MQGAN_VQ_2LAUNCH_MAXD");             // read per call: tests switch it
    const int q = e ? atoi(e) : 256;
    return q < 0 ? 0 : (q > 512 ? 512 : q);
}
static bool search_one_kernel(int d) { return d <= search_maxd() && getenv("MQGAN_VQ_3LAUNCH") == nullptr; }

extern "C" size_t mq_l2_screen_workspace(int64_t n, int k, int d)
{
    if (!screen_shape_ok(n, k, d)) return 0;
    const size_t a = screen_layout(n > 0 ? n : 0, k, d).ws;
    const size_t b = d <= 512 ? search_layout(n > 0 ? n : 0, k, d).ws : 0;
    return a > b ? a : b;
}

extern "C" int mq_l2_screen_prepare(const float *codebook, int k, int d, void *prepared, mqStream stream)
{
    MQ_REQUIRE(codebook && prepared, "mq_l2_screen_prepare: null pointer");
    MQ_REQUIRE(screen_shape_ok(0, k, d), "mq_l2_screen_prepare: need k > 0, d %% 8 == 0, k * d * 4 < 4 GiB (k=%d, d=%d)", k, d);
    MQ_REQUIRE((uintptr_t)codebook % 16 == 0 && (uintptr_t)prepared % 256 == 0, "mq_l2_screen_prepare: alignment (codebook 16, blob 256)");
    const ScreenLayout L = screen_layout(0, k, d);
    char *b = reinterpret_cast<char *>(prepared);
    unsigned *cmax2 = reinterpret_cast<unsigned *>(b + L.cmax2);
    hipError_t e = hipMemsetAsync(cmax2, 0, 16, S(stream));
    MQ_REQUIRE(e == hipSuccess, "mq_l2_screen_prepare: memset failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(l2_screen_norms_kernel, dim3((unsigned)cdiv(L.k_pad, 256)), dim3(256), 0, S(stream), codebook, k, L.k_pad, d,
                       reinterpret_cast<float *>(b + L.cn), reinterpret_cast<float *>(b + L.cnh));
    hipLaunchKernelGGL(l2_screen_cbprep_kernel, dim3((unsigned)cdiv(k, 4)), dim3(256), 0, S(stream), codebook, k, d,
                       reinterpret_cast<unsigned short *>(b + L.ch), cmax2);
    MQ_LAUNCH_CHECK("mq_l2_screen_prepare");
    return 0;
}

extern "C" int mq_l2_argmin_screened(const float *x, const float *codebook, const void *prepared, int64_t n, int k, int d,
                                     int32_t *idx, float *best_dist, void *workspace, size_t workspace_bytes, mqStream stream)
{
    MQ_REQUIRE(x && codebook && prepared && idx && workspace, "mq_l2_argmin_screened: null pointer");
    MQ_REQUIRE(screen_shape_ok(n, k, d), "mq_l2_argmin_screened: shape not supported (ask mq_l2_screen_supported; n=%lld k=%d d=%d)",
               (long long)n, k, d);
    MQ_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)codebook % 16 == 0 && (uintptr_t)prepared % 256 == 0 && (uintptr_t)workspace % 256 == 0,
               "mq_l2_argmin_screened: alignment (x / codebook 16, blob / workspace 256)");
    if (n == 0) return 0;
    MQ_REQUIRE(workspace_bytes >= mq_l2_screen_workspace(n, k, d), "mq_l2_argmin_screened: workspace too small (%zu < %zu)", workspace_bytes,
               mq_l2_screen_workspace(n, k, d));
    const ScreenLayout L = screen_layout(n, k, d);
    if (search_one_kernel(d)) {
        const SearchLayout Q = search_layout(n, k, d);
        const char *pb = reinterpret_cast<const char *>(prepared);
        char *w = reinterpret_cast<char *>(workspace);
        hipStream_t st = S(stream);
        hipError_t e = hipSuccess;
        const dim3 grid((unsigned)cdiv(n, Q.rt), (unsigned)Q.splits);
#define MQ_SEARCH(WC_, WR_, TM_, TN_)                                                                                                        \
    do {                                                                                                                                     \
        static bool attr_done = false;                                                                                                       \
        if (!attr_done) {                                                                                                                    \
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&l2_search_kernel<WC_, WR_, TM_, TN_>),                                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);                                                \
            MQ_REQUIRE(e == hipSuccess, "mq_l2_argmin_screened: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));             \
            attr_done = true;                                                                                                                \
        }                                                                                                                                    \
        hipLaunchKernelGGL((l2_search_kernel<WC_, WR_, TM_, TN_>), grid, dim3(512), Q.dyn_lds, st, x,                                       \
                           reinterpret_cast<const unsigned short *>(pb + L.ch), reinterpret_cast<const float *>(pb + L.cnh),               \
                           reinterpret_cast<const float *>(pb + L.cmax2), n, k, d, Q.tiles_per_split, Q.splits,                            \
                           reinterpret_cast<uint2 *>(w + Q.cand), reinterpret_cast<uint2 *>(w + Q.head), reinterpret_cast<float *>(w + Q.gsplit), \
                           reinterpret_cast<float *>(w + Q.bound), reinterpret_cast<unsigned *>(w + Q.counters));                          \
    } while (0)
        if (Q.rt == 128) MQ_SEARCH(4, 2, 1, 2);
        else MQ_SEARCH(4, 2, 1, 1);
#undef MQ_SEARCH
        hipLaunchKernelGGL(l2_rerank2_kernel, dim3((unsigned)cdiv(n, RR_ROWS)), dim3(256), 0, st, x, codebook, reinterpret_cast<const float *>(pb + L.cn),
                           reinterpret_cast<const uint2 *>(w + Q.cand), reinterpret_cast<const uint2 *>(w + Q.head),
                           reinterpret_cast<const float *>(w + Q.gsplit), reinterpret_cast<const float *>(w + Q.bound), n, k, d, Q.splits, Q.lpr, idx,
                           best_dist, reinterpret_cast<unsigned *>(w + Q.counters));
        MQ_LAUNCH_CHECK("mq_l2_argmin_screened");
        return 0;
    }
    const char *pb = reinterpret_cast<const char *>(prepared);
    const float *cn = reinterpret_cast<const float *>(pb + L.cn);
    const float *cnh = reinterpret_cast<const float *>(pb + L.cnh);
    const float *cmax2 = reinterpret_cast<const float *>(pb + L.cmax2);
    const unsigned short *ch = reinterpret_cast<const unsigned short *>(pb + L.ch);
    char *w = reinterpret_cast<char *>(workspace);
    uint2 *cand = reinterpret_cast<uint2 *>(w + L.cand);
    unsigned char *cntb = reinterpret_cast<unsigned char *>(w + L.cntb);
    unsigned *gmax = reinterpret_cast<unsigned *>(w + L.gmax);
    float *bound = reinterpret_cast<float *>(w + L.bound);
    unsigned *counters = reinterpret_cast<unsigned *>(w + L.counters);
    hipStream_t st = S(stream);
    unsigned short *xh = reinterpret_cast<unsigned short *>(w + L.xh);
    hipLaunchKernelGGL(l2_screen_prep_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, st, x, n, d, xh, cmax2, bound, gmax, counters);
    hipLaunchKernelGGL(l2_screen_kernel, dim3((unsigned)cdiv(n, SC_RT), (unsigned)L.splits), dim3(256), 0, st, xh, ch, cnh, bound, n, k, d,
                       L.tiles_per_split, L.splits, cand, cntb, gmax);
    hipLaunchKernelGGL(l2_rerank_kernel, dim3((unsigned)cdiv(n, RR_ROWS)), dim3(256), 0, st, x, codebook, cn, cand, cntb, gmax, bound, n, k, d,
                       L.splits, idx, best_dist, counters);
    MQ_LAUNCH_CHECK("mq_l2_argmin_screened");
    return 0;
}

// diagnostics of the last call that used `workspace` (synchronises the stream): entries the screening pass appended to the private
// lists, list entries available, rows that were searched exhaustively
extern "C" int mq_l2_screen_stats(const void *workspace, int64_t n, int k, int d, uint64_t *appended, uint64_t *slots,
                                  uint64_t *rows_over, mqStream stream)
{
    MQ_REQUIRE(workspace && appended && slots && rows_over && n > 0 && screen_shape_ok(n, k, d), "mq_l2_screen_stats: bad arguments");
    if (search_one_kernel(d)) {
        // one-kernel path: a list's header holds {largest score pushed out of the lane's window, entries in the window}
        const SearchLayout Q = search_layout(n, k, d);
        const size_t ne = (size_t)n * Q.splits * Q.lpr;
        uint2 *h = reinterpret_cast<uint2 *>(malloc(ne * 8 + 8));
        MQ_REQUIRE(h != nullptr, "mq_l2_screen_stats: out of host memory");
        const char *w = reinterpret_cast<const char *>(workspace);
        unsigned over = 0;
        hipError_t e = hipMemcpyAsync(h, w + Q.head, ne * 8, hipMemcpyDeviceToHost, S(stream));
        if (e == hipSuccess) e = hipMemcpyAsync(&over, w + Q.counters, 4, hipMemcpyDeviceToHost, S(stream));
        if (e == hipSuccess) e = hipStreamSynchronize(S(stream));
        if (e != hipSuccess) {
            free(h);
            return ::mq::fail("mq_l2_screen_stats: copy failed: %s", hipGetErrorString(e));
        }
        uint64_t a = 0;
        for (size_t i = 0; i < ne; ++i) a += h[i].y;
        free(h);
        *appended = a;
        *slots = ne * SR_CAP;
        *rows_over = over;
        return 0;
    }
    const ScreenLayout L = screen_layout(n, k, d);
    const size_t nl = (size_t)n * L.splits * SC_LISTS;
    unsigned char *h = reinterpret_cast<unsigned char *>(malloc(nl + 4));
    MQ_REQUIRE(h != nullptr, "mq_l2_screen_stats: out of host memory");
    const char *w = reinterpret_cast<const char *>(workspace);
    hipError_t e = hipMemcpyAsync(h, w + L.cntb, nl, hipMemcpyDeviceToHost, S(stream));
    if (e == hipSuccess) e = hipMemcpyAsync(h + nl, w + L.counters, 4, hipMemcpyDeviceToHost, S(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(S(stream));
    if (e != hipSuccess) {
        free(h);
        return ::mq::fail("mq_l2_screen_stats: copy failed: %s", hipGetErrorString(e));
    }
    uint64_t a = 0;
    for (size_t i = 0; i < nl; ++i) a += h[i];
    unsigned over = 0;
    memcpy(&over, h + nl, 4);
    free(h);
    *appended = a;
    *slots = (uint64_t)nl * SC_PL;
    *rows_over = over;
    return 0;
}
