// L2 nearest-neighbour search past the f32-MFMA roof: bf16 candidate SCREENING + exact f32 re-rank
// ("bf16 MFMA for candidate scoring + exact fp32 re-rank of near-ties to keep indices exact").
// Results are bit-identical to mq_l2_argmin / oracle/vq_ref.c: same indices, same best distances.
//
// Definition (oracle/vq_ref.c):  dist(n,k) = fmaf(-2, dot(n,k), cn(k)),  dot = sequential f32 fma chain over d,
//                                idx(n) = argmin_k dist(n,k), lowest k on ties, NaN counts as -inf.
//
// 1. prep      x -> bf16 copy xh (RNE) and, per row, a bound B2(n) = 2 E(n) with  |score(n,k) - dist(n,k)| <= E(n)  for
//              EVERY k (proof below); rows for which no finite bound exists (NaN / inf / huge) are marked for step 4.
// 2. screen    score(n,k) = fmaf(-2, <xh_n, ch_k> on the bf16 MFMA, cn(k)) for all (n,k): 16x the f32 matrix rate.  The
//              GEMM is formed TRANSPOSED (A = code tile, B = row tile), so one lane holds 32 codes of ONE row per 64 x 64
//              wave tile and row-wise work is lane-local.  Pass 1 keeps g(n) = min_k score(n,k) (atomicMin); pass 2 forms
//              the same scores again and puts every (n,k) with score <= g(n) + B2(n) into one of the row's SC_SLOTS
//              candidate slots (per-row atomic counter: no contended global counter); what does not fit there (rows with
//              many near-ties) goes to a shared overflow list.
// 3. re-rank   every candidate gets its exact distance (the oracle's sequential fma chain, one lane per candidate) and
//              goes into the same packed (distance, index) atomicMin as the f32 kernel: minimum distance, lowest index
//              on ties.
// 4. fallback  marked rows (and every row, should the overflow list itself overflow) are searched exhaustively with the
//              exact chain, a block per row - slow, rare, exact.
//
// Why the winner is always a candidate.  Let k* be the oracle's answer for row n.  Then
//   score(k*) <= dist(k*) + E <= dist(k) + E <= score(k) + 2E for all k, so score(k*) <= g + 2E = g + B2: appended.
// Every code with dist == dist(k*) satisfies the same inequalities, so the lowest such index is among the re-ranked
// candidates too; all other candidates have a larger exact distance.  (Both passes compute bit-identical scores: same
// tiles, same MFMA order.)
//
// The bound.  bf16 has 8 significant bits: |xh - x| <= 2^-8 |x|, so |xh ch - x c| <= (2^-7 + 2^-16) |x c| per term; bf16
// products are exact in f32; the MFMA's f32 accumulation and the oracle's chain each err by at most D 2^-23 sum|x c|
// (any summation order).  With sum|x_i c_i| <= |x| |c| (Cauchy-Schwarz):
//   |dot_bf16 - dot_chain| <= delta |x_n| max_k|c_k|,   delta = 1.02 * 2^-7 + D * 2^-21,
//   E(n) = 2.02 delta |x_n| cmax + 2^-20 cmax^2 + 1e-30   (the last two: final fmaf roundings, flushed denormals).
// |x_n| and cmax are taken 0.1 % high to cover their own rounding.  Rows with |x_n| or cmax above 1e18 or not finite
// have no bound (products may overflow): step 4.
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
constexpr int SC_SLOTS = 8;           // candidate slots per row; further candidates of a row go to a shared overflow list

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

// ------------------------------------------------------------------------------------------ 1. prep
// one wave per row of src [rows][d]: bf16 copy, |row|^2.  MODE 0: x rows (bound, per-row state); MODE 1: codebook rows
// (max |c|^2 over the codes).
template <int MODE>
__global__ __launch_bounds__(256) void l2_screen_prep_kernel(const float *__restrict__ src, int64_t rows, int d,
                                                             unsigned short *__restrict__ dst_h, const float *__restrict__ cmax2,
                                                             float *__restrict__ bound2, unsigned *__restrict__ amin,
                                                             unsigned long long *__restrict__ packed, uint2 *__restrict__ rowcnt,
                                                             unsigned *__restrict__ counters, unsigned *__restrict__ cmax2_bits)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (MODE == 0 && blockIdx.x == 0 && threadIdx.x < 4) counters[threadIdx.x] = 0u;
    if (row >= rows) return;
    const float *s = src + row * d;
    unsigned short *o = dst_h + row * d;
    float ss = 0.f;
    for (int t = lane * 4; t < d; t += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(s + t);
        ss = fmaf(v.x, v.x, ss);
        ss = fmaf(v.y, v.y, ss);
        ss = fmaf(v.z, v.z, ss);
        ss = fmaf(v.w, v.w, ss);
        *reinterpret_cast<uint2 *>(o + t) = pack_bf16x4(v);
    }
    ss = wave_sum(ss);
    if (lane != 0) return;
    if (MODE == 1) {
        // NaN orders above +inf as an unsigned pattern: a non-finite code poisons the bound of every row (-> fallback)
        atomicMax(cmax2_bits, __builtin_bit_cast(unsigned, ss));
        return;
    }
    const float nx = sqrtf(ss) * 1.001f;
    const float cm = sqrtf(cmax2[0]) * 1.001f;
    const float delta = 1.02f * 0.0078125f + (float)d * 4.76837158203125e-07f;        // 2^-7, 2^-21
    const float e = 2.02f * delta * nx * cm + 9.5367431640625e-07f * cm * cm + 1e-30f;
    const bool ok = nx < 1e18f && cm < 1e18f && e < 1e30f;                              // false for NaN too
    bound2[row] = ok ? 2.f * e : -INFINITY;
    amin[row] = 0xFFFFFFFFu;
    packed[row] = ~0ull;
    rowcnt[row] = make_uint2(0u, 0u);                  // candidates appended, candidates re-ranked
}

// pads the code norms to a multiple of the code tile with +inf (a padded code never scores below anything)
__global__ void l2_screen_cnpad_kernel(float *__restrict__ cn, int k, int k_pad)
{
    const int i = k + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k_pad) cn[i] = INFINITY;
}

// ------------------------------------------------------------------------------------------ 2. screen
// grid (row tiles, code splits), 256 threads = 4 waves: wave (wc, wr) owns codes [64 wc, +64) x rows [64 wr, +64) of the
// block tile as 2 x 2 MFMA tiles of 32 x 32 (v_mfma_f32_32x32x16_bf16, A = codes, B = rows).  Both operand tiles
// (128 x 64 k's, 128-byte rows, 16-byte chunks XOR-swizzled with (row >> 1) & 7) are staged by LDS-DMA into a ring of two
// stages; the x tile is re-streamed from L2 for every code tile (D <= 512: 128 KB per block, it stays in L2).
// Two passes over the same GEMM (it runs at 16x the f32 rate, so twice is still cheap):
//   EMIT = false: the per-row minimum score -> amin (atomicMin on the order-preserving integer image);
//   EMIT = true : every (row, code) with score <= amin[row] + B2[row] is appended to the candidate list - with the FINAL
//                 minimum as reference that is the candidate set itself (1-3 entries per row on random data), where a
//                 single pass could only compare against a running minimum and appended 30-300 per row.
// Appending is aggregated per wave and code tile (one atomicAdd for all of the wave's candidates).
template <bool EMIT>
__global__ __launch_bounds__(256, 2) void l2_screen_kernel(const unsigned short *__restrict__ xh, const unsigned short *__restrict__ ch,
                                                           const float *__restrict__ cn_pad, const float *__restrict__ bound2,
                                                           int64_t n, int k, int d, int tiles_per_split, unsigned *__restrict__ amin,
                                                           int *__restrict__ cand, uint2 *__restrict__ rowcnt,
                                                           unsigned long long *__restrict__ ovf, unsigned *__restrict__ counters,
                                                           unsigned ovf_cap)
{
    __shared__ __attribute__((aligned(1024))) char smem[2 * SC_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wr = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * SC_RT;
    const int ntiles = (k + SC_CT - 1) / SC_CT;
    const int t_begin = blockIdx.y * tiles_per_split;
    const int t_end = min(ntiles, t_begin + tiles_per_split);
    const int nk = (d + 63) >> 6;
    const int total = nk * (t_end - t_begin);
    if (total <= 0) return;

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

    // ---- fragment addresses: lane (li, lh) reads chunk (2 s + lh) ^ ((li >> 1) & 7) of row li of a 32-row MFMA tile
    const int sw = (li >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = li * SC_ROWB + ((((2 * s) ^ (lh ^ sw)) & 7) << 4);
    const int a_base = (wc * 64) * SC_ROWB, b_base = SC_OP + (wr * 64) * SC_ROWB;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float rm[2], thr[2];
    bool rvalid[2];
    int64_t rown[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        rown[tn] = m0 + wr * 64 + tn * 32 + li;
        rvalid[tn] = rown[tn] < n;
        rm[tn] = INFINITY;
        // rows without a bound: B2 = -inf and amin was never written (its image reads back as NaN): nothing compares <= thr
        thr[tn] = (EMIT && rvalid[tn]) ? unord_f32(amin[rown[tn]]) + bound2[rown[tn]] : -INFINITY;
    }

    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
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
            // ---- fold the finished code tile.  Lane (li, lh) holds, for row tn*32 + li, the codes
            //      cbase + tm*32 + 8 (r >> 2) + 4 lh + (r & 3): norms as float4 per (tm, r >> 2)
            const int cbase = (t_begin + tile) * SC_CT + wc * 64;
            float4 cnv[2][4];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    cnv[tm][q] = *reinterpret_cast<const float4 *>(cn_pad + cbase + tm * 32 + 8 * q + 4 * lh);
            auto cn_of = [&](int tm, int r) {
                const float4 v = cnv[tm][r >> 2];
                return (r & 3) == 0 ? v.x : (r & 3) == 1 ? v.y : (r & 3) == 2 ? v.z : v.w;
            };
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                float tmin = INFINITY;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tmin = fminf(tmin, fmaf(-2.0f, acc[tm][tn][r], cn_of(tm, r)));
                if constexpr (!EMIT) {
                    rm[tn] = fminf(rm[tn], tmin);
                } else if (__any(tmin <= thr[tn])) {                 // wave-uniform: at large K most wave tiles hold no candidate
                    // Four codes at a time: one test of their minimum, the per-code tests only where that passes.  A lane
                    // first collects its candidates in registers (up to four; more is rare and goes out one by one), then
                    // reserves their slots with ONE atomic: a returning atomic per candidate serialised ~6 round trips per
                    // wave and tile.
                    int c0 = 0, c1 = 0, c2 = 0, c3 = 0, cnt = 0;
                    auto put = [&](int code) {
                        const unsigned slot = atomicAdd(&rowcnt[rown[tn]].x, 1u);
                        if (slot < (unsigned)SC_SLOTS) {
                            cand[rown[tn] * SC_SLOTS + slot] = code;
                        } else {
                            const unsigned pos = atomicAdd(&counters[0], 1u);
                            if (pos < ovf_cap) ovf[pos] = ((unsigned long long)rown[tn] << 32) | (unsigned)code;
                        }
                    };
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float dv[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) dv[e] = fmaf(-2.0f, acc[tm][tn][4 * q + e], cn_of(tm, 4 * q + e));
                            if (fminf(fminf(dv[0], dv[1]), fminf(dv[2], dv[3])) <= thr[tn]) {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (dv[e] <= thr[tn] && dv[e] < INFINITY) {       // +inf: padded codes
                                        const int code = cbase + tm * 32 + 8 * q + 4 * lh + e;
                                        if (cnt >= 4) put(code);
                                        c0 = cnt == 0 ? code : c0;
                                        c1 = cnt == 1 ? code : c1;
                                        c2 = cnt == 2 ? code : c2;
                                        c3 = cnt == 3 ? code : c3;
                                        ++cnt;
                                    }
                            }
                        }
                    if (cnt > 0) {
                        const int nreg = cnt < 4 ? cnt : 4;
                        const unsigned base = atomicAdd(&rowcnt[rown[tn]].x, (unsigned)nreg);
                        const int cs[4] = {c0, c1, c2, c3};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (j < nreg) {
                                const unsigned slot = base + (unsigned)j;
                                if (slot < (unsigned)SC_SLOTS) {
                                    cand[rown[tn] * SC_SLOTS + slot] = cs[j];
                                } else {
                                    const unsigned pos = atomicAdd(&counters[0], 1u);
                                    if (pos < ovf_cap) ovf[pos] = ((unsigned long long)rown[tn] << 32) | (unsigned)cs[j];
                                }
                            }
                    }
                }
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
            }
            ks = 0;
            ++tile;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if constexpr (!EMIT) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const float m = fminf(rm[tn], __shfl_xor(rm[tn], 32, 64));
            if (lh == 0 && rvalid[tn] && bound2[rown[tn]] > -INFINITY) atomicMin(&amin[rown[tn]], ord_f32(m));
        }
    }
}

// ------------------------------------------------------------------------------------------ 3 + 4. re-rank, fallback
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

// Three block ranges.  [0, slot_blocks): one lane per (row, slot) - the lanes of a row are neighbours, their x loads
// coincide.  [slot_blocks, slot_blocks + ovf_blocks): the overflow list, grid-stride.  The remaining blocks: one block per
// 64 rows, exhaustive exact search of the rows that have no bound or (defensively) no candidate at all - and of every row
// if the overflow list ran over.  Results meet in packed[row] (atomicMin); l2_screen_unpack_kernel splits them.  (Having
// the last candidate of a row publish it instead needs a device-scope fence per candidate: 40 k L2 write-backs cost more
// than the extra launch.)
__device__ __forceinline__ void rerank_one(const float *__restrict__ x, const float *__restrict__ cb, const float *__restrict__ cn_pad,
                                           int64_t row, int code, int d, unsigned long long *__restrict__ packed)
{
    const float dist = exact_dist(x + row * d, cb + (int64_t)code * d, d, cn_pad[code]);
    atomicMin(&packed[row], pack_di(dist, code));
}

__global__ __launch_bounds__(256) void l2_rerank_kernel(const float *__restrict__ x, const float *__restrict__ cb,
                                                        const float *__restrict__ cn_pad, const float *__restrict__ bound2,
                                                        const int *__restrict__ cand, uint2 *__restrict__ rowcnt,
                                                        const unsigned long long *__restrict__ ovf, const unsigned *__restrict__ counters,
                                                        unsigned ovf_cap, int64_t n, int k, int d, int slot_blocks, int ovf_blocks,
                                                        unsigned long long *__restrict__ packed)
{
    const unsigned n_ovf = counters[0];
    const bool all_exhaustive = n_ovf > ovf_cap;
    if ((int)blockIdx.x < slot_blocks) {
        if (all_exhaustive) return;
        const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
        const int64_t row = i / SC_SLOTS;
        const int slot = (int)(i - row * SC_SLOTS);
        if (row >= n) return;
        if (slot >= (int)rowcnt[row].x) return;
        rerank_one(x, cb, cn_pad, row, cand[row * SC_SLOTS + slot], d, packed);
        return;
    }
    if ((int)blockIdx.x < slot_blocks + ovf_blocks) {
        if (all_exhaustive) return;
        for (unsigned e = (blockIdx.x - slot_blocks) * 256u + threadIdx.x; e < n_ovf; e += (unsigned)ovf_blocks * 256u) {
            const unsigned long long key = ovf[e];
            rerank_one(x, cb, cn_pad, (int64_t)(key >> 32), (int)(key & 0xffffffffu), d, packed);
        }
        return;
    }
    __shared__ unsigned long long s_todo;
    __shared__ unsigned long long s_best[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)(blockIdx.x - slot_blocks - ovf_blocks) * 64;
    if (wave == 0) {
        const int64_t myrow = r0 + lane;
        const bool need = myrow < n && (all_exhaustive || !(bound2[myrow] > -INFINITY) || rowcnt[myrow].x == 0u);
        const unsigned long long m = __ballot(need);
        if (lane == 0) s_todo = m;
    }
    __syncthreads();
    unsigned long long todo = s_todo;
    while (todo) {                                                     // block-uniform
        const int b = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int64_t row = r0 + b;
        unsigned long long best = ~0ull;
        for (int code = threadIdx.x; code < k; code += 256) {
            const unsigned long long p = pack_di(exact_dist(x + row * d, cb + (int64_t)code * d, d, cn_pad[code]), code);
            best = p < best ? p : best;
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_xor(best, off, 64);
            best = o < best ? o : best;
        }
        if (lane == 0) s_best[wave] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) best = s_best[w] < best ? s_best[w] : best;
            packed[row] = best;                                       // no candidate of this row is re-ranked elsewhere
        }
        __syncthreads();
    }
}

__global__ void l2_screen_unpack_kernel(const unsigned long long *__restrict__ packed, int64_t n, int32_t *__restrict__ idx,
                                        float *__restrict__ dist)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long p = packed[i];
        idx[i] = (int32_t)(p & 0xffffffffu);
        if (dist) dist[i] = unord_f32((unsigned)(p >> 32));
    }
}

// sequential-chain code norms (the oracle's cn), written into the padded array
__global__ void l2_screen_norms_kernel(const float *__restrict__ c, int k, int d, float *__restrict__ cn)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    float acc = 0.f;
    for (int t = 0; t < d; ++t) acc = __fmaf_rn(c[(size_t)j * d + t], c[(size_t)j * d + t], acc);
    cn[j] = acc;
}

struct ScreenLayout {              // byte offsets inside the prepared-codebook blob and the workspace
    size_t cn, cmax2, ch, blob;
    size_t packed, amin, bound2, rowcnt, counters, xh, cand, ovf, ws;
    unsigned ovf_cap;
    int k_pad;
};
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static ScreenLayout screen_layout(int64_t n, int k, int d)
{
    ScreenLayout L;
    L.k_pad = (int)(cdiv(k, SC_CT) * SC_CT);
    L.cn = 0;
    L.cmax2 = al256((size_t)L.k_pad * 4);
    L.ch = L.cmax2 + 256;
    L.blob = L.ch + al256((size_t)k * d * 2);
    size_t o = 0;
    L.packed = o; o += al256((size_t)n * 8);
    L.amin = o; o += al256((size_t)n * 4);
    L.bound2 = o; o += al256((size_t)n * 4);
    L.rowcnt = o; o += al256((size_t)n * 8);
    L.xh = o; o += al256((size_t)n * d * 2);
    L.cand = o; o += al256((size_t)n * SC_SLOTS * 4);
    L.counters = o; o += 256;
    int64_t cap = 4 * n + 4096;
    if (cap > 0x7FFFFFF0ll) cap = 0x7FFFFFF0ll;
    L.ovf_cap = (unsigned)cap;
    L.ovf = o; o += al256((size_t)cap * 8);
    L.ws = o;
    return L;
}

}  // namespace mq

using namespace mq;

static bool screen_shape_ok(int64_t n, int k, int d)
{
    return n >= 0 && n < ((int64_t)1 << 31) && k > 0 && d > 0 && d % 8 == 0 && (int64_t)SC_RT * d * 2 < 0x7FFFFFF0ll &&
           (int64_t)k * d * 2 < 0xFFFFFFF0ll;
}

extern "C" int mq_l2_screen_supported(int64_t n, int k, int d) { return screen_shape_ok(n, k, d) ? 1 : 0; }

extern "C" size_t mq_l2_screen_codebook_bytes(int k, int d) { return screen_shape_ok(0, k, d) ? screen_layout(0, k, d).blob : 0; }

extern "C" size_t mq_l2_screen_workspace(int64_t n, int k, int d)
{
    return screen_shape_ok(n, k, d) ? screen_layout(n > 0 ? n : 0, k, d).ws : 0;
}

extern "C" int mq_l2_screen_prepare(const float *codebook, int k, int d, void *prepared, mqStream stream)
{
    MQ_REQUIRE(codebook && prepared, "mq_l2_screen_prepare: null pointer");
    MQ_REQUIRE(screen_shape_ok(0, k, d), "mq_l2_screen_prepare: need k > 0, d %% 8 == 0, k * d * 2 < 4 GiB (k=%d, d=%d)", k, d);
    MQ_REQUIRE((uintptr_t)codebook % 16 == 0 && (uintptr_t)prepared % 256 == 0, "mq_l2_screen_prepare: alignment (codebook 16, blob 256)");
    const ScreenLayout L = screen_layout(0, k, d);
    char *b = reinterpret_cast<char *>(prepared);
    float *cn = reinterpret_cast<float *>(b + L.cn);
    unsigned *cmax2 = reinterpret_cast<unsigned *>(b + L.cmax2);
    hipError_t e = hipMemsetAsync(cmax2, 0, 16, S(stream));
    MQ_REQUIRE(e == hipSuccess, "mq_l2_screen_prepare: memset failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(l2_screen_norms_kernel, dim3((unsigned)cdiv(k, 256)), dim3(256), 0, S(stream), codebook, k, d, cn);
    if (L.k_pad > k)
        hipLaunchKernelGGL(l2_screen_cnpad_kernel, dim3((unsigned)cdiv(L.k_pad - k, 128)), dim3(128), 0, S(stream), cn, k, L.k_pad);
    hipLaunchKernelGGL(l2_screen_prep_kernel<1>, dim3((unsigned)cdiv(k, 4)), dim3(256), 0, S(stream), codebook, (int64_t)k, d,
                       reinterpret_cast<unsigned short *>(b + L.ch), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, cmax2);
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
    const ScreenLayout L = screen_layout(n, k, d);
    MQ_REQUIRE(workspace_bytes >= L.ws, "mq_l2_argmin_screened: workspace too small (%zu < %zu)", workspace_bytes, L.ws);
    const char *pb = reinterpret_cast<const char *>(prepared);
    const float *cn = reinterpret_cast<const float *>(pb + L.cn);
    const float *cmax2 = reinterpret_cast<const float *>(pb + L.cmax2);
    const unsigned short *ch = reinterpret_cast<const unsigned short *>(pb + L.ch);
    char *w = reinterpret_cast<char *>(workspace);
    unsigned long long *packed = reinterpret_cast<unsigned long long *>(w + L.packed);
    unsigned *amin = reinterpret_cast<unsigned *>(w + L.amin);
    float *bound2 = reinterpret_cast<float *>(w + L.bound2);
    uint2 *rowcnt = reinterpret_cast<uint2 *>(w + L.rowcnt);
    unsigned short *xh = reinterpret_cast<unsigned short *>(w + L.xh);
    int *cand = reinterpret_cast<int *>(w + L.cand);
    unsigned *counters = reinterpret_cast<unsigned *>(w + L.counters);
    unsigned long long *ovf = reinterpret_cast<unsigned long long *>(w + L.ovf);
    hipStream_t st = S(stream);
    hipLaunchKernelGGL(l2_screen_prep_kernel<0>, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, st, x, n, d, xh, cmax2, bound2, amin, packed,
                       rowcnt, counters, nullptr);
    const int row_tiles = (int)cdiv(n, SC_RT);
    const int ntiles = (int)cdiv(k, SC_CT);
    int splits = (int)cdiv(1024, row_tiles);            // about two rounds of 2 blocks per CU
    if (splits > ntiles) splits = ntiles;
    if (splits < 1) splits = 1;
    const int tps = (int)cdiv(ntiles, splits);
    splits = (int)cdiv(ntiles, tps);
    hipLaunchKernelGGL(l2_screen_kernel<false>, dim3(row_tiles, splits), dim3(256), 0, st, xh, ch, cn, bound2, n, k, d, tps, amin, cand,
                       rowcnt, ovf, counters, L.ovf_cap);
    hipLaunchKernelGGL(l2_screen_kernel<true>, dim3(row_tiles, splits), dim3(256), 0, st, xh, ch, cn, bound2, n, k, d, tps, amin, cand,
                       rowcnt, ovf, counters, L.ovf_cap);
    const int slot_blocks = (int)cdiv(n * SC_SLOTS, 256), ovf_blocks = 64;
    const int fb_blocks = (int)cdiv(n, 64);
    hipLaunchKernelGGL(l2_rerank_kernel, dim3(slot_blocks + ovf_blocks + fb_blocks), dim3(256), 0, st, x, codebook, cn, bound2, cand, rowcnt,
                       ovf, counters, L.ovf_cap, n, k, d, slot_blocks, ovf_blocks, packed);
    hipLaunchKernelGGL(l2_screen_unpack_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, st, packed, n, idx, best_dist);
    MQ_LAUNCH_CHECK("mq_l2_argmin_screened");
    return 0;
}

// candidates appended / slots available in the last call that used `workspace` (diagnostics; synchronises the stream)
extern "C" int mq_l2_screen_stats(const void *workspace, int64_t n, int k, int d, uint64_t *appended, uint64_t *slots,
                                  uint64_t *rows_over, mqStream stream)
{
    MQ_REQUIRE(workspace && appended && slots && rows_over && n > 0 && screen_shape_ok(n, k, d), "mq_l2_screen_stats: bad arguments");
    const ScreenLayout L = screen_layout(n, k, d);
    uint2 *h = reinterpret_cast<uint2 *>(malloc((size_t)n * 8));
    MQ_REQUIRE(h != nullptr, "mq_l2_screen_stats: out of host memory");
    hipError_t e = hipMemcpyAsync(h, reinterpret_cast<const char *>(workspace) + L.rowcnt, (size_t)n * 8, hipMemcpyDeviceToHost, S(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(S(stream));
    if (e != hipSuccess) {
        free(h);
        return ::mq::fail("mq_l2_screen_stats: copy failed: %s", hipGetErrorString(e));
    }
    uint64_t a = 0, o = 0;
    for (int64_t i = 0; i < n; ++i) {
        a += h[i].x;
        o += h[i].x > (unsigned)SC_SLOTS ? h[i].x - SC_SLOTS : 0;
    }
    free(h);
    *appended = a;
    *slots = (uint64_t)n * SC_SLOTS;
    *rows_over = o > L.ovf_cap ? (uint64_t)n : 0;      // overflow list ran over: every row was searched exhaustively
    return 0;
}
