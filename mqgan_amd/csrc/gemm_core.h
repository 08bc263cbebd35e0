// f32-MFMA "NT" GEMM core shared by the implicit-GEMM convolutions and the L2 nearest-neighbour
// search: C[m][n] = sum_k A[m][k] * B[n][k], both operands K-contiguous per row.
//
// Block = 256 threads = 4 waves; block tile BM(=128) x BN x BK(=16); each wave owns TM x TN tiles of
// 32x32 computed with v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma chain).  Operands are staged
// global -> registers -> LDS (double buffered, one barrier per K-step); the global loads of step t+1
// are issued before the MFMAs of step t.
//
// LDS image: rows of BK floats padded to LD = 18 (2 x odd: the 32 lanes a ds_read_b64 serves per pass then
// cover all 64 banks - the FRAGMENT READS are conflict-free: 18 li mod 64 takes 32 distinct even values).
// The staging WRITES are not: a wave writes rows r .. r+15 at columns 4 kq (+2) as float2, and
// (18 r + 4 kq) mod 64 collides between rows (row 0 / kq 2 with row 4 / kq 0, ...): that is where the
// measured 25 % SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of these kernels comes from.  It costs nothing
// measurable: writes are a fifth of the LDS instructions, the LDS is busy about half of the time and the
// matrix pipe 98.5 % of the wave cycles in the main loop (profiles/r2_mfma_counters.json).
// Inside each aligned group of 4 k's the middle two are swapped when written,
// (k0,k2,k1,k3), so that lane half h reads ONE float2 at offset 4q+2h = (k = 4q+h, k = 4q+2+h): the
// operands of MFMA steps 2q and 2q+1.  Every accumulator therefore sums k in natural order, which
// the nearest-neighbour search relies on for bit-exactness against oracle/vq_ref.c.
#pragma once
#include <type_traits>

#include "common.h"

namespace mq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_BM = 128;
constexpr int GEMM_BK = 16;
constexpr int GEMM_LD = 18;
constexpr int GEMM_LDH = 40;       // bf16 engine: 32 k's per row + 8 pad (80-byte rows: conflict-free ds_read_b128)

// Engine policies.  A tile row of BK k's is loaded as QPR float4 quads, so the 256 threads cover ROWS rows per
// pass and a 128-row operand tile takes NS passes ("slots").
struct EngF32 {        // v_mfma_f32_32x32x2_f32: exact f32 products, k-ordered accumulation
    static constexpr int BK = 16, QPR = 4, ROWS = 64, NS = 2;
    static constexpr int ESZ = 4, EPL = 4;              // operand element size in HBM, elements per 16-byte load
    static constexpr bool BF16 = false, SPLIT = false, RAW = false;
};
struct EngBF16 {       // v_mfma_f32_32x32x16_bf16: operands rounded to bf16 (RNE) on the way into LDS, f32 accumulate
    static constexpr int BK = 32, QPR = 8, ROWS = 32, NS = 4;
    static constexpr int ESZ = 4, EPL = 4;
    static constexpr bool BF16 = true, SPLIT = false, RAW = false;
};
struct EngBF16H {      // same MFMA and LDS image, but the operands are ALREADY bf16 in HBM (mq_cast_bf16 copies):
    static constexpr int BK = 32, QPR = 4, ROWS = 64, NS = 2;      // half the bytes, half the loads, no conversion
    static constexpr int ESZ = 2, EPL = 8;
    static constexpr bool BF16 = true, SPLIT = false, RAW = true;
};
// f32 on the bf16 matrix pipe: every operand x is decomposed exactly into three bf16 terms x = h + m + l (8 + 8 + 8
// significant bits) on its way into LDS and the product is formed from 6 of the 9 cross terms (everything down to
// 2^-24 relative: hh, hm, mh, hl, lh, mm), each exact in the f32 accumulator - the accuracy of an f32 FMA chain
// at 6 x 32 cycles per 32x32x16 block instead of 8 x 64.
struct EngSplit {
    static constexpr int BK = 16, QPR = 4, ROWS = 64, NS = 2;
    static constexpr int ESZ = 4, EPL = 4;
    static constexpr bool BF16 = false, SPLIT = true, RAW = false;
};
// split engine: 16 bf16 k's per row, no padding (three images must fit three blocks per CU); the two 16-byte halves
// of a row are swapped in rows with bit 3 set, so the 16 rows a ds_read_b128 lane group touches (distinct mod 16)
// land on 16 distinct bank quads
constexpr int GEMM_LDS3 = 16;

template <int BN, class E = EngF32>
struct GemmCfg {
    static constexpr int WM = (BN == 32) ? 4 : 2;       // waves along M
    static constexpr int WN = 4 / WM;                   // waves along N
    static constexpr int TM = GEMM_BM / WM / 32;        // 32x32 tiles per wave along M
    static constexpr int TN = BN / WN / 32;
    static constexpr int A_SLOTS = GEMM_BM / E::ROWS;   // float4 loads per thread for the A tile
    static constexpr int B_SLOTS = (BN >= E::ROWS) ? BN / E::ROWS : 1;
    static constexpr int LDS_FLOATS = E::SPLIT ? 3 * (GEMM_BM + BN) * GEMM_LDS3                       // in 4-byte units
                                      : E::BF16 ? (GEMM_BM + BN) * GEMM_LDH : 2 * (GEMM_BM + BN) * GEMM_LD;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0xFFFFFFFFu;

// 16-byte loads through a buffer descriptor: an out-of-range offset returns zeros without touching memory
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, size_t bytes)
{
    const unsigned n = bytes > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), (short)0, (int)n, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
// the same with a wave-uniform byte offset in the instruction's SCALAR operand (added to the address after the bounds check of the
// vector offset: callers keep vector + scalar offset inside the row the vector offset points into)
__device__ __forceinline__ float4 buf_load4s(__amdgpu_buffer_rsrc_t r, unsigned byte_off, int scalar_off)
{
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, scalar_off, 0));
}

// KORDER: (k0,k2,k1,k3) image = natural accumulation order (the nearest-neighbour search needs it, see the header
// comment).  Otherwise the quad is stored as loaded - no register shuffles - and every accumulator sums k in the
// order 4q, 4q+2, 4q+1, 4q+3 (fine for the convolutions, whose parity bound is 1e-4, not bit-exactness).
template <bool KORDER>
__device__ __forceinline__ void lds_store_k4(float *dst, float4 v)
{
    if (KORDER) {
        *reinterpret_cast<float2 *>(dst) = make_float2(v.x, v.z);
        *reinterpret_cast<float2 *>(dst + 2) = make_float2(v.y, v.w);
    } else {
        *reinterpret_cast<float2 *>(dst) = make_float2(v.x, v.y);
        *reinterpret_cast<float2 *>(dst + 2) = make_float2(v.z, v.w);
    }
}

// ALoad / BLoad: .stage(k0) once per K-step, in increasing k0 (0, BK, 2BK, ...): computes and keeps the
// addresses of the thread's loads for that step; then .fetch(slot) -> float4 of k = k0 + 4*(tid&3) .. +3
// for tile row (tid>>2) + 64*slot.  Out-of-range elements come back as 0.
//
// Per K-step the program order is pinned with sched_barriers into three phases:
//   1. issue the global loads of tile kt+1 (addresses were computed during the previous step),
//   2. the 32 MFMAs of tile kt with the LDS fragment reads AND the address arithmetic of tile kt+2
//      interleaved between them (VALU work rides in the shadow of the matrix pipe),
//   3. wait for the loads, write them to the other LDS buffer, barrier.
// Global-load latency is covered by phase 2; there are no branches in the loop (past the end of K the
// loaders' buffer descriptors return zeros without touching memory).
// ZERO_FIRST: the accumulators come in undefined and the very first MFMA of every tile takes C = 0 (an inline constant of the
// instruction) instead of 64 register moves per wave in front of the loop - worth it when the reduction is only a few K-steps deep.
template <int BN, bool KORDER = true, bool ZERO_FIRST = false, class ALoad, class BLoad>
__device__ __forceinline__ void nt_gemm_mainloop(ALoad &la, BLoad &lb, int nk, float *smem,
                                                 f32x16 (&acc)[GemmCfg<BN>::TM][GemmCfg<BN>::TN])
{
    using Cfg = GemmCfg<BN>;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int li = lane & 31, lh = lane >> 5;
    float *As = smem;
    float *Bs = smem + 2 * GEMM_BM * GEMM_LD;
    const int ld_row = tid >> 2, ld_kq = tid & 3;
    const bool b_active = (BN >= 64) || (tid < BN * 4);

    float4 ra[Cfg::A_SLOTS], rb[Cfg::B_SLOTS];
    la.stage(0);
    lb.stage(0);
#pragma unroll
    for (int j = 0; j < Cfg::A_SLOTS; ++j) ra[j] = la.fetch(j);
#pragma unroll
    for (int j = 0; j < Cfg::B_SLOTS; ++j) rb[j] = b_active ? lb.fetch(j) : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < Cfg::A_SLOTS; ++j) lds_store_k4<KORDER>(&As[(ld_row + 64 * j) * GEMM_LD + 4 * ld_kq], ra[j]);
    if (b_active) {
#pragma unroll
        for (int j = 0; j < Cfg::B_SLOTS; ++j) lds_store_k4<KORDER>(&Bs[(ld_row + 64 * j) * GEMM_LD + 4 * ld_kq], rb[j]);
    }
    la.stage(GEMM_BK);
    lb.stage(GEMM_BK);
    __syncthreads();

    // One K-step; `cur` (the LDS buffer being read) is a compile-time constant, the loop below is unrolled by two:
    // all LDS addresses are loop-invariant registers + immediates instead of per-step vector adds.
    auto k_step = [&](int kt, auto cur_tag, auto first_tag) {
        constexpr int cur = decltype(cur_tag)::value;
        constexpr bool first = decltype(first_tag)::value;
        // phase 1
#pragma unroll
        for (int j = 0; j < Cfg::A_SLOTS; ++j) ra[j] = la.fetch(j);
#pragma unroll
        for (int j = 0; j < Cfg::B_SLOTS; ++j) rb[j] = b_active ? lb.fetch(j) : make_float4(0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // phase 2
        const float *Ac = As + cur * GEMM_BM * GEMM_LD + (wm * Cfg::TM * 32 + li) * GEMM_LD + 2 * lh;
        const float *Bc = Bs + cur * BN * GEMM_LD + (wn * Cfg::TN * 32 + li) * GEMM_LD + 2 * lh;
        auto mfma_quads = [&](int q0, int q1) {
#pragma unroll
            for (int q = q0; q < q1; ++q) {
                float2 a[Cfg::TM], b[Cfg::TN];
#pragma unroll
                for (int t = 0; t < Cfg::TM; ++t) a[t] = *reinterpret_cast<const float2 *>(Ac + t * 32 * GEMM_LD + 4 * q);
#pragma unroll
                for (int t = 0; t < Cfg::TN; ++t) b[t] = *reinterpret_cast<const float2 *>(Bc + t * 32 * GEMM_LD + 4 * q);
#pragma unroll
                for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < Cfg::TN; ++tn) {
                        if (first && q == 0) {
                            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].x, b[tn].x, zero, 0, 0, 0);
                        } else {
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].x, b[tn].x, acc[tm][tn], 0, 0, 0);
                        }
                    }
#pragma unroll
                for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < Cfg::TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].y, b[tn].y, acc[tm][tn], 0, 0, 0);
            }
        };
        auto store_next = [&]() {
            float *An = As + (cur ^ 1) * GEMM_BM * GEMM_LD;
            float *Bn = Bs + (cur ^ 1) * BN * GEMM_LD;
#pragma unroll
            for (int j = 0; j < Cfg::A_SLOTS; ++j) lds_store_k4<KORDER>(&An[(ld_row + 64 * j) * GEMM_LD + 4 * ld_kq], ra[j]);
            if (b_active) {
#pragma unroll
                for (int j = 0; j < Cfg::B_SLOTS; ++j)
                    lds_store_k4<KORDER>(&Bn[(ld_row + 64 * j) * GEMM_LD + 4 * ld_kq], rb[j]);
            }
        };
        mfma_quads(0, 4);
        la.stage((kt + 2) * GEMM_BK);
        lb.stage((kt + 2) * GEMM_BK);
        __builtin_amdgcn_sched_barrier(0);
        // phase 3  (writing the tile in the middle of the MFMA phase instead was measured: no difference)
        store_next();
        __syncthreads();
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    if constexpr (ZERO_FIRST) {
        k_step(0, c0{}, std::true_type{});            // nk >= 1
        int kt = 1;
        for (; kt + 1 < nk; kt += 2) {
            k_step(kt, c1{}, std::false_type{});
            k_step(kt + 1, c0{}, std::false_type{});
        }
        if (kt < nk) k_step(kt, c1{}, std::false_type{});
    } else {
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            k_step(kt, c0{}, std::false_type{});
            k_step(kt + 1, c1{}, std::false_type{});
        }
        if (kt < nk) k_step(kt, c0{}, std::false_type{});
    }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint2 pack_bf16x4(float4 v)
{
    const f32x4v f = {v.x, v.y, v.z, v.w};
    return __builtin_bit_cast(uint2, __builtin_convertvector(f, bf16x4));     // v_cvt_pk_bf16_f32 x2 (RNE, NaN-safe)
}

// bf16 engine: same loader protocol with EngBF16's mapping (.fetch(slot) -> k = k0 + 4*(tid&7) .. +3 of tile row
// (tid>>3) + 32*slot), BK = 32.  LDS rows hold 32 bf16 k's in natural order (+8 pad); lane (r, h) reads the 8 k's
// 16s + 8h .. +7 of row r with one ds_read_b128 per 32x32x16 MFMA operand.
template <int BN, class E, class ALoad, class BLoad>
__device__ __forceinline__ void nt_gemm_mainloop_bf16(ALoad &la, BLoad &lb, int nk, float *smem_f,
                                                      f32x16 (&acc)[GemmCfg<BN, E>::TM][GemmCfg<BN, E>::TN])
{
    using Cfg = GemmCfg<BN, E>;
    constexpr int LD = GEMM_LDH;
    unsigned short *smem = reinterpret_cast<unsigned short *>(smem_f);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int li = lane & 31, lh = lane >> 5;
    unsigned short *As = smem;
    unsigned short *Bs = smem + 2 * GEMM_BM * LD;
    const int ld_row = tid / E::QPR, ld_kq = tid % E::QPR;
    const bool b_active = (BN >= E::ROWS) || (tid < BN * E::QPR);
    // RAW: the 16 bytes loaded are 8 bf16 - stored as they are; otherwise 4 floats are rounded and packed into 8 bytes
    auto put = [&](unsigned short *row, float4 v) {
        if constexpr (E::RAW) *reinterpret_cast<float4 *>(row + 8 * ld_kq) = v;
        else *reinterpret_cast<uint2 *>(row + 4 * ld_kq) = pack_bf16x4(v);
    };

    float4 ra[Cfg::A_SLOTS], rb[Cfg::B_SLOTS];
    la.stage(0);
    lb.stage(0);
#pragma unroll
    for (int j = 0; j < Cfg::A_SLOTS; ++j) ra[j] = la.fetch(j);
#pragma unroll
    for (int j = 0; j < Cfg::B_SLOTS; ++j) rb[j] = b_active ? lb.fetch(j) : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < Cfg::A_SLOTS; ++j) put(&As[(ld_row + E::ROWS * j) * LD], ra[j]);
    if (b_active) {
#pragma unroll
        for (int j = 0; j < Cfg::B_SLOTS; ++j) put(&Bs[(ld_row + E::ROWS * j) * LD], rb[j]);
    }
    la.stage(E::BK);
    lb.stage(E::BK);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
#pragma unroll
        for (int j = 0; j < Cfg::A_SLOTS; ++j) ra[j] = la.fetch(j);
#pragma unroll
        for (int j = 0; j < Cfg::B_SLOTS; ++j) rb[j] = b_active ? lb.fetch(j) : make_float4(0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned short *Ac = As + cur * GEMM_BM * LD + (wm * Cfg::TM * 32 + li) * LD + 8 * lh;
        const unsigned short *Bc = Bs + cur * BN * LD + (wn * Cfg::TN * 32 + li) * LD + 8 * lh;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[Cfg::TM], b[Cfg::TN];
#pragma unroll
            for (int t = 0; t < Cfg::TM; ++t) a[t] = *reinterpret_cast<const bf16x8 *>(Ac + t * 32 * LD + 16 * s);
#pragma unroll
            for (int t = 0; t < Cfg::TN; ++t) b[t] = *reinterpret_cast<const bf16x8 *>(Bc + t * 32 * LD + 16 * s);
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < Cfg::TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
        la.stage((kt + 2) * E::BK);
        lb.stage((kt + 2) * E::BK);
        __builtin_amdgcn_sched_barrier(0);
        {
            unsigned short *An = As + (cur ^ 1) * GEMM_BM * LD;
            unsigned short *Bn = Bs + (cur ^ 1) * BN * LD;
#pragma unroll
            for (int j = 0; j < Cfg::A_SLOTS; ++j) put(&An[(ld_row + E::ROWS * j) * LD], ra[j]);
            if (b_active) {
#pragma unroll
                for (int j = 0; j < Cfg::B_SLOTS; ++j) put(&Bn[(ld_row + E::ROWS * j) * LD], rb[j]);
            }
        }
        __syncthreads();
    }
}

// x = h + m + l exactly, each term a bf16 (truncation splits: the remainders x - h and (x - h) - m are exact in f32)
__device__ __forceinline__ void split3_bf16x4(float4 v, uint2 &h, uint2 &m, uint2 &l)
{
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned hb[4], mb[4], lb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned xi = __builtin_bit_cast(unsigned, x[i]);
        hb[i] = xi;                                                    // its upper half is h
        const float r1 = x[i] - __builtin_bit_cast(float, xi & 0xFFFF0000u);
        const unsigned ri = __builtin_bit_cast(unsigned, r1);
        mb[i] = ri;
        lb[i] = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, ri & 0xFFFF0000u));
    }
    // v_perm_b32: bytes 3,2 of the second element over bytes 3,2 of the first = two bf16 in k order
    h = make_uint2(__builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u), __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u));
    m = make_uint2(__builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u), __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u));
    l = make_uint2(__builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u), __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u));
}

// Split engine main loop: EngF32's loader mapping (BK = 16), three bf16 LDS images per operand, double buffered.
template <int BN, class ALoad, class BLoad>
__device__ __forceinline__ void nt_gemm_mainloop_split(ALoad &la, BLoad &lb, int nk, float *smem_f,
                                                       f32x16 (&acc)[GemmCfg<BN, EngSplit>::TM][GemmCfg<BN, EngSplit>::TN])
{
    using Cfg = GemmCfg<BN, EngSplit>;
    constexpr int LD = GEMM_LDS3;
    constexpr int A_IMG = 2 * GEMM_BM * LD, B_IMG = 2 * BN * LD;        // one image, both buffers (in bf16 elements)
    unsigned short *As = reinterpret_cast<unsigned short *>(smem_f);     // [3][2][128][LD]
    unsigned short *Bs = As + 3 * A_IMG;                                 // [3][2][BN][LD]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int li = lane & 31, lh = lane >> 5;
    const int ld_row = tid >> 2, ld_kq = tid & 3;
    const bool b_active = (BN >= 64) || (tid < BN * 4);
    // element offset of this thread's k-quad inside its row (rows ld_row + 64 j share bit 3 with ld_row)
    const int st_k = 8 * ((ld_kq >> 1) ^ ((ld_row >> 3) & 1)) + 4 * (ld_kq & 1);
    const int rd_k = 8 * (lh ^ ((li >> 3) & 1));

    float4 ra[Cfg::A_SLOTS], rb[Cfg::B_SLOTS];
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < Cfg::A_SLOTS; ++j) {
            uint2 h, m, l;
            split3_bf16x4(ra[j], h, m, l);
            unsigned short *d = As + (buf * GEMM_BM + ld_row + 64 * j) * LD + st_k;
            *reinterpret_cast<uint2 *>(d) = h;
            *reinterpret_cast<uint2 *>(d + A_IMG) = m;
            *reinterpret_cast<uint2 *>(d + 2 * A_IMG) = l;
        }
        if (b_active) {
#pragma unroll
            for (int j = 0; j < Cfg::B_SLOTS; ++j) {
                uint2 h, m, l;
                split3_bf16x4(rb[j], h, m, l);
                unsigned short *d = Bs + (buf * BN + ld_row + 64 * j) * LD + st_k;
                *reinterpret_cast<uint2 *>(d) = h;
                *reinterpret_cast<uint2 *>(d + B_IMG) = m;
                *reinterpret_cast<uint2 *>(d + 2 * B_IMG) = l;
            }
        }
    };
    la.stage(0);
    lb.stage(0);
#pragma unroll
    for (int j = 0; j < Cfg::A_SLOTS; ++j) ra[j] = la.fetch(j);
#pragma unroll
    for (int j = 0; j < Cfg::B_SLOTS; ++j) rb[j] = b_active ? lb.fetch(j) : make_float4(0, 0, 0, 0);
    store_tile(0);
    la.stage(EngSplit::BK);
    lb.stage(EngSplit::BK);
    __syncthreads();

    auto k_step = [&](int kt, auto cur_tag) {                  // LDS buffer index as a constant, loop unrolled by two
        constexpr int cur = decltype(cur_tag)::value;
#pragma unroll
        for (int j = 0; j < Cfg::A_SLOTS; ++j) ra[j] = la.fetch(j);
#pragma unroll
        for (int j = 0; j < Cfg::B_SLOTS; ++j) rb[j] = b_active ? lb.fetch(j) : make_float4(0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned short *Ac = As + (cur * GEMM_BM + wm * Cfg::TM * 32 + li) * LD + rd_k;
        const unsigned short *Bc = Bs + (cur * BN + wn * Cfg::TN * 32 + li) * LD + rd_k;
        bf16x8 a[3][Cfg::TM], b[3][Cfg::TN];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int t = 0; t < Cfg::TM; ++t) a[g][t] = *reinterpret_cast<const bf16x8 *>(Ac + g * A_IMG + t * 32 * LD);
#pragma unroll
            for (int t = 0; t < Cfg::TN; ++t) b[g][t] = *reinterpret_cast<const bf16x8 *>(Bc + g * B_IMG + t * 32 * LD);
        }
        // h*h first: its fragments are the first to arrive from LDS, the MFMAs start while the m and l images are
        // still being read (measured +5 % over smallest-terms-first; the accumulator is f32 either way)
        constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < Cfg::TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[p]][tm], b[PB[p]][tn], acc[tm][tn], 0, 0, 0);
        la.stage((kt + 2) * EngSplit::BK);
        lb.stage((kt + 2) * EngSplit::BK);
        store_tile(cur ^ 1);      // no scheduling fence here: the splitting VALU work may overlap the MFMA tail
        __syncthreads();
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        k_step(kt, std::integral_constant<int, 0>{});
        k_step(kt + 1, std::integral_constant<int, 1>{});
    }
    if (kt < nk) k_step(kt, std::integral_constant<int, 0>{});
}

template <class E, int BN, bool ZERO_FIRST = false, class ALoad, class BLoad>
__device__ __forceinline__ void nt_gemm_run(ALoad &la, BLoad &lb, int nk, float *smem,
                                            f32x16 (&acc)[GemmCfg<BN, E>::TM][GemmCfg<BN, E>::TN])
{
    if constexpr (E::SPLIT) nt_gemm_mainloop_split<BN>(la, lb, nk, smem, acc);
    else if constexpr (E::BF16) nt_gemm_mainloop_bf16<BN, E>(la, lb, nk, smem, acc);
    else nt_gemm_mainloop<BN, false, ZERO_FIRST>(la, lb, nk, smem, acc);
}

// row / column of accumulator register r of tile (tm, tn) inside the block tile
template <int BN>
__device__ __forceinline__ int acc_row(int tm, int r)
{
    using Cfg = GemmCfg<BN>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave / Cfg::WN) * Cfg::TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
template <int BN>
__device__ __forceinline__ int acc_col(int tn)
{
    using Cfg = GemmCfg<BN>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave % Cfg::WN) * Cfg::TN * 32 + tn * 32 + (lane & 31);
}

// Bijective XCD-aware remap of a linear block id (guide T1): blocks that are consecutive after the
// remap run on the same XCD (round-robin dispatch), so they share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int lin, int total)
{
    const int q = total >> 3, r = total & 7, x = lin & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (lin >> 3);
}

// Remap that keeps the dispatch order of M tiles (so tiles of unequal cost, listed class by class, are
// spread evenly over the XCDs) while still putting the nNt column tiles of one M tile on the same XCD:
// blocks b and b + 8 run on one XCD under round-robin dispatch.
__device__ __forceinline__ int xcd_pair_remap(int lin, int total, int nNt)
{
    const int group = 8 * nNt;
    const int full = (total / group) * group;
    if (lin >= full) return lin;
    const int base = (lin / group) * group, r = lin - base;
    return base + (r & 7) * nNt + (r >> 3);
}

}  // namespace mq
