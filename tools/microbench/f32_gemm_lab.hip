// f32-MFMA grouped NT GEMM lab (round 6): what holds conv_nt_kernel<128,false,4> at 0.80-0.84 of the f32 matrix peak?
//   C[g][m][n] = sum_k A[g][m][k] * B[g][n][k]      (the Winograd-domain product: 42 groups, M = tiles, N = K_out, K = C_in)
// Variants (argv[1]):
//   0  production loop: gemm_core.h nt_gemm_mainloop<128, false, true> (LD 18, ds_read_b64 fragments, ds_write_b64 staging), 4 blocks / CU
//   1  swizzled 64-byte LDS rows, ds_read_b128 fragments (4 k's per lane), ds_write_b128 staging, 128 x 128 tile, 4 blocks / CU
//   2  as 1, block tile 128 x 256 (wave tile 64 x 128), 2 blocks / CU
//   3  as 1, block tile 256 x 256 (wave tile 128 x 128), 1 block / CU
//   4  as 1, block tile 256 x 128 (wave tile 128 x 64), 2 blocks / CU
//   7 / 8  one wave per SIMD at 256 x 256 / two at 256 x 128, software-pipelined (fragments of the next group and the LDS writes of the next tile
//          between the MFMAs of this one, the K-step's barrier between its two MFMA groups)
//   11 / 12  256 x 128 with both operands staged by LDS-DMA (buffer_load ... lds), 2 / 3 stages
// Ablations (argv[2], bit mask, results WRONG on purpose): 1 = fragments read once (no LDS reads in the loop), 2 = no global loads /
// LDS stores in the loop, 4 = no barrier.
// Every block records s_memtime (shader cycles) and s_memrealtime (100 MHz) around its main loop: the clock the chip held and
// the shader cycles one K-step took on a SIMD are printed beside the TF/s.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I mqgan_amd/csrc tools/microbench/f32_gemm_lab.hip -o build_exp/f32_gemm_lab
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gemm_core.h"

namespace mq {
thread_local char g_err[512] = "";
thread_local char g_kernel[128] = "";
}
using namespace mq;

struct RowLoader {            // DenseRowLoader of igemm.hip, NS slots of 64 rows
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned row_[4], off_[4];
    int Kdim, soff, ns;
    __device__ __forceinline__ void init(const float *base, int rows, int Kdim_, int ns_)
    {
        Kdim = Kdim_;
        ns = ns_;
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
        const bool kv = k0 < Kdim;
        soff = kv ? k0 * 4 : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) off_[j] = kv ? row_[j] : OOB;
    }
    __device__ __forceinline__ float4 fetch(int j) const { return buf_load4s(rsrc, off_[j], soff); }
};

struct Stamp { unsigned long long c0, c1, w0, w1, e0, e1, e2; };       // loop: shader cycles, 100 MHz ticks; whole block: 100 MHz ticks

template <int ABL>
struct Abl { static constexpr bool NOFRAG = ABL & 1, NOLOAD = ABL & 2, NOBAR = ABL & 4; };

// ---- the new loop: BM x BN block tile, WM x WN waves, swizzled 16-float rows, b128 fragments
template <int BM, int BN, int WM, int WN, int ABL>
__device__ __forceinline__ void loop_b128(RowLoader &la, RowLoader &lb, int nk, float *smem, f32x16 (&acc)[BM / WM / 32][BN / WN / 32])
{
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, AS = BM / 64, BS = BN / 64;
    using A = Abl<ABL>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    float *As = smem, *Bs = smem + 2 * BM * 16;
    const int ld_row = tid >> 2, ld_kq = tid & 3;
    const int st_off = ld_row * 16 + 4 * (ld_kq ^ ((ld_row >> 2) & 3));            // rows ld_row + 64 j share (row >> 2) & 3
    const int sw = (li >> 2) & 3;
    const int fo0 = 4 * ((0 + lh) ^ sw), fo1 = 4 * ((2 + lh) ^ sw);
    float4 ra[AS], rb[BS];
    la.stage(0);
    lb.stage(0);
#pragma unroll
    for (int j = 0; j < AS; ++j) ra[j] = la.fetch(j);
#pragma unroll
    for (int j = 0; j < BS; ++j) rb[j] = lb.fetch(j);
#pragma unroll
    for (int j = 0; j < AS; ++j) *reinterpret_cast<float4 *>(&As[st_off + 64 * 16 * j]) = ra[j];
#pragma unroll
    for (int j = 0; j < BS; ++j) *reinterpret_cast<float4 *>(&Bs[st_off + 64 * 16 * j]) = rb[j];
    la.stage(16);
    lb.stage(16);
    __syncthreads();
    float4 fa[2][TM], fb[2][TN];
    auto k_step = [&](int kt, auto cur_tag, auto first_tag) {
        constexpr int cur = decltype(cur_tag)::value;
        constexpr bool first = decltype(first_tag)::value;
        if (!A::NOLOAD) {
#pragma unroll
            for (int j = 0; j < AS; ++j) ra[j] = la.fetch(j);
#pragma unroll
            for (int j = 0; j < BS; ++j) rb[j] = lb.fetch(j);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float *Ac = As + cur * BM * 16 + (wm * TM * 32 + li) * 16;
        const float *Bc = Bs + cur * BN * 16 + (wn * TN * 32 + li) * 16;
        if (!A::NOFRAG || first) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int t = 0; t < TM; ++t) fa[s][t] = *reinterpret_cast<const float4 *>(Ac + t * 32 * 16 + (s ? fo1 : fo0));
#pragma unroll
                for (int t = 0; t < TN; ++t) fb[s][t] = *reinterpret_cast<const float4 *>(Bc + t * 32 * 16 + (s ? fo1 : fo0));
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const float av = e == 0 ? fa[s][tm].x : e == 1 ? fa[s][tm].y : e == 2 ? fa[s][tm].z : fa[s][tm].w;
                        const float bv = e == 0 ? fb[s][tn].x : e == 1 ? fb[s][tn].y : e == 2 ? fb[s][tn].z : fb[s][tn].w;
                        if (first && s == 0 && e == 0) {
                            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, zero, 0, 0, 0);
                        } else {
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tm][tn], 0, 0, 0);
                        }
                    }
        }
        la.stage((kt + 2) * 16);
        lb.stage((kt + 2) * 16);
        __builtin_amdgcn_sched_barrier(0);
        if (!A::NOLOAD) {
            float *An = As + (cur ^ 1) * BM * 16, *Bn = Bs + (cur ^ 1) * BN * 16;
#pragma unroll
            for (int j = 0; j < AS; ++j) *reinterpret_cast<float4 *>(&An[st_off + 64 * 16 * j]) = ra[j];
#pragma unroll
            for (int j = 0; j < BS; ++j) *reinterpret_cast<float4 *>(&Bn[st_off + 64 * 16 * j]) = rb[j];
        }
        if (!A::NOBAR) __syncthreads();
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
}

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void store_tile(float *C, int M, int N, int m0, int n0, f32x16 (&acc)[BM / WM / 32][BN / WN / 32])
{
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int rows_left = M - m0 < BM ? M - m0 : BM;
    const unsigned pitch = (unsigned)N * 4u;
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(C + (size_t)m0 * N, (size_t)rows_left * pitch);
    const int lane_row = wm * TM * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wn * TN * 32 + tn * 32 + (lane & 31);
        const unsigned voff = n < N ? (unsigned)lane_row * pitch + (unsigned)n * 4u : 0x80000000u;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned rstep = (unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * pitch;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[tm][tn][r]), ro, (int)(voff + rstep), 0, 0);
            }
    }
}

template <int BM, int BN, int WM, int WN, int OCC, int ABL>
__global__ __launch_bounds__(256, OCC) void gemm_new(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N,
                                                     int K, Stamp *stamps)
{
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * 16];
    const unsigned long long e0 = (unsigned long long)wall_clock64();
    const int nNt = (N + BN - 1) / BN, nMt = (M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, nMt * nNt);
    const int mt = tile / nNt, nt = tile - mt * nNt;
    const int m0 = mt * BM, n0 = nt * BN, g = blockIdx.y;
    RowLoader la, lb;
    la.init(A + ((size_t)g * M + m0) * K, M - m0, K, BM / 64);
    lb.init(B + ((size_t)g * N + n0) * K, N - n0, K, BN / 64);
    f32x16 acc[BM / WM / 32][BN / WN / 32];
    unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    loop_b128<BM, BN, WM, WN, ABL>(la, lb, K / 16, smem, acc);
    unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (!(ABL & 8)) store_tile<BM, BN, WM, WN>(C + (size_t)g * M * N, M, N, m0, n0, acc);
    else if (M < 0) store_tile<BM, BN, WM, WN>(C + (size_t)g * M * N, M, N, m0, n0, acc);       // keeps the accumulators alive
    const unsigned long long e2 = (unsigned long long)wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && stamps) stamps[blockIdx.y * gridDim.x + blockIdx.x] = Stamp{c0, c1, w0, w1, e0, (unsigned long long)wall_clock64(), e2};
}


// ---- one wave per SIMD, software-pipelined (variant 7): block 256 x BN, 4 waves as 2 x 2, wave tile 128 x (BN / 2).
// K-step kt = two groups of MFMAs (the two float4 fragments of a lane: k's 4 lh .. + 3 and 8 + 4 lh .. + 3 of the 16):
//   group 0 on fragments (kt, 0): LDS writes of tile kt + 1 (global loads issued a group earlier), fragment reads (kt, 1); barrier
//   group 1 on fragments (kt, 1): global loads of tile kt + 2, fragment reads (kt + 1, 0) out of the buffer just written
// so no MFMA ever waits for an LDS read issued right in front of it, and the one barrier of a K-step sits between two groups.
template <int BN, int ABL>
__device__ __forceinline__ void loop_ws1(RowLoader &la, RowLoader &lb, int nk, float *smem, f32x16 (&acc)[4][BN / 64])
{
    constexpr int BM = 256, TM = 4, TN = BN / 64, AS = 4, BS = BN / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    float *As = smem, *Bs = smem + 2 * BM * 16;
    const int ld_row = tid >> 2, ld_kq = tid & 3;
    const int st_off = ld_row * 16 + 4 * (ld_kq ^ ((ld_row >> 2) & 3));
    const int sw = (li >> 2) & 3;
    const int fo[2] = {4 * ((0 + lh) ^ sw), 4 * ((2 + lh) ^ sw)};
    const float *Ar = As + (wm * 128 + li) * 16, *Br = Bs + (wn * (BN / 2) + li) * 16;
    float4 ra[AS], rb[BS];
    float4 fa[2][TM], fb[2][TN];
    auto frag = [&](int buf, int s, int t, bool isA) {
        if (isA) fa[s][t] = *reinterpret_cast<const float4 *>(Ar + buf * BM * 16 + t * 32 * 16 + fo[s]);
        else fb[s][t] = *reinterpret_cast<const float4 *>(Br + buf * BN * 16 + t * 32 * 16 + fo[s]);
    };
    auto put = [&](int buf, int j, bool isA) {
        if (isA) *reinterpret_cast<float4 *>(&As[buf * BM * 16 + st_off + 64 * 16 * j]) = ra[j];
        else *reinterpret_cast<float4 *>(&Bs[buf * BN * 16 + st_off + 64 * 16 * j]) = rb[j];
    };
    // prologue: tile 0 -> LDS buffer 0, tile 1 -> staging registers, fragments (0, 0)
    la.stage(0);
    lb.stage(0);
#pragma unroll
    for (int j = 0; j < AS; ++j) ra[j] = la.fetch(j);
#pragma unroll
    for (int j = 0; j < BS; ++j) rb[j] = lb.fetch(j);
#pragma unroll
    for (int j = 0; j < AS; ++j) put(0, j, true);
#pragma unroll
    for (int j = 0; j < BS; ++j) put(0, j, false);
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
    for (int t = 0; t < TM; ++t) frag(0, 0, t, true);
#pragma unroll
    for (int t = 0; t < TN; ++t) frag(0, 0, t, false);

    // 16 MFMAs of one k (element e of fragment s) over the wave tile
    auto mfma16 = [&](int s, int e, bool zero_first) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const float av = e == 0 ? fa[s][tm].x : e == 1 ? fa[s][tm].y : e == 2 ? fa[s][tm].z : fa[s][tm].w;
                const float bv = e == 0 ? fb[s][tn].x : e == 1 ? fb[s][tn].y : e == 2 ? fb[s][tn].z : fb[s][tn].w;
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
        // ---- group 0: MFMAs on (kt, 0); fragments (kt, 1) early, LDS writes of tile kt + 1 late
        mfma16(0, 0, first);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TM; ++t) frag(cur, 1, t, true);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(0, 1, false);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TN; ++t) frag(cur, 1, t, false);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(0, 2, false);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 2)) {
#pragma unroll
            for (int j = 0; j < AS; ++j) put(cur ^ 1, j, true);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma16(0, 3, false);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 2)) {
#pragma unroll
            for (int j = 0; j < BS; ++j) put(cur ^ 1, j, false);
        }
        __syncthreads();                                   // tile kt + 1 is in LDS; every read of tile kt has been issued and awaited
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 1: MFMAs on (kt, 1); global loads of tile kt + 2, fragments (kt + 1, 0) - the last of them a whole chunk of
        // MFMAs ahead of their first use
        mfma16(1, 0, false);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 2)) {
#pragma unroll
            for (int j = 0; j < AS; ++j) ra[j] = la.fetch(j);
        }
#pragma unroll
        for (int t = 0; t < TM; ++t) frag(cur ^ 1, 0, t, true);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(1, 1, false);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 2)) {
#pragma unroll
            for (int j = 0; j < BS; ++j) rb[j] = lb.fetch(j);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) frag(cur ^ 1, 0, t, false);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(1, 2, false);
        __builtin_amdgcn_sched_barrier(0);
        la.stage((kt + 3) * 16);
        lb.stage((kt + 3) * 16);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(1, 3, false);
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
}

template <int BN, int ABL>
__global__ __launch_bounds__(256, 1) void gemm_ws1(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N, int K,
                                                   Stamp *stamps)
{
    constexpr int BM = 256;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * 16];
    const unsigned long long e0 = (unsigned long long)wall_clock64();
    const int nNt = (N + BN - 1) / BN, nMt = (M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, nMt * nNt);
    const int mt = tile / nNt, nt = tile - mt * nNt;
    const int m0 = mt * BM, n0 = nt * BN, g = blockIdx.y;
    RowLoader la, lb;
    la.init(A + ((size_t)g * M + m0) * K, M - m0, K, 4);
    lb.init(B + ((size_t)g * N + n0) * K, N - n0, K, BN / 64);
    f32x16 acc[4][BN / 64];
    unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    loop_ws1<BN, ABL>(la, lb, K / 16, smem, acc);
    unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    store_tile<BM, BN, 2, 2>(C + (size_t)g * M * N, M, N, m0, n0, acc);
    const unsigned long long e2 = (unsigned long long)wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && stamps) stamps[blockIdx.y * gridDim.x + blockIdx.x] = Stamp{c0, c1, w0, w1, e0, (unsigned long long)wall_clock64(), e2};
}

// ---- LDS-DMA staging (variant 9): both operand tiles go global -> LDS by buffer_load_dwordx4 ... lds (no staging registers, no
// ds_write, no mid-loop vmcnt on a register): a wave's instruction writes 1 KB lane-linearly = 16 rows x 64 bytes; the chunk
// swizzle is applied to the SOURCE address (lane l of a piece fetches logical chunk (l & 3) ^ ((row >> 2) & 3) of its row).
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ lds_void_t *ldsp(const void *p) { return reinterpret_cast<lds_void_t *>(reinterpret_cast<uintptr_t>(p)); }

struct DmaRows {               // rows of a dense [rows][K] f32 matrix: piece j of wave w = tile rows 64 j + 16 w .. + 15
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned row_[4];
    int Kdim;
    __device__ __forceinline__ void init(const float *base, int rows, int Kdim_)
    {
        Kdim = Kdim_;
        rsrc = make_rsrc(base, (size_t)(rows < 0 ? 0 : rows) * Kdim_ * 4);
        const int tid = threadIdx.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (tid >> 2) + 64 * j;
            const int lc = (tid & 3) ^ ((tid >> 4) & 3);
            row_[j] = r < rows ? (unsigned)r * (unsigned)Kdim_ * 4u + 16u * lc : OOB;
        }
    }
    // tile k0 .. k0 + 15 of piece j -> 1 KB at dst (wave-uniform LDS address)
    __device__ __forceinline__ void issue(int j, int k0, float *dst) const
    {
        const bool kv = k0 < Kdim;            // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, ldsp(dst), 16, kv ? row_[j] : OOB, kv ? k0 * 4 : 0, 0, 0);
    }
};

template <int BM, int BN, int WM, int WN, int NST, int ABL>
__device__ __forceinline__ void loop_dma(DmaRows &la, DmaRows &lb, int nk, float *smem, f32x16 (&acc)[BM / WM / 32][BN / WN / 32])
{
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, AS = BM / 64, BS = BN / 64;
    constexpr int STAGE = (BM + BN) * 16;                       // floats per stage: A rows then B rows
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int sw = (li >> 2) & 3;
    const int fo0 = 4 * ((0 + lh) ^ sw), fo1 = 4 * ((2 + lh) ^ sw);
    auto issue = [&](int kt) {
        float *st = smem + (kt % NST) * STAGE;
#pragma unroll
        for (int j = 0; j < AS; ++j) la.issue(j, kt * 16, st + (64 * j + 16 * wave) * 16);
#pragma unroll
        for (int j = 0; j < BS; ++j) lb.issue(j, kt * 16, st + BM * 16 + (64 * j + 16 * wave) * 16);
    };
#pragma unroll
    for (int p = 0; p < NST - 1; ++p) issue(p);
    float4 fa[2][TM], fb[2][TN];
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt landed (the NST - 2 younger tiles may still be in flight), and everybody has finished reading tile kt - 1
        if (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AS + BS) * (NST - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        if (!(ABL & 2)) issue(kt + NST - 1);          // into the stage tile kt - 1 occupied
        const float *Ac = smem + (kt % NST) * STAGE + (wm * TM * 32 + li) * 16;
        const float *Bc = smem + (kt % NST) * STAGE + BM * 16 + (wn * TN * 32 + li) * 16;
        // LDS reads as inline asm: in front of a C++ load from LDS the compiler would wait for every LDS-DMA in flight
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const unsigned a = (unsigned)reinterpret_cast<uintptr_t>(Ac + t * 32 * 16 + (s ? fo1 : fo0));
                f32x4v v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
                fa[s][t] = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const unsigned a = (unsigned)reinterpret_cast<uintptr_t>(Bc + t * 32 * 16 + (s ? fo1 : fo0));
                f32x4v v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
                fb[s][t] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // tie every fragment register to the wait (the compiler sees no dependency between an asm read and an asm wait)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int t = 0; t < TM; ++t) asm volatile("" : "+v"(fa[s][t].x), "+v"(fa[s][t].y), "+v"(fa[s][t].z), "+v"(fa[s][t].w));
#pragma unroll
            for (int t = 0; t < TN; ++t) asm volatile("" : "+v"(fb[s][t].x), "+v"(fb[s][t].y), "+v"(fb[s][t].z), "+v"(fb[s][t].w));
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const float av = e == 0 ? fa[s][tm].x : e == 1 ? fa[s][tm].y : e == 2 ? fa[s][tm].z : fa[s][tm].w;
                        const float bv = e == 0 ? fb[s][tn].x : e == 1 ? fb[s][tn].y : e == 2 ? fb[s][tn].z : fb[s][tn].w;
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tm][tn], 0, 0, 0);
                    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int BM, int BN, int WM, int WN, int NST, int OCC, int ABL>
__global__ __launch_bounds__(256, OCC) void gemm_dma(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N,
                                                     int K, Stamp *stamps)
{
    __shared__ __attribute__((aligned(1024))) float smem[NST * (BM + BN) * 16];
    const unsigned long long e0 = (unsigned long long)wall_clock64();
    const int nNt = (N + BN - 1) / BN, nMt = (M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, nMt * nNt);
    const int mt = tile / nNt, nt = tile - mt * nNt;
    const int m0 = mt * BM, n0 = nt * BN, g = blockIdx.y;
    DmaRows la, lb;
    la.init(A + ((size_t)g * M + m0) * K, M - m0, K);
    lb.init(B + ((size_t)g * N + n0) * K, N - n0, K);
    f32x16 acc[BM / WM / 32][BN / WN / 32];
#pragma unroll
    for (int a = 0; a < BM / WM / 32; ++a)
#pragma unroll
        for (int b = 0; b < BN / WN / 32; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    loop_dma<BM, BN, WM, WN, NST, ABL>(la, lb, K / 16, smem, acc);
    unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    store_tile<BM, BN, WM, WN>(C + (size_t)g * M * N, M, N, m0, n0, acc);
    const unsigned long long e2 = (unsigned long long)wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && stamps) stamps[blockIdx.y * gridDim.x + blockIdx.x] = Stamp{c0, c1, w0, w1, e0, (unsigned long long)wall_clock64(), e2};
}

struct RowLoader2 : RowLoader {      // protocol adapter for the production loop (NS = 2)
    static constexpr int NS = 2;
};

__global__ __launch_bounds__(256, 4) void gemm_prod(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N, int K,
                                                    Stamp *stamps)
{
    __shared__ __attribute__((aligned(16))) float smem[GemmCfg<128, EngF32>::LDS_FLOATS];
    const unsigned long long e0 = (unsigned long long)wall_clock64();
    const int nNt = (N + 127) / 128, nMt = (M + 127) / 128;
    const int tile = xcd_remap(blockIdx.x, nMt * nNt);
    const int mt = tile / nNt, nt = tile - mt * nNt;
    const int m0 = mt * 128, n0 = nt * 128, g = blockIdx.y;
    RowLoader la, lb;
    la.init(A + ((size_t)g * M + m0) * K, M - m0, K, 2);
    lb.init(B + ((size_t)g * N + n0) * K, N - n0, K, 2);
    f32x16 acc[2][2];
    unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    nt_gemm_mainloop<128, false, true>(la, lb, K / 16, smem, acc);
    unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    store_tile<128, 128, 2, 2>(C + (size_t)g * M * N, M, N, m0, n0, acc);
    const unsigned long long e2 = (unsigned long long)wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && stamps) stamps[blockIdx.y * gridDim.x + blockIdx.x] = Stamp{c0, c1, w0, w1, e0, (unsigned long long)wall_clock64(), e2};
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class Launch>
static void run(const char *name, int G, int M, int N, int K, int BM, int BN, int occ, Launch launch, const float *A, const float *B, float *C, bool check)
{
    const int nblk = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    Stamp *st;
    CK(hipMalloc(&st, sizeof(Stamp) * nblk * G));
    // steady state: the chip's clock ramps over tens of milliseconds (a 0.5 ms kernel measured cold reads 1.5 GHz, a 1.8 ms one 2.05);
    // inside a training step these kernels run back to back for 190 ms.  Warm up for ~0.4 s, then time.
    launch(dim3(nblk, G), st);
    CK(hipDeviceSynchronize());
    hipEvent_t s, e;
    CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    float ms = 0.f;
    {
        CK(hipEventRecord(s));
        for (int i = 0; i < 3; ++i) launch(dim3(nblk, G), st);
        CK(hipEventRecord(e));
        CK(hipEventSynchronize(e));
        CK(hipEventElapsedTime(&ms, s, e));
        const int warm = (int)(400.0f / (ms / 3)) + 1;
        for (int i = 0; i < warm; ++i) launch(dim3(nblk, G), st);
    }
    const int reps = 20;
    CK(hipEventRecord(s));
    for (int i = 0; i < reps; ++i) launch(dim3(nblk, G), st);
    CK(hipEventRecord(e));
    CK(hipEventSynchronize(e));
    CK(hipEventElapsedTime(&ms, s, e));
    ms /= reps;
    std::vector<Stamp> h((size_t)nblk * G);
    CK(hipMemcpy(h.data(), st, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
    double cyc = 0, wall = 0, life = 0, pre = 0, sto = 0, ack = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (auto &v : h) {
        cyc += (double)(v.c1 - v.c0); wall += (double)(v.w1 - v.w0); life += (double)(v.e1 - v.e0);
        pre += (double)(v.w0 - v.e0); sto += (double)(v.e2 - v.w1); ack += (double)(v.e1 - v.e2);
        tmin = v.e0 < tmin ? v.e0 : tmin; tmax = v.e1 > tmax ? v.e1 : tmax;
    }
    const double span = (double)(tmax - tmin);            // 100 MHz ticks from the first block's entry to the last block's exit
    printf("    per block (wave 0): entry -> loop %.2f us, stores issued in %.2f us, acknowledged after another %.2f us\n", pre / h.size() / 100.0,
           sto / h.size() / 100.0, ack / h.size() / 100.0);
    printf("    blocks %zu: mean life %.1f us (loop %.1f us), kernel span %.1f us, resident blocks per CU (sum of lives / span / 256) %.2f, in the loop %.2f\n",
           h.size(), life / h.size() / 100.0, wall / h.size() / 100.0, span / 100.0, life / span / 256.0, wall / span / 256.0);
    const double mhz = cyc / wall * 100.0;
    const double cyc_per_kstep = cyc / h.size() / (K / 16);
    // ideal: one K-step of a block = (BM/32)(BN/32) tiles x 8 MFMAs x 64 cycles / 4 SIMDs, with `occ` blocks sharing the SIMDs
    const double ideal = (double)(BM / 32) * (BN / 32) * 8 * 64 / 4 * occ;
    const double tf = 2.0 * G * M * N * K / ms / 1e9;
    printf("%-34s G %d M %d N %d K %d: %.3f ms %.1f TF/s (%.3f of 157.3)  clock %.0f MHz  cycles / K-step of a block %.0f (MFMA-only at %d blocks/CU: %.0f, ratio %.3f)\n",
           name, G, M, N, K, ms, tf, tf / 157.3, mhz, cyc_per_kstep, occ, ideal, cyc_per_kstep / ideal);
    if (check) {
        std::vector<float> hc(64);
        CK(hipMemcpy(hc.data(), C, 64 * 4, hipMemcpyDeviceToHost));
        std::vector<float> ha((size_t)K), hb((size_t)K * 64);
        CK(hipMemcpy(ha.data(), A, K * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), B, (size_t)K * 64 * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int n = 0; n < 64 && n < N; ++n) {
            double r = 0;
            for (int k = 0; k < K; ++k) r += (double)ha[k] * hb[(size_t)n * K + k];
            worst = fmax(worst, fabs(r - hc[n]));
        }
        printf("    check row 0: max |err| %.3e\n", worst);
    }
    CK(hipFree(st));
}

static int LOWBITS = 0;
int main(int argc, char **argv)
{
    LOWBITS = getenv("LAB_LOWBITS") ? 1 : 0;
    const int var = argc > 1 ? atoi(argv[1]) : 0, abl = argc > 2 ? atoi(argv[2]) : 0;
    const int G = argc > 3 ? atoi(argv[3]) : 42, M = argc > 4 ? atoi(argv[4]) : 9216, N = argc > 5 ? atoi(argv[5]) : 512, K = argc > 6 ? atoi(argv[6]) : 512;
    float *A, *B, *C;
    CK(hipMalloc(&A, (size_t)G * M * K * 4));
    CK(hipMalloc(&B, (size_t)G * N * K * 4));
    CK(hipMalloc(&C, (size_t)G * M * N * 4));
    {
        std::vector<float> h((size_t)G * M * K);
        unsigned s = 12345u;
        for (auto &v : h) { s = s * 1664525u + 1013904223u; const unsigned a = s; s = s * 1664525u + 1013904223u; v = (((a >> 8) ^ (s >> 20)) & 0xFFFFFF) / 16777216.0f - 0.5f + (LOWBITS ? 0.f : 0.f); if (LOWBITS) v = (float)((int)(v * 65536.0f)) / 65536.0f; }
        CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        h.resize((size_t)G * N * K);
        for (auto &v : h) { s = s * 1664525u + 1013904223u; const unsigned a = s; s = s * 1664525u + 1013904223u; v = (((a >> 8) ^ (s >> 20)) & 0xFFFFFF) / 16777216.0f - 0.5f + (LOWBITS ? 0.f : 0.f); if (LOWBITS) v = (float)((int)(v * 65536.0f)) / 65536.0f; }
        CK(hipMemcpy(B, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    const bool chk = abl == 0;
#define NEW(BM, BN, WM, WN, OCC, ABL, NAME) \
    run(NAME, G, M, N, K, BM, BN, OCC, [&](dim3 grid, Stamp *st) { hipLaunchKernelGGL((gemm_new<BM, BN, WM, WN, OCC, ABL>), grid, dim3(256), 0, 0, A, B, C, M, N, K, st); }, A, B, C, chk)
    if (var == 0) run("prod 128x128 b64 LD18 occ4", G, M, N, K, 128, 128, 4, [&](dim3 grid, Stamp *st) { hipLaunchKernelGGL(gemm_prod, grid, dim3(256), 0, 0, A, B, C, M, N, K, st); }, A, B, C, chk);
    if (var == 1 && abl == 0) NEW(128, 128, 2, 2, 4, 0, "b128 128x128 occ4");
    if (var == 1 && abl == 1) NEW(128, 128, 2, 2, 4, 1, "b128 128x128 occ4 NOFRAG");
    if (var == 1 && abl == 2) NEW(128, 128, 2, 2, 4, 2, "b128 128x128 occ4 NOLOAD");
    if (var == 1 && abl == 3) NEW(128, 128, 2, 2, 4, 3, "b128 128x128 occ4 NOFRAG+NOLOAD");
    if (var == 1 && abl == 7) NEW(128, 128, 2, 2, 4, 7, "b128 128x128 occ4 NOFRAG+NOLOAD+NOBAR");
    if (var == 1 && abl == 8) NEW(128, 128, 2, 2, 4, 8, "b128 128x128 occ4 NOSTORE");
    if (var == 1 && abl == 10) NEW(128, 128, 2, 2, 4, 10, "b128 128x128 occ4 NOLOAD+NOSTORE");
#define WS1(BN, ABL, NAME) \
    run(NAME, G, M, N, K, 256, BN, 1, [&](dim3 grid, Stamp *st) { hipLaunchKernelGGL((gemm_ws1<BN, ABL>), grid, dim3(256), 0, 0, A, B, C, M, N, K, st); }, A, B, C, chk)
    if (var == 7 && abl == 0) WS1(256, 0, "ws1 256x256 pipelined");
    if (var == 7 && abl == 2) WS1(256, 2, "ws1 256x256 pipelined NOLOAD");
    if (var == 8 && abl == 0) WS1(128, 0, "ws1 256x128 pipelined");
#define DMA(BM, BN, NST, OCC, ABL, NAME) \
    run(NAME, G, M, N, K, BM, BN, OCC, [&](dim3 grid, Stamp *st) { hipLaunchKernelGGL((gemm_dma<BM, BN, 2, 2, NST, OCC, ABL>), grid, dim3(256), 0, 0, A, B, C, M, N, K, st); }, A, B, C, chk)
    if (var == 11 && abl == 0) DMA(256, 128, 2, 2, 0, "dma 256x128 2 stages occ2");
    if (var == 12 && abl == 0) DMA(256, 128, 3, 2, 0, "dma 256x128 3 stages occ2");
    if (var == 2 && abl == 0) NEW(128, 256, 2, 2, 2, 0, "b128 128x256 occ2");
    if (var == 3 && abl == 0) NEW(256, 256, 2, 2, 1, 0, "b128 256x256 occ1");
    if (var == 4 && abl == 0) NEW(256, 128, 2, 2, 2, 0, "b128 256x128 occ2");
    if (var == 5 && abl == 0) NEW(128, 128, 2, 2, 3, 0, "b128 128x128 occ3");
    if (var == 6 && abl == 0) NEW(128, 128, 2, 2, 2, 0, "b128 128x128 occ2");
    return 0;
}
