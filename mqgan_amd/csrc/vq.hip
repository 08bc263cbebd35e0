// Vector-quantiser kernels for gfx950: FSQ (the reference's quantiser) and the generic L2
// nearest-neighbour search + codebook EMA (north-star micro-benchmark shapes).
#include <type_traits>

#include "gemm_core.h"

namespace mq {

// ------------------------------------------------------------------------------------------ FSQ
// Index search by thresholds: the reference pipeline  z -> rint(f32(f32(tanh(z+shift))*half_l) - offset)
// is a monotone step function of z with L-1 jumps.  The host finds, by bisection over float bit
// patterns with exactly that arithmetic (tanh evaluated in double), the smallest float theta_k with
// level >= k, and the kernel just counts  z >= theta_k.  No transcendental, pure streaming, and
// bit-identical to oracle/vq_ref.c by construction.
struct FsqDev {
    int d;
    int nthr[8];       // levels - 1
    int half_width[8];
    int basis[8];
    float thresh[8][8];
    float half_l[8];
    float shift[8];
};

template <int D>
__device__ __forceinline__ int fsq_quantise(const FsqDev &p, const float (&zv)[D], float (&cv)[D])
{
    int id = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        if (D != 4 && j >= p.d) break;
        int lvl = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < p.nthr[j]) lvl += (zv[j] >= p.thresh[j][k]) ? 1 : 0;
        const int q = lvl - p.half_width[j];
        cv[j] = (float)q / (float)p.half_width[j];
        id += lvl * p.basis[j];
    }
    return id;
}

// Streaming kernel: D == 4 takes four vectors per thread and iteration - the four 16-byte loads are issued before
// any arithmetic so that enough bytes are in flight to cover HBM latency; inputs and outputs are touched once, so
// they bypass the caches' allocation (non-temporal).
template <int D, int U = 4, bool NT = true>
__global__ void fsq_forward_kernel(const float *__restrict__ z, int64_t n, FsqDev p, float *__restrict__ codes,
                                   int32_t *__restrict__ idx)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (D == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 *zin = reinterpret_cast<const f4 *>(z);
        f4 *cout = reinterpret_cast<f4 *>(codes);
        for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += U * stride) {
            f4 t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + u * stride;
                t[u] = i < n ? __builtin_nontemporal_load(zin + i) : f4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + u * stride;
                if (i >= n) break;
                const float zv[D] = {t[u].x, t[u].y, t[u].z, t[u].w};
                float cv[D];
                const int id = fsq_quantise<D>(p, zv, cv);
                if (NT) {
                    if (codes) __builtin_nontemporal_store(f4{cv[0], cv[1], cv[2], cv[3]}, cout + i);
                    if (idx) __builtin_nontemporal_store(id, idx + i);
                } else {
                    if (codes) cout[i] = f4{cv[0], cv[1], cv[2], cv[3]};
                    if (idx) idx[i] = id;
                }
            }
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float zv[D], cv[D];
        for (int j = 0; j < D; ++j) zv[j] = j < p.d ? z[i * p.d + j] : 0.f;
        const int id = fsq_quantise<D>(p, zv, cv);
        if (codes)
            for (int j = 0; j < p.d; ++j) codes[i * p.d + j] = cv[j];
        if (idx) idx[i] = id;
    }
}

__global__ void fsq_backward_kernel(const float *__restrict__ z, const float *__restrict__ dc, int64_t total, FsqDev p,
                                    float *__restrict__ dz)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % p.d);
        const float x = z[i] + p.shift[j];
        const float e = __expf(-2.0f * fabsf(x));
        const float r = __builtin_amdgcn_rcpf(1.0f + e);
        const float sech2 = 4.0f * e * r * r;                       // 1 - tanh^2, no cancellation
        dz[i] = dc[i] * (p.half_l[j] * sech2 / (float)p.half_width[j]);
    }
}

__global__ void fsq_decode_kernel(const int64_t *__restrict__ idx, int64_t n, FsqDev p, float *__restrict__ codes)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t id = idx[i];
        for (int j = 0; j < p.d; ++j) {
            const int l = (int)((id / p.basis[j]) % (p.nthr[j] + 1));
            codes[i * p.d + j] = (float)(l - p.half_width[j]) / (float)p.half_width[j];
        }
    }
}

// ------------------------------------------------------------------------------------------ L2 nearest neighbour
__global__ void codebook_norms_kernel(const float *__restrict__ c, int k, int d, float *__restrict__ cn)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    float acc = 0.f;
    for (int t = 0; t < d; ++t) acc = fmaf(c[(size_t)j * d + t], c[(size_t)j * d + t], acc);
    cn[j] = acc;
}

// Row loader of the streaming search loop: rows of a dense [rows][d] matrix, K-contiguous.  stage() is called once per
// K-step in order and keeps its own position: k runs 0, 16, ... over one pass, then wraps and the loader moves on by
// `tile_stride` bytes (the next 128 codes; 0 for the x tile, which is simply re-read).  Everything outside the buffer
// descriptor - rows past the end, code tiles past the block's range - reads as zeros.
struct StreamRowLoader {
    __amdgpu_buffer_rsrc_t rsrc;
    int d, nk, kstep;
    unsigned rowoff[2], off_[2], tile_off, tile_stride;
    __device__ __forceinline__ void init(const float *base, int64_t rows_here, int d_, int nk_, unsigned tile_stride_)
    {
        d = d_;
        nk = nk_;
        kstep = 0;
        tile_off = 0;
        tile_stride = tile_stride_;
        rsrc = make_rsrc(base, (size_t)(rows_here > 0 ? rows_here : 0) * d_ * sizeof(float));
#pragma unroll
        for (int j = 0; j < 2; ++j) rowoff[j] = (unsigned)((threadIdx.x >> 2) + 64 * j) * (unsigned)d_ * 4u;
    }
    __device__ __forceinline__ void stage(int)
    {
        const int kbase = kstep * GEMM_BK + 4 * (threadIdx.x & 3);
#pragma unroll
        for (int j = 0; j < 2; ++j) off_[j] = kbase < d ? rowoff[j] + tile_off + (unsigned)kbase * 4u : OOB;
        ++kstep;
        const bool wrap = kstep >= nk;                  // wave-uniform
        kstep = wrap ? 0 : kstep;
        tile_off += wrap ? tile_stride : 0u;
    }
    __device__ __forceinline__ float4 fetch(int j) const { return buf_load4(rsrc, off_[j]); }
};

__device__ __forceinline__ unsigned long long pack_dist_idx(float dist, int idx)
{
    unsigned u = __builtin_bit_cast(unsigned, dist);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // order-preserving map float -> uint
    return ((unsigned long long)u << 32) | (unsigned)idx;
}

// distances of a finished code tile -> running minimum; clears the accumulators.
// IDX16: the running indices are kept as 16-bit halves (two per register; codebooks of at most 65535 codes), which brings
// the kernel under the register budget of three blocks per CU.
template <bool IDX16>
__device__ __forceinline__ void argmin_fold(f32x16 (&acc)[GemmCfg<128>::TM][GemmCfg<128>::TN], float (&best)[GemmCfg<128>::TM][16],
                                            unsigned (&bidx)[GemmCfg<128>::TM][IDX16 ? 8 : 16], int tile, int k,
                                            const float *__restrict__ cn)
{
    using Cfg = GemmCfg<128>;
#pragma unroll
    for (int tn = 0; tn < Cfg::TN; ++tn) {
        const int col = tile * 128 + acc_col<128>(tn);
        const bool cv = col < k;
        const float cnv = cv ? cn[col] : 0.f;
#pragma unroll
        for (int tm = 0; tm < Cfg::TM; ++tm) {
            f32x16 v = acc[tm][tn];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float dist = fmaf(-2.0f, v[r], cnv);
                dist = dist != dist ? -INFINITY : dist;      // NaN counts as -inf: the first NaN column wins
                const bool take = cv && dist < best[tm][r];  // strict: lowest index wins ties inside a lane
                best[tm][r] = take ? dist : best[tm][r];
                if constexpr (IDX16) {
                    unsigned &b = bidx[tm][r >> 1];
                    const unsigned repl = (r & 1) ? ((b & 0x0000ffffu) | ((unsigned)col << 16)) : ((b & 0xffff0000u) | (unsigned)col);
                    b = take ? repl : b;
                } else {
                    bidx[tm][r] = take ? (unsigned)col : bidx[tm][r];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = 0.f;
            acc[tm][tn] = v;
        }
    }
}

// grid (row tiles, code splits).  Each block: 128 rows x its range of 128-code tiles, as ONE software-pipelined loop
// over (code tile, k): the loads of the next code tile are already in flight while the distances of the finished one
// are folded into the running minimum (r1 restarted the pipeline - first loads, first LDS fill, barrier - once per code
// tile).  Measured: the same 88-103 TFLOP/s as before - the kernel is bound by its register footprint (64 accumulators +
// 64 running (distance, index) registers: 2 blocks per CU), not by the restarts.  The accumulation order inside a tile is
// unchanged (k in natural order, see gemm_core.h), so indices and distances stay bit-exact against oracle/vq_ref.c.
template <bool IDX16>
__global__ __launch_bounds__(256, IDX16 ? 3 : 2) void l2_argmin_kernel(const float *__restrict__ x, const float *__restrict__ cb,
                                                        const float *__restrict__ cn, int64_t n, int k, int d,
                                                        int tiles_per_split, unsigned long long *__restrict__ packed)
{
    constexpr int BN = 128;
    using Cfg = GemmCfg<BN>;
    __shared__ __attribute__((aligned(16))) float smem[Cfg::LDS_FLOATS];
    const int64_t m0 = (int64_t)blockIdx.x * GEMM_BM;
    const int ntiles = (k + BN - 1) / BN;
    const int t_begin = blockIdx.y * tiles_per_split;
    const int t_end = min(ntiles, t_begin + tiles_per_split);
    const int nk = (d + GEMM_BK - 1) / GEMM_BK;

    float best[Cfg::TM][16];
    unsigned bidx[Cfg::TM][IDX16 ? 8 : 16];
#pragma unroll
    for (int a = 0; a < Cfg::TM; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) best[a][r] = INFINITY;
#pragma unroll
        for (int r = 0; r < (IDX16 ? 8 : 16); ++r) bidx[a][r] = IDX16 ? 0xffffffffu : 0x7fffffffu;
    }
    f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int a = 0; a < Cfg::TM; ++a)
#pragma unroll
        for (int b = 0; b < Cfg::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    StreamRowLoader la, lb;
    la.init(x + m0 * d, n - m0 < 128 ? n - m0 : 128, d, nk, 0u);
    const int64_t code0 = (int64_t)t_begin * BN;
    const int64_t codes_here = min((int64_t)k, (int64_t)t_end * BN) - code0;
    lb.init(cb + code0 * d, codes_here, d, nk, (unsigned)BN * (unsigned)d * 4u);

    // ---- the mainloop of gemm_core.h (KORDER image, three pinned phases per K-step) over nk * (t_end - t_begin) steps
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int li = lane & 31, lh = lane >> 5;
    float *As = smem;
    float *Bs = smem + 2 * GEMM_BM * GEMM_LD;
    const int ld_row = tid >> 2, ld_kq = tid & 3;
    float4 ra[2], rb[2];
    la.stage(0);
    lb.stage(0);
#pragma unroll
    for (int j = 0; j < 2; ++j) { ra[j] = la.fetch(j); rb[j] = lb.fetch(j); }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        lds_store_k4<true>(&As[(ld_row + 64 * j) * GEMM_LD + 4 * ld_kq], ra[j]);
        lds_store_k4<true>(&Bs[(ld_row + 64 * j) * GEMM_LD + 4 * ld_kq], rb[j]);
    }
    la.stage(0);
    lb.stage(0);
    __syncthreads();
    int tile = t_begin, kin = 0;                         // code tile and K-step inside it of the step being computed
    auto k_step = [&](auto cur_tag) {
        constexpr int cur = decltype(cur_tag)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) { ra[j] = la.fetch(j); rb[j] = lb.fetch(j); }
        __builtin_amdgcn_sched_barrier(0);
        const float *Ac = As + cur * GEMM_BM * GEMM_LD + (wm * Cfg::TM * 32 + li) * GEMM_LD + 2 * lh;
        const float *Bc = Bs + cur * BN * GEMM_LD + (wn * Cfg::TN * 32 + li) * GEMM_LD + 2 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float2 a[Cfg::TM], b[Cfg::TN];
#pragma unroll
            for (int t = 0; t < Cfg::TM; ++t) a[t] = *reinterpret_cast<const float2 *>(Ac + t * 32 * GEMM_LD + 4 * q);
#pragma unroll
            for (int t = 0; t < Cfg::TN; ++t) b[t] = *reinterpret_cast<const float2 *>(Bc + t * 32 * GEMM_LD + 4 * q);
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < Cfg::TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].x, b[tn].x, acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < Cfg::TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].y, b[tn].y, acc[tm][tn], 0, 0, 0);
        }
        la.stage(0);
        lb.stage(0);
        __builtin_amdgcn_sched_barrier(0);
        float *An = As + (cur ^ 1) * GEMM_BM * GEMM_LD;
        float *Bn = Bs + (cur ^ 1) * BN * GEMM_LD;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            lds_store_k4<true>(&An[(ld_row + 64 * j) * GEMM_LD + 4 * ld_kq], ra[j]);
            lds_store_k4<true>(&Bn[(ld_row + 64 * j) * GEMM_LD + 4 * ld_kq], rb[j]);
        }
        __syncthreads();
        if (++kin == nk) {                               // wave-uniform
            argmin_fold<IDX16>(acc, best, bidx, tile, k, cn);
            kin = 0;
            ++tile;
        }
    };
    const int total = nk * (t_end - t_begin);
    int st = 0;
    for (; st + 1 < total; st += 2) {
        k_step(std::integral_constant<int, 0>{});
        k_step(std::integral_constant<int, 1>{});
    }
    if (st < total) k_step(std::integral_constant<int, 0>{});

    // reduce over the 32 lanes that hold different columns of the same row, then one atomic per row
#pragma unroll
    for (int tm = 0; tm < Cfg::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int bi;
            if constexpr (IDX16) {
                const unsigned h = (r & 1) ? (bidx[tm][r >> 1] >> 16) : (bidx[tm][r >> 1] & 0xffffu);
                bi = h == 0xffffu ? 0x7fffffff : (int)h;          // 0xffff: this lane never saw a valid column (k <= 65535)
            } else {
                bi = (int)bidx[tm][r];
            }
            unsigned long long p = pack_dist_idx(best[tm][r], bi);
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                const unsigned long long o = __shfl_xor(p, off, 64);
                p = o < p ? o : p;
            }
            const int64_t m = m0 + acc_row<BN>(tm, r);
            if ((lane & 31) == 0 && m < n) atomicMin(&packed[m], p);
        }
}

__global__ void l2_unpack_kernel(const unsigned long long *__restrict__ packed, int64_t n, int32_t *__restrict__ idx,
                                 float *__restrict__ dist)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long p = packed[i];
        idx[i] = (int32_t)(p & 0xffffffffu);
        if (dist) {
            unsigned u = (unsigned)(p >> 32);
            u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
            dist[i] = __builtin_bit_cast(float, u);
        }
    }
}

// ------------------------------------------------------------------------------------------ codebook EMA
// Arithmetic and summation order are DEFINED in oracle/vq_ref.c (section 3 of its header); every step below is one
// correctly rounded f32 operation (the _rn intrinsics keep the compiler from contracting them), so the result is
// bit-exact against that definition and identical from run to run (no float atomics).
constexpr int EMA_SEG = 2048;         // rows per segment
constexpr int EMA_DCH = 1024;         // feature columns per pass of a wave (16 registers per lane)

// grid (k, segments), one wave per (code, segment): part[s][c][:] = sum of the segment's rows with idx == c, in row order
__global__ __launch_bounds__(64) void ema_partial_kernel(const float *__restrict__ x, const int32_t *__restrict__ idx, int64_t n,
                                                         int k, int d, float *__restrict__ part, int *__restrict__ counts)
{
    const int c = blockIdx.x, s = blockIdx.y, lane = threadIdx.x;
    const int64_t i0 = (int64_t)s * EMA_SEG, i1 = min(n, i0 + EMA_SEG);
    float *o = part + ((size_t)s * k + c) * d;
    int cnt = 0;
    for (int t0 = 0; t0 < d; t0 += EMA_DCH) {
        float acc[EMA_DCH / 64];
#pragma unroll
        for (int j = 0; j < EMA_DCH / 64; ++j) acc[j] = 0.f;
        for (int64_t i = i0; i < i1; i += 64) {
            const bool hit = (i + lane < i1) && idx[i + lane] == c;
            unsigned long long m = __ballot(hit);
            if (t0 == 0) cnt += __popcll(m);
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float *row = x + (size_t)(i + b) * d + t0;
#pragma unroll
                for (int j = 0; j < EMA_DCH / 64; ++j) {
                    const int t = lane + 64 * j;
                    if (t0 + t < d) acc[j] = __fadd_rn(acc[j], row[t]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < EMA_DCH / 64; ++j) {
            const int t = lane + 64 * j;
            if (t0 + t < d) o[t0 + t] = acc[j];
        }
    }
    if (lane == 0 && cnt) atomicAdd(&counts[c], cnt);       // integer: exact and order-independent
}
// one block of 1024: cluster sizes, then their total in the fixed order of the definition
__global__ __launch_bounds__(1024) void ema_sizes_kernel(const int *__restrict__ counts, int k, float decay,
                                                         float *__restrict__ cluster_size, float *__restrict__ total)
{
    __shared__ float lanes[1024];
    const float omd = __fsub_rn(1.0f, decay);
    float s = 0.f;
    for (int c = threadIdx.x; c < k; c += 1024) {
        const float v = __fmaf_rn(omd, (float)counts[c], __fmul_rn(cluster_size[c], decay));
        cluster_size[c] = v;
        s = __fadd_rn(s, v);
    }
    lanes[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int l = 0; l < 1024; ++l) t = __fadd_rn(t, lanes[l]);
        total[0] = t;
    }
}
__global__ void ema_codebook_kernel(const float *__restrict__ part, int nseg, const float *__restrict__ cluster_size,
                                    const float *__restrict__ total, int k, int d, float decay, float eps,
                                    float *__restrict__ embed_sum, float *__restrict__ codebook)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t kd = (int64_t)k * d;
    if (i >= kd) return;
    const int c = (int)(i / d);
    float sum = 0.f;
    for (int s = 0; s < nseg; ++s) sum = __fadd_rn(sum, part[(size_t)s * kd + i]);
    const float es = __fmaf_rn(__fsub_rn(1.0f, decay), sum, __fmul_rn(embed_sum[i], decay));
    embed_sum[i] = es;
    const float tot = total[0];
    const float den = __fadd_rn(tot, __fmul_rn((float)k, eps));
    const float sm = __fmul_rn(__fdiv_rn(__fadd_rn(cluster_size[c], eps), den), tot);      // Laplace smoothing
    codebook[i] = __fdiv_rn(es, sm);
}

static int to_dev(const mqFsqDesc *h, FsqDev &p, const float (*thresh)[8], const char *who)
{
    MQ_REQUIRE(h && h->d >= 1 && h->d <= 8, "%s: d must be 1..8", who);
    p.d = h->d;
    for (int j = 0; j < 8; ++j) {
        const bool on = j < h->d;
        MQ_REQUIRE(!on || (h->levels[j] >= 2 && h->levels[j] <= 9), "%s: levels[%d]=%d out of range 2..9", who, j,
                   h->levels[j]);
        p.nthr[j] = on ? h->levels[j] - 1 : 0;
        p.half_width[j] = on ? h->half_width[j] : 1;
        p.basis[j] = on ? h->basis[j] : 0;
        p.half_l[j] = on ? h->half_l[j] : 0.f;
        p.shift[j] = on ? h->shift[j] : 0.f;
        for (int k = 0; k < 8; ++k) p.thresh[j][k] = (on && thresh) ? thresh[j][k] : INFINITY;
    }
    return 0;
}

}  // namespace mq

using namespace mq;

extern "C" int mq_fsq_forward(const mqFsqDesc *desc, const float *z, int64_t n, float *codes, int32_t *idx,
                              mqStream stream)
{
    FsqDev p;
    if (to_dev(desc, p, desc ? desc->thresh : nullptr, "mq_fsq_forward")) return 1;
    if (n == 0) return 0;
    MQ_REQUIRE(z && n > 0 && (codes || idx), "mq_fsq_forward: bad arguments");
    if (p.d == 4 && (uintptr_t)z % 16 == 0 && (uintptr_t)codes % 16 == 0) {
        // two vectors per thread and one block per 512 vectors: 5.8 TB/s at 16.7 M vectors; four per thread on 2,048 looping blocks
        // (r1-r3) 4.9, eight per thread 4.7, plain instead of non-temporal stores -5 % (tools/fsq_micro.py, r4)
        hipLaunchKernelGGL((fsq_forward_kernel<4, 2, true>), dim3((unsigned)std::min<int64_t>(cdiv(n, 512), 1 << 20)), dim3(256), 0, S(stream),
                           z, n, p, codes, idx);
        MQ_LAUNCH_CHECK("mq_fsq_forward");
        return 0;
    }
    hipLaunchKernelGGL((fsq_forward_kernel<8>), dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), z, n, p, codes, idx);
    MQ_LAUNCH_CHECK("mq_fsq_forward");
    return 0;
}

extern "C" int mq_fsq_backward(const mqFsqDesc *desc, const float *z, const float *dcodes, int64_t n, float *dz,
                               mqStream stream)
{
    FsqDev p;
    if (to_dev(desc, p, nullptr, "mq_fsq_backward")) return 1;
    if (n == 0) return 0;
    MQ_REQUIRE(z && dcodes && dz && n > 0, "mq_fsq_backward: bad arguments");
    hipLaunchKernelGGL(fsq_backward_kernel, dim3(ew_grid(n * p.d, 256)), dim3(256), 0, S(stream), z, dcodes, n * p.d, p,
                       dz);
    MQ_LAUNCH_CHECK("mq_fsq_backward");
    return 0;
}

extern "C" int mq_fsq_decode(const mqFsqDesc *desc, const int64_t *idx, int64_t n, float *codes, mqStream stream)
{
    FsqDev p;
    if (to_dev(desc, p, nullptr, "mq_fsq_decode")) return 1;
    if (n == 0) return 0;
    MQ_REQUIRE(idx && codes && n > 0, "mq_fsq_decode: bad arguments");
    hipLaunchKernelGGL(fsq_decode_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), idx, n, p, codes);
    MQ_LAUNCH_CHECK("mq_fsq_decode");
    return 0;
}

extern "C" int mq_l2_codebook_norms(const float *codebook, int k, int d, float *cnorm, mqStream stream)
{
    MQ_REQUIRE(codebook && cnorm && k > 0 && d > 0, "mq_l2_codebook_norms: bad arguments");
    hipLaunchKernelGGL(codebook_norms_kernel, dim3((unsigned)cdiv(k, 256)), dim3(256), 0, S(stream), codebook, k, d,
                       cnorm);
    MQ_LAUNCH_CHECK("mq_l2_codebook_norms");
    return 0;
}

extern "C" size_t mq_l2_argmin_workspace(int64_t n) { return (size_t)(n > 0 ? n : 0) * sizeof(unsigned long long); }

extern "C" int mq_l2_argmin(const float *x, const float *codebook, const float *cnorm, int64_t n, int k, int d,
                            int32_t *idx, float *best_dist, void *workspace, mqStream stream)
{
    MQ_REQUIRE(x && codebook && cnorm && idx && workspace, "mq_l2_argmin: null pointer");
    MQ_REQUIRE(n >= 0 && k > 0 && d > 0 && d % 4 == 0, "mq_l2_argmin: need n >= 0, k > 0, d %% 4 == 0 (d=%d)", d);
    MQ_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)codebook % 16 == 0, "mq_l2_argmin: x / codebook must be 16-byte aligned");
    if (n == 0) return 0;
    hipError_t e = hipMemsetAsync(workspace, 0xFF, (size_t)n * 8, S(stream));
    MQ_REQUIRE(e == hipSuccess, "mq_l2_argmin: memset failed: %s", hipGetErrorString(e));
    const int row_tiles = (int)cdiv(n, GEMM_BM);
    const int ntiles = (int)cdiv(k, 128);
    int splits = (int)cdiv(768, row_tiles);
    if (splits > ntiles) splits = ntiles;
    if (splits < 1) splits = 1;
    const int tps = (int)cdiv(ntiles, splits);
    splits = (int)cdiv(ntiles, tps);
    if (k <= 65535)
        hipLaunchKernelGGL(l2_argmin_kernel<true>, dim3(row_tiles, splits), dim3(256), 0, S(stream), x, codebook, cnorm, n, k, d, tps,
                           reinterpret_cast<unsigned long long *>(workspace));
    else
        hipLaunchKernelGGL(l2_argmin_kernel<false>, dim3(row_tiles, splits), dim3(256), 0, S(stream), x, codebook, cnorm, n, k, d, tps,
                           reinterpret_cast<unsigned long long *>(workspace));
    hipLaunchKernelGGL(l2_unpack_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream),
                       reinterpret_cast<const unsigned long long *>(workspace), n, idx, best_dist);
    MQ_LAUNCH_CHECK("mq_l2_argmin");
    return 0;
}

static int ema_segments(int64_t n) { return (int)cdiv(n, EMA_SEG); }

extern "C" size_t mq_codebook_ema_workspace(int64_t n, int k, int d)
{
    if (n <= 0 || k <= 0 || d <= 0) return 0;
    return ((size_t)ema_segments(n) * k * d + 4) * sizeof(float) + (size_t)k * sizeof(int);
}

extern "C" int mq_codebook_ema_update(const float *x, const int32_t *idx, int64_t n, int k, int d, float decay, float eps,
                                      float *cluster_size, float *embed_sum, float *codebook, void *workspace,
                                      size_t workspace_bytes, mqStream stream)
{
    MQ_REQUIRE(x && idx && cluster_size && embed_sum && codebook && workspace, "mq_codebook_ema_update: null pointer");
    MQ_REQUIRE(n > 0 && k > 0 && d > 0 && k <= 65535, "mq_codebook_ema_update: need n > 0, 0 < k <= 65535, d > 0");
    MQ_REQUIRE(workspace_bytes >= mq_codebook_ema_workspace(n, k, d), "mq_codebook_ema_update: workspace too small");
    const int nseg = ema_segments(n);
    MQ_REQUIRE(nseg <= 65535, "mq_codebook_ema_update: n too large");
    float *part = reinterpret_cast<float *>(workspace);
    float *total = part + (size_t)nseg * k * d;
    int *counts = reinterpret_cast<int *>(total + 4);
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)k * sizeof(int), S(stream));
    MQ_REQUIRE(e == hipSuccess, "mq_codebook_ema_update: memset failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(ema_partial_kernel, dim3(k, nseg), dim3(64), 0, S(stream), x, idx, n, k, d, part, counts);
    hipLaunchKernelGGL(ema_sizes_kernel, dim3(1), dim3(1024), 0, S(stream), counts, k, decay, cluster_size, total);
    hipLaunchKernelGGL(ema_codebook_kernel, dim3((unsigned)cdiv((int64_t)k * d, 256)), dim3(256), 0, S(stream), part, nseg,
                       cluster_size, total, k, d, decay, eps, embed_sum, codebook);
    MQ_LAUNCH_CHECK("mq_codebook_ema_update");
    return 0;
}
