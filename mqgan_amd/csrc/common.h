// Shared device/host helpers for libmqgan_hip.so (gfx950 only).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#ifndef MQ_VARIANTS
#define MQ_VARIANTS 0      // 1: the test-variant build (Makefile: make variants) - rejected kernel variants and the building blocks as exports
#endif
#include "../../include/mqgan_hip.h"

namespace mq {

// ----------------------------------------------------------------------------- errors
extern thread_local char g_err[512];

inline int fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
inline int fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

// Name of the conv / GEMM kernel instantiation the last mq_conv_* call on this thread launched (mq_last_kernel()): the
// dispatch decides it, so whoever labels timings asks instead of re-deriving the decision.
extern thread_local char g_kernel[128];
inline void kname(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
inline void kname(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}
inline const char *tf(bool b) { return b ? "true" : "false"; }

#define MQ_REQUIRE(cond, ...)                      \
    do {                                           \
        if (!(cond)) return ::mq::fail(__VA_ARGS__); \
    } while (0)

#define MQ_LAUNCH_CHECK(name)                                                              \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess) return ::mq::fail("%s: launch failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

inline hipStream_t S(mqStream s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grid for a grid-stride elementwise kernel.  Until r4 the grid stopped at 256 CUs x 8 blocks ("enough to fill the chip"): one
// resident set of waves, each looping.  Measured in r4: streaming kernels run FASTER as many short-lived blocks than as few
// looping ones (FSQ 4.9 -> 5.8 TB/s at 32 k blocks; the config-2 step 189-192 -> 186.9 ms at a cap of 8 k or 32 k blocks), so the
// cap is 256 x 128 now (it only bounds the launch for huge tensors; MQGAN_EW_CAP overrides it for A/B runs).
inline int ew_grid(int64_t work_items, int block)
{
    int64_t g = cdiv(work_items, block);
    static const int64_t cap = getenv("MQGAN_EW_CAP") ? atoll(getenv("MQGAN_EW_CAP")) : 256 * 128;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ----------------------------------------------------------------------------- filter-row layout
// A filter row (all taps x channels of one output channel) is stored channel-BLOCKED:
//     index(rs, c) = ((c / CB) * RS + rs) * CB + c % CB,   CB = 16 when C % 16 == 0, else CB = C (= plain RSC).
// The GEMM reduction then walks the R*S taps of one 16-channel slice before moving to the next slice, so the
// 64-byte-per-pixel slices of the activation tile are re-read from L1/L2 by consecutive K-steps instead of
// once per tap from the fabric (measured: 51x the tensor size fetched with the plain order).
__host__ __device__ inline int chan_block(int C) { return (C % 16 == 0) ? 16 : C; }
__host__ __device__ inline int kidx(int rs, int c, int RS, int CB)
{
    const int cb = c / CB;
    return (cb * RS + rs) * CB + (c - cb * CB);
}

// ----------------------------------------------------------------------------- wave64 reductions (DPP)
// sum over the 64 lanes, result valid in lane 63 (row_bcast form, gfx9 family).
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND_CTRL = false>
__device__ __forceinline__ float dpp_mov(float old, float v)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK,
                                           BANK_MASK, BOUND_CTRL));
}

// full wave sum, every lane gets the total
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_mov<0xB1>(0.f, v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(0.f, v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(0.f, v);  // row_half_mirror
    v += dpp_mov<0x140>(0.f, v);  // row_mirror  -> every lane of a 16-row holds the row sum
    // across the four rows of 16: use ds_swizzle-free readlane of lanes 0,16,32,48
    float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, dpp_mov<0xB1>(v, v));
    v = fmaxf(v, dpp_mov<0x4E>(v, v));
    v = fmaxf(v, dpp_mov<0x141>(v, v));
    v = fmaxf(v, dpp_mov<0x140>(v, v));
    float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// block-wide sum for blockDim.x = NT (multiple of 64); result valid in every thread.
// `red` is LDS scratch of NT/64 floats.  Deterministic (fixed order).
template <int NT>
__device__ __forceinline__ float block_sum(float v, float *red)
{
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// ----------------------------------------------------------------------------- counter-based RNG for dropout
// 128 bits per group of 8 elements as a function of (seed, group index) alone - the backward pass draws the same bits again.
// r1-r3 used Philox4x32-7: 28 quarter-rate 32-bit multiplications per group, as many issue slots as the APTx arithmetic of the
// eight elements, which held the bf16-storage pointwise kernels (half the bytes per element) at 4.0 TB/s.  r4: the group index is
// hashed (the 'lowbias32' finaliser) into the state of an xorshift32, whose next four states + a Weyl term are the four words.
// Dropout needs independent-looking bits at a fixed rate, not a cryptographic stream.
// r5: the SEED enters as two keys, each a hash of both of its words, around two hash rounds of the group index.  r4 XOR-ed the
// seed's high word into the state AFTER the hash: xorshift is linear over GF(2), so two seeds that differed only there produced
// streams that differed by a fixed XOR stream (the review measured a cross-seed correlation of 5.7e-3 of the decisions); and
// with one keyed round hash(group ^ key) the mask of one seed is the mask of another under a permutation of the groups.  The keys
// are wave-uniform (scalar unit); per group the cost is two more multiplications.
__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
struct SeedKeys {
    uint32_t k0, k1;
};
__device__ __forceinline__ SeedKeys seed_keys(uint64_t seed)
{
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    SeedKeys k;
    k.k0 = lo ^ mix32(hi ^ 0x85ebca6bu);
    k.k1 = mix32(hi + mix32(lo ^ 0xc2b2ae35u));
    return k;
}
__device__ __forceinline__ uint4 drop_words(uint64_t seed, uint64_t group)
{
    const SeedKeys key = seed_keys(seed);
    uint32_t s = mix32(mix32((uint32_t)group ^ key.k0) + key.k1) ^ ((uint32_t)(group >> 32) * 0x9E3779B1u);
    s = s ? s : 0x6d2b79f5u;                                      // xorshift has no zero state
    uint32_t w = s * 0x9E3779B9u;
    uint4 r;
    s ^= s << 13; s ^= s >> 17; s ^= s << 5; w += 0x9E3779B9u; r.x = s + w;
    s ^= s << 13; s ^= s >> 17; s ^= s << 5; w += 0x9E3779B9u; r.y = s + w;
    s ^= s << 13; s ^= s >> 17; s ^= s << 5; w += 0x9E3779B9u; r.z = s + w;
    s ^= s << 13; s ^= s >> 17; s ^= s << 5; w += 0x9E3779B9u; r.w = s + w;
    return r;
}

// Dropout keep-scale for 8 consecutive elements [8*group, 8*group+8): each 32-bit word gives two
// 16-bit uniforms; drop iff u16 < thresh16 where thresh16 = round(p * 65536).
struct Drop8 {
    uint4 r;
    __device__ __forceinline__ Drop8() : r(make_uint4(~0u, ~0u, ~0u, ~0u)) {}   // keeps everything
    __device__ __forceinline__ Drop8(uint64_t seed, uint64_t group)
    {
        r = drop_words(seed, group);
    }
    __device__ __forceinline__ bool keep(int e, uint32_t thresh16) const
    {
        uint32_t w = (e >> 1) == 0 ? r.x : (e >> 1) == 1 ? r.y : (e >> 1) == 2 ? r.z : r.w;
        uint32_t u = (e & 1) ? (w >> 16) : (w & 0xffffu);
        return u >= thresh16;
    }
};

__host__ __device__ inline uint32_t drop_thresh16(float p)
{
    float t = p * 65536.0f + 0.5f;
    return t < 0.f ? 0u : (t > 65535.f ? 65535u : (uint32_t)t);
}
// the keep probability actually realised (so that the scale is exact for the realised p)
__host__ __device__ inline float drop_scale(float p)
{
    return 65536.0f / (65536.0f - (float)drop_thresh16(p));
}

// Index of pixel m in a frame mask: (m / d1) * d2 + m % d3 (the map every masked epilogue takes).  The operands are 64-bit by
// signature; whenever they fit 32 bits - every shape of the configurations does - the two divisions cost a fifth as 32-bit ones.
// (The Winograd output transforms evaluate this per output pixel, 20 times per thread for a 4 x 5 tile.  Measured: their times
// did not move - those kernels wait for the fabric, not for the vector unit - so this is tidiness, not speed.)
__device__ __forceinline__ int64_t mask_index(int64_t m, int64_t d1, int64_t d2, int64_t d3)
{
    if ((((uint64_t)m | (uint64_t)d1 | (uint64_t)d3) >> 32) == 0) {
        const uint32_t mm = (uint32_t)m;
        return (int64_t)(mm / (uint32_t)d1) * d2 + (int64_t)(mm % (uint32_t)d3);
    }
    return (m / d1) * d2 + m % d3;
}

// ----------------------------------------------------------------------------- APTx
// aptx(x) = (1 + tanh(b x)) * g * x = 2*sigmoid(2 b x) * g * x   (no cancellation for any x)
__device__ __forceinline__ float aptx_s(float bx)  // s = 1 + tanh(bx)
{
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * bx));
}

}  // namespace mq
