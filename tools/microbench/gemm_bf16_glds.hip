// Stand-alone bf16 NT GEMM on the structure the bf16-storage conv kernels use (round 2):
//   block 512 threads = 8 waves (4 along M x 2 along N), tile 256 x (64*TN) x 64, v_mfma_f32_32x32x16_bf16,
//   both operands staged by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR staging, no ds_write), 128-byte LDS rows
//   with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7 (applied on the SOURCE address of the DMA and on the
//   fragment reads; the DMA itself writes lane-linear), NSTAGE-deep ring, one barrier per K-tile.
// C[m][n] = sum_k A[m][k] * B[n][k];  A (M, K) bf16, B (N, K) bf16, C (M, N) f32.   K % 64 == 0.
// Build: hipcc --offload-arch=gfx950 -O3 -o gemm_bf16_glds gemm_bf16_glds.hip ; run: ./gemm_bf16_glds M N K [TN]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
constexpr unsigned OOB = 0xFFFFFFFFu;
constexpr int BM = 256, BK = 64, ROWB = 128;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, size_t bytes)
{
    const unsigned n = bytes > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), (short)0, (int)n, 0x00020000);
}

// generic pointer into a __shared__ array -> LDS (address space 3) pointer: the low 32 bits are the LDS offset
__device__ __forceinline__ lds_void *lds_ptr(const void *p) { return reinterpret_cast<lds_void *>(reinterpret_cast<uintptr_t>(p)); }

// (the body lives in a __device__ function: amdgcn builtins written directly in a __global__ template make the HOST
// pass drop the kernel's launch stub without a diagnostic)
template <int TN, int NSTAGE>
__device__ __forceinline__ void gemm_body(const uint16_t *__restrict__ A, const uint16_t *__restrict__ B,
                                          float *__restrict__ C, int M, int N, int K, char *smem)
{
    constexpr int BN = 64 * TN;
    constexpr int STAGE = (BM + BN) * ROWB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware tile order: the N tiles of one M tile are neighbours on one XCD
    const int nNt = (N + BN - 1) / BN, nMt = (M + BM - 1) / BM;
    const int total = nNt * nMt;
    const int lin = blockIdx.x;
    const int q8 = total >> 3, r8 = total & 7, x8 = lin & 7;
    const int tile = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (lin >> 3);
    const int mt = tile / nNt, nt = tile - mt * nNt;
    const int m0 = mt * BM, n0 = nt * BN;

    const __amdgpu_buffer_rsrc_t ra = make_rsrc(A + (size_t)m0 * K, (size_t)(M - m0 < BM ? M - m0 : BM) * K * 2);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(B + (size_t)n0 * K, (size_t)(N - n0 < BN ? N - n0 : BN) * K * 2);
    // DMA slots: piece p = wave + 8 j covers tile rows 8p .. 8p+7; lane -> row 8p + (lane >> 3), physical chunk lane & 7
    const int lrow = lane >> 3;
    const int lc = (lane & 7) ^ ((((wave & 1) << 2) | (lane >> 4)) & 7);      // logical chunk this lane must fetch
    unsigned va[4], vb[TN];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * (wave + 8 * j) + lrow;
        va[j] = (m0 + row < M) ? (unsigned)row * (unsigned)K * 2u + lc * 16u : OOB;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = 8 * (wave + 8 * j) + lrow;
        vb[j] = (n0 + row < N) ? (unsigned)row * (unsigned)K * 2u + lc * 16u : OOB;
    }
    auto issue = [&](int stage, int k0) {
        char *sa = smem + stage * STAGE + wave * 1024;
        char *sb = sa + BM * ROWB;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, lds_ptr(sa + j * 8192), 16, va[j], k0 * 2, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, lds_ptr(sb + j * 8192), 16, vb[j], k0 * 2, 0, 0);
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int qx = lh ^ ((li >> 1) & 7);
    int aoff[4], boff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        aoff[s] = (wm * 64 + li) * ROWB + (((2 * s) ^ qx) << 4);
        boff[s] = BM * ROWB + (wn * 32 * TN + li) * ROWB + (((2 * s) ^ qx) << 4);
    }
    const int nk = K / BK;
    // prologue: NSTAGE - 1 tiles in flight
#pragma unroll
    for (int p = 0; p < NSTAGE - 1; ++p) issue(p, p * BK);
    if (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + TN) : "memory");
    __builtin_amdgcn_s_barrier();

    auto compute = [&](int stage) {
        const char *base = smem + stage * STAGE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 a[2], b[TN];
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const bf16x8 *>(base + aoff[s] + t * 32 * ROWB);
#pragma unroll
            for (int t = 0; t < TN; ++t) b[t] = *reinterpret_cast<const bf16x8 *>(base + boff[s] + t * 32 * ROWB);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
    };
    for (int t = 0; t < nk; ++t) {
        // tile t + NSTAGE - 1 goes into the stage that was read in iteration t - 1 (everyone passed the barrier since)
        issue((t + NSTAGE - 1) % NSTAGE, (t + NSTAGE - 1) * BK);     // past the end of K: offsets beyond the descriptor -> zeros
        compute(t % NSTAGE);
        if (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + TN) : "memory");      // tile t + 1 landed, t + 2 stays in flight
        __builtin_amdgcn_s_barrier();
    }

    // epilogue: f32 C
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + wn * 32 * TN + tn * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M && n < N) C[(size_t)m * N + n] = acc[tm][tn][r];
            }
        }
}

template <int TN, int NSTAGE>
__global__ __launch_bounds__(512, 2) void gemm_kernel(const uint16_t *__restrict__ A, const uint16_t *__restrict__ B,
                                                      float *__restrict__ C, int M, int N, int K)
{
    __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * (BM + 64 * TN) * ROWB];
    gemm_body<TN, NSTAGE>(A, B, C, M, N, K, smem);
}

static uint16_t f2bf(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h)
{
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int TN, int NSTAGE>
static double run(const uint16_t *dA, const uint16_t *dB, float *dC, int M, int N, int K, int iters)
{
    const int BN = 64 * TN;
    const int grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_kernel<TN, NSTAGE>), dim3(grid), dim3(512), 0, 0, dA, dB, dC, M, N, K);
    CK(hipDeviceSynchronize());
    hipEvent_t s, e;
    CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    CK(hipEventRecord(s));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_kernel<TN, NSTAGE>), dim3(grid), dim3(512), 0, 0, dA, dB, dC, M, N, K);
    CK(hipEventRecord(e));
    CK(hipEventSynchronize(e));
    float ms;
    CK(hipEventElapsedTime(&ms, s, e));
    return ms / iters;
}

int main(int argc, char **argv)
{
    int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    int TN = argc > 4 ? atoi(argv[4]) : 4, NST = argc > 5 ? atoi(argv[5]) : 2;
    if (K % 64) { printf("K %% 64 != 0\n"); return 1; }
    std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
    uint32_t st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto &v : hA) v = f2bf(rnd());
    for (auto &v : hB) v = f2bf(rnd());
    uint16_t *dA, *dB;
    float *dC;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xFF, (size_t)M * N * 4));
    double ms = 0;
    const int it = 20;
#define RUN(T, S) if (TN == T && NST == S) ms = run<T, S>(dA, dB, dC, M, N, K, it)
    RUN(4, 2); RUN(3, 2); RUN(2, 2); RUN(1, 2); RUN(2, 3); RUN(1, 3);
    if (ms == 0) { printf("unsupported TN/NSTAGE\n"); return 1; }
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    // check a sample of rows fully (incl. the last rows / cols: ragged tiles)
    double maxerr = 0, maxref = 0;
    int checked = 0;
    for (int mi = 0; mi < 24; ++mi) {
        const int m = mi < 8 ? mi * 37 % M : (mi < 16 ? M - 1 - (mi - 8) * 29 % M : (int)((uint64_t)mi * 2654435761u % M));
        for (int n = 0; n < N; n += (N > 512 ? 7 : 1)) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
            const double err = fabs(ref - hC[(size_t)m * N + n]);
            if (err > maxerr) maxerr = err;
            if (fabs(ref) > maxref) maxref = fabs(ref);
            ++checked;
        }
    }
    const double tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12;
    printf("M=%d N=%d K=%d TN=%d NSTAGE=%d: %.4f ms  %.1f TFLOP/s   check: %d entries, max err %.3e (max |ref| %.3e) %s\n", M, N, K, TN,
           NST, ms, tf, checked, maxerr, maxref, maxerr <= 2e-3 * (maxref + 1) ? "OK" : "MISMATCH");
    return maxerr <= 2e-3 * (maxref + 1) ? 0 : 2;
}
