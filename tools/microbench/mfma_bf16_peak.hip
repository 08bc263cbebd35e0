// Bare v_mfma_f32_32x32x16_bf16 loop: sustained rate (and so the clock the chip holds under matrix load).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int VALU>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(a + threadIdx.x + j); y[j] = (__bf16)(a * j); }
    int dummy = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < VALU; ++v) dummy = dummy * 3 + u + v;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + dummy;
}
template <int VALU>
void run(int blocks_per_cu, int iters)
{
    float *out;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int grid = 256 * blocks_per_cu;
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    hipLaunchKernelGGL((k<VALU>), dim3(grid), dim3(256), 0, 0, out, 10, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    hipLaunchKernelGGL((k<VALU>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms;
    (void)hipEventElapsedTime(&ms, s, e);
    const double flop = (double)grid * 4 * iters * 8 * 4 * 32 * 32 * 16 * 2;
    printf("bf16 mfma, %d VALU per 4 MFMA, blocks/CU=%d, %.1f ms: %.0f TFLOP/s (%.2f GHz-equivalent of the 2.5 PF @ 2.4 GHz peak)\n", VALU,
           blocks_per_cu, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500.0 * 2.4);
    (void)hipFree(out);
}
int main()
{
    run<0>(1, 4000); run<0>(2, 4000); run<0>(3, 4000); run<0>(2, 40000);
    run<4>(2, 4000); run<8>(2, 4000);
    return 0;
}
