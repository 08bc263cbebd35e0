#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// MODE: 0 none; 1 s_nop 0 after each mfma; 2 s_nop 3 after each; 3 v_nop after each; 4 s_nop 7 after each group of 4;
// 5 4x v_nop after each group of 4; 6 one v_nop after group; 7: s_nop 15 after group; 8: 2 v_nop after each mfma
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
                if (MODE == 1) asm volatile("s_nop 0");
                if (MODE == 2) asm volatile("s_nop 3");
                if (MODE == 3) asm volatile("v_nop");
                if (MODE == 8) asm volatile("v_nop\n v_nop");
            }
            if (MODE == 4) asm volatile("s_nop 7");
            if (MODE == 5) asm volatile("v_nop\n v_nop\n v_nop\n v_nop");
            if (MODE == 6) asm volatile("v_nop");
            if (MODE == 7) asm volatile("s_nop 15");
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(int blocks_per_cu)
{
    float *out;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, out, 10, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms;
    (void)hipEventElapsedTime(&ms, s, e);
    const double flop = (double)grid * 4 * iters * 8 * 4 * 32 * 32 * 2 * 2;
    printf("mode %d blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", MODE, blocks_per_cu, ms, flop / ms / 1e9);
    (void)hipFree(out);
}
int main()
{
    for (int b : {4, 3, 2}) {
        run<0>(b); run<1>(b); run<2>(b); run<3>(b); run<8>(b); run<4>(b); run<5>(b); run<6>(b); run<7>(b);
    }
    return 0;
}
