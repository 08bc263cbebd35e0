#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int VALU>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x = a + threadIdx.x, y = b;
    int dummy = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < VALU; ++v) dummy = dummy * 3 + u + v;   // v_mad-ish integer VALU filler
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + dummy;
}
template <int NACC, int VALU>
void run(const char *name, int blocks_per_cu)
{
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL((k<NACC, VALU>), dim3(grid), dim3(256), 0, 0, out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL((k<NACC, VALU>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    const double flop = (double)grid * 4 * iters * 8 * NACC * 32 * 32 * 2 * 2;
    printf("%s blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, flop / ms / 1e9);
    hipFree(out);
}
int main()
{
    run<4, 0>("acc4 valu0", 1);
    run<4, 0>("acc4 valu0", 2);
    run<4, 0>("acc4 valu0", 4);
    run<4, 1>("acc4 valu1/4mfma", 4);
    run<4, 4>("acc4 valu4/4mfma", 4);
    run<4, 8>("acc4 valu8/4mfma", 4);
    run<4, 16>("acc4 valu16/4mfma", 4);
    run<4, 8>("acc4 valu8/4mfma", 1);
    return 0;
}
