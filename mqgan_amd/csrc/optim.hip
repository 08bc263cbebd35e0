// clip_grad_norm_ + Adam on flat fp32 arenas (train.py:401-407, 485-492).
// One launch for the squared norm (two-stage, deterministic) and one for the update; the clip
// coefficient never leaves the device.
#include "common.h"

namespace mq {

__global__ void sumsq_partial_kernel(const float *__restrict__ g, int64_t n, float *__restrict__ part)
{
    __shared__ float red[4];
    float s = 0.f;
    const int64_t n4 = n >> 2;
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = g4[i];
        s = fmaf(v.x, v.x, s);
        s = fmaf(v.y, v.y, s);
        s = fmaf(v.z, v.z, s);
        s = fmaf(v.w, v.w, s);
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s = fmaf(g[i], g[i], s);
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void sumsq_final_kernel(const float *__restrict__ part, int nb, float *__restrict__ out)
{
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) out[0] = s;
}

// torch.optim.Adam (single-tensor path) op order:
//   g *= clip_coef ; m.lerp_(g, 1-b1) ; v = v*b2 + (1-b2) g*g ; denom = sqrt(v)/sqrt(bc2) + eps ; p -= (lr/bc1) * m/denom
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, int64_t n,
                            float step_size, float beta1, float beta2, float eps, float bc2_sqrt, const float *__restrict__ sumsq,
                            float max_norm, float grad_scale)
{
    // grad_scale turns all-reduced SUMS into means (1 / world size) without an extra pass
    float coef = grad_scale;
    if (max_norm > 0.f && sumsq) {
        const float tn = sqrtf(sumsq[0]) * grad_scale;
        coef = fminf(max_norm / (tn + 1e-6f), 1.0f) * grad_scale;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);
        const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// Gradient arena fill.  Autograd leaves every parameter's gradient in a tensor of its own; one launch copies them all
// into the flat arena the clip / Adam kernels and the bucketed all-reduce read (instead of one small add per parameter
// into pre-zeroed views).  Table rows: {source address (0: no gradient this step -> zeros), arena offset, elements,
// first block}; a block covers MG_CHUNK elements of one tensor and finds its row by bisection.
constexpr int MG_CHUNK = 2048;
__global__ __launch_bounds__(256) void multi_gather_kernel(const int64_t *__restrict__ tab, int nt, float *__restrict__ dst)
{
    const int64_t b = blockIdx.x;
    int lo = 0, hi = nt - 1;
    while (lo < hi) {                                   // last row whose first block is <= b
        const int mid = (lo + hi + 1) >> 1;
        if (tab[4 * mid + 3] <= b) lo = mid; else hi = mid - 1;
    }
    const float *src = reinterpret_cast<const float *>(tab[4 * lo]);
    float *d = dst + tab[4 * lo + 1];
    const int64_t n = tab[4 * lo + 2];
    const int64_t base = (b - tab[4 * lo + 3]) * MG_CHUNK;
    const bool vec = (reinterpret_cast<uintptr_t>(src) & 15) == 0;      // arena slices are 16-byte aligned by construction
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t e = base + h * 1024 + threadIdx.x * 4;
        if (e + 4 <= n && vec) {
            const float4 v = src ? *reinterpret_cast<const float4 *>(src + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(d + e) = v;
        } else {
            for (int k = 0; k < 4; ++k)
                if (e + k < n) d[e + k] = src ? src[e + k] : 0.f;
        }
    }
}

// The same fill with the table passed BY VALUE in the kernel arguments: the launch captures the rows, so the caller's host
// table may be reused the moment the call returns (no staging buffer whose reuse would have to be ordered against the
// stream).  A launch carries up to MG_ARG_ROWS rows; the block -> row search runs on scalar loads from the kernarg segment.
constexpr int MG_ARG_ROWS = 120;                        // 120 rows x 32 B + dst + count < the 4 KiB kernarg limit
struct MgArgTable {
    int64_t row[MG_ARG_ROWS][4];
};
__global__ __launch_bounds__(256) void multi_gather_args_kernel(const MgArgTable tab, int nt, float *__restrict__ dst)
{
    const int64_t b = blockIdx.x;
    int lo = 0, hi = nt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab.row[mid][3] <= b) lo = mid; else hi = mid - 1;
    }
    const float *src = reinterpret_cast<const float *>(tab.row[lo][0]);
    float *d = dst + tab.row[lo][1];
    const int64_t n = tab.row[lo][2];
    const int64_t base = (b - tab.row[lo][3]) * MG_CHUNK;
    const bool vec = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t e = base + h * 1024 + threadIdx.x * 4;
        if (e + 4 <= n && vec) {
            const float4 v = src ? *reinterpret_cast<const float4 *>(src + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(d + e) = v;
        } else {
            for (int k = 0; k < 4; ++k)
                if (e + k < n) d[e + k] = src ? src[e + k] : 0.f;
        }
    }
}

}  // namespace mq

using namespace mq;

extern "C" int mq_multi_gather_args(const int64_t *table_host, int n_tensors, float *dst, mqStream stream)
{
    MQ_REQUIRE(table_host && dst && n_tensors > 0, "mq_multi_gather_args: bad arguments");
    MQ_REQUIRE((uintptr_t)dst % 16 == 0, "mq_multi_gather_args: dst must be 16-byte aligned");
    for (int i = 0; i < n_tensors; ++i) {
        const int64_t *r = table_host + 4 * i;
        MQ_REQUIRE(r[1] >= 0 && (r[1] & 3) == 0 && r[2] > 0, "mq_multi_gather_args: row with a negative / unaligned offset or no elements");
    }
    for (int first = 0; first < n_tensors; first += MG_ARG_ROWS) {
        const int nt = n_tensors - first < MG_ARG_ROWS ? n_tensors - first : MG_ARG_ROWS;
        MgArgTable t;
        int64_t blocks = 0;
        for (int i = 0; i < nt; ++i) {
            const int64_t *r = table_host + 4 * (first + i);
            t.row[i][0] = r[0];
            t.row[i][1] = r[1];
            t.row[i][2] = r[2];
            t.row[i][3] = blocks;                      // first block of the row inside THIS launch
            blocks += (r[2] + MG_CHUNK - 1) / MG_CHUNK;
        }
        MQ_REQUIRE(blocks < INT32_MAX, "mq_multi_gather_args: too many elements for one launch");
        hipLaunchKernelGGL(multi_gather_args_kernel, dim3((unsigned)blocks), dim3(256), 0, S(stream), t, nt, dst);
    }
    MQ_LAUNCH_CHECK("mq_multi_gather_args");
    return 0;
}

extern "C" int mq_multi_gather(const int64_t *table_dev, int n_tensors, int64_t total_blocks, float *dst, mqStream stream)
{
    MQ_REQUIRE(table_dev && dst && n_tensors > 0 && total_blocks > 0 && total_blocks < INT32_MAX, "mq_multi_gather: bad arguments");
    MQ_REQUIRE((uintptr_t)dst % 16 == 0, "mq_multi_gather: dst must be 16-byte aligned");
    hipLaunchKernelGGL(multi_gather_kernel, dim3((unsigned)total_blocks), dim3(256), 0, S(stream), table_dev, n_tensors, dst);
    MQ_LAUNCH_CHECK("mq_multi_gather");
    return 0;
}

extern "C" int mq_sumsq(const float *g, int64_t n, float *out, float *workspace, mqStream stream)
{
    MQ_REQUIRE(g && out && workspace && n > 0, "mq_sumsq: bad arguments");
    MQ_REQUIRE((uintptr_t)g % 16 == 0, "mq_sumsq: g must be 16-byte aligned");
    int nb = ew_grid(n / 4 + 1, 256);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, S(stream), g, n, workspace);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, S(stream), workspace, nb, out);
    MQ_LAUNCH_CHECK("mq_sumsq");
    return 0;
}

extern "C" int mq_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr, double beta1,
                            double beta2, double eps, int step_host, const float *sumsq, float max_norm, float grad_scale,
                            mqStream stream)
{
    MQ_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step_host >= 1, "mq_adam_step: bad arguments");
    const double bc1 = 1.0 - pow(beta1, (double)step_host);
    const double bc2 = 1.0 - pow(beta2, (double)step_host);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), param, grad, exp_avg, exp_avg_sq, n, step_size, (float)beta1,
                       (float)beta2, (float)eps, bc2_sqrt, sumsq, max_norm, grad_scale);
    MQ_LAUNCH_CHECK("mq_adam_step");
    return 0;
}
