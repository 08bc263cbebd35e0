/* A consumer of libmqgan_hip.so written in plain C99: no Python, no PyTorch, no C++ - the drop-in boundary of SURVEY.md 8b as a
 * program.  It allocates device memory with the HIP runtime's C API and calls the library through include/mqgan_hip.h only:
 *
 *   1. the codebook search (quantizer path): mq_l2_codebook_norms + mq_l2_argmin, and the screened search
 *      mq_l2_screen_prepare + mq_l2_argmin_screened, checked BIT FOR BIT (indices and best distances) against the definition
 *      restated here with fmaf chains (oracle/vq_ref.c is the test suite's copy of the same definition; this file does not use it);
 *   2. one refiner-type convolution through the PLANNED calls (mq_weight_prep -> mq_conv_plan -> mq_conv_fprop_planned: the library
 *      picks the Winograd form itself), checked against a double-precision loop at 1e-5 of the output's rms;
 *   3. the data-parallel exchange with one rank (mq_comm_unique_id -> mq_comm_init -> mq_allreduce_bucket on a side stream ->
 *      mq_comm_destroy): a sum over one rank leaves the buffer as it was.
 *
 * Build / run: make -C examples/c_consumer && examples/c_consumer/c_consumer     (tests/test_gpu_c_consumer.py does both) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "mqgan_hip.h"

#define HIP(call)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d: %s: %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)
#define MQ(call)                                                                          \
    do {                                                                                  \
        if ((call) != 0) {                                                                \
            fprintf(stderr, "%s:%d: %s: %s\n", __FILE__, __LINE__, #call, mq_last_error()); \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float rnd(void) /* uniform in [-1, 1), 24 random bits */
{
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((rng_state >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

static int search(hipStream_t st)
{
    const int n = 1500, k = 384, d = 64;
    float *x = malloc(sizeof(float) * n * d), *c = malloc(sizeof(float) * k * d), *cn = malloc(sizeof(float) * k);
    float *bd = malloc(sizeof(float) * n), *bd2 = malloc(sizeof(float) * n);
    int32_t *idx = malloc(sizeof(int32_t) * n), *idx2 = malloc(sizeof(int32_t) * n);
    float *dx, *dc, *dcn, *dbd;
    int32_t *didx;
    void *ws, *blob, *ws2;
    int i, j, e, bad = 0;
    for (i = 0; i < n * d; ++i) x[i] = rnd() * 2.0f;
    for (i = 0; i < k * d; ++i) c[i] = rnd() * 2.0f;
    memcpy(c + 7 * d, c + 3 * d, sizeof(float) * d); /* a tie: the lowest index wins */
    HIP(hipMalloc((void **)&dx, sizeof(float) * n * d));
    HIP(hipMalloc((void **)&dc, sizeof(float) * k * d));
    HIP(hipMalloc((void **)&dcn, sizeof(float) * k));
    HIP(hipMalloc((void **)&dbd, sizeof(float) * n));
    HIP(hipMalloc((void **)&didx, sizeof(int32_t) * n));
    HIP(hipMalloc(&ws, mq_l2_argmin_workspace(n) + 256));
    HIP(hipMemcpyAsync(dx, x, sizeof(float) * n * d, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(dc, c, sizeof(float) * k * d, hipMemcpyHostToDevice, st));
    MQ(mq_l2_codebook_norms(dc, k, d, dcn, st));
    MQ(mq_l2_argmin(dx, dc, dcn, n, k, d, didx, dbd, ws, st));
    HIP(hipMemcpyAsync(cn, dcn, sizeof(float) * k, hipMemcpyDeviceToHost, st));
    HIP(hipMemcpyAsync(idx, didx, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    HIP(hipMemcpyAsync(bd, dbd, sizeof(float) * n, hipMemcpyDeviceToHost, st));
    /* the screened search: bf16 scores + exact re-rank, the same answers */
    if (!mq_l2_screen_supported(n, k, d)) {
        fprintf(stderr, "screened search refuses %d x %d x %d\n", n, k, d);
        return 1;
    }
    HIP(hipMalloc(&blob, mq_l2_screen_codebook_bytes(k, d)));
    HIP(hipMalloc(&ws2, mq_l2_screen_workspace(n, k, d)));
    MQ(mq_l2_screen_prepare(dc, k, d, blob, st));
    MQ(mq_l2_argmin_screened(dx, dc, blob, n, k, d, didx, dbd, ws2, mq_l2_screen_workspace(n, k, d), st));
    HIP(hipMemcpyAsync(idx2, didx, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    HIP(hipMemcpyAsync(bd2, dbd, sizeof(float) * n, hipMemcpyDeviceToHost, st));
    HIP(hipStreamSynchronize(st));
    for (j = 0; j < k; ++j) { /* cnorm[k] = fma chain of c^2 */
        float s = 0.f;
        for (e = 0; e < d; ++e) s = fmaf(c[j * d + e], c[j * d + e], s);
        if (s != cn[j]) ++bad;
    }
    for (i = 0; i < n; ++i) {
        float best = INFINITY;
        int bi = 0;
        for (j = 0; j < k; ++j) {
            float dot = 0.f, dist;
            for (e = 0; e < d; ++e) dot = fmaf(x[i * d + e], c[j * d + e], dot);
            dist = fmaf(-2.0f, dot, cn[j]);
            if (dist < best) { best = dist; bi = j; }
        }
        if (idx[i] != bi || bd[i] != best || idx2[i] != bi || bd2[i] != best) ++bad;
    }
    printf("codebook search %d x %d x %d: %s (f32 kernel and screened search against the fmaf-chain definition, bit for bit)\n", n, k, d,
           bad ? "MISMATCH" : "identical");
    hipFree(dx); hipFree(dc); hipFree(dcn); hipFree(dbd); hipFree(didx); hipFree(ws); hipFree(blob); hipFree(ws2);
    free(x); free(c); free(cn); free(bd); free(bd2); free(idx); free(idx2);
    return bad != 0;
}

static int conv(hipStream_t st)
{
    const int N = 2, H = 16, W = 20, C = 64, K = 64, R = 3, S = 3;
    const size_t nx = (size_t)N * H * W * C, nw = (size_t)K * C * R * S, ny = (size_t)N * H * W * K;
    float *x = malloc(sizeof(float) * nx), *v = malloc(sizeof(float) * nw), *b = malloc(sizeof(float) * K), *y = malloc(sizeof(float) * ny);
    float *dx, *dv, *dwk, *dwc, *db, *dy;
    void *ws = NULL;
    mqConvDesc d;
    mqConvPlan p;
    size_t i;
    int n, h, w, k, c, r, s;
    double err2 = 0.0, ref2 = 0.0;
    for (i = 0; i < nx; ++i) x[i] = rnd();
    for (i = 0; i < nw; ++i) v[i] = rnd() * 0.1f; /* the reference's layout: (K, C, R * S) */
    for (k = 0; k < K; ++k) b[k] = rnd();
    memset(&d, 0, sizeof(d));
    d.N = N; d.H = H; d.W = W; d.C = C; d.K = K; d.R = R; d.S = S; d.P = H; d.Q = W;
    d.stride_h = d.stride_w = 1; d.pad_h = d.pad_w = 1; d.dil_h = d.dil_w = 1; d.groups = 1; d.precision = MQ_PRECISION_F32;
    HIP(hipMalloc((void **)&dx, sizeof(float) * nx));
    HIP(hipMalloc((void **)&dv, sizeof(float) * nw));
    HIP(hipMalloc((void **)&dwk, sizeof(float) * nw));
    HIP(hipMalloc((void **)&dwc, sizeof(float) * nw));
    HIP(hipMalloc((void **)&db, sizeof(float) * K));
    HIP(hipMalloc((void **)&dy, sizeof(float) * ny));
    HIP(hipMemcpyAsync(dx, x, sizeof(float) * nx, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(dv, v, sizeof(float) * nw, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(db, b, sizeof(float) * K, hipMemcpyHostToDevice, st));
    MQ(mq_weight_prep(0, dv, NULL, K, C, R * S, dwk, dwc, NULL, st));
    MQ(mq_conv_plan(&d, 0, NULL, &p));
    if (p.ws_fprop) HIP(hipMalloc(&ws, p.ws_fprop));
    MQ(mq_conv_fprop_planned(&p, dx, dwk, db, 0, 0.0f, NULL, 1, 0, 1, dy, NULL, ws, p.ws_fprop, st));
    HIP(hipMemcpyAsync(y, dy, sizeof(float) * ny, hipMemcpyDeviceToHost, st));
    HIP(hipStreamSynchronize(st));
    for (n = 0; n < N; ++n)
        for (h = 0; h < H; ++h)
            for (w = 0; w < W; ++w)
                for (k = 0; k < K; ++k) {
                    double acc = b[k];
                    for (r = 0; r < R; ++r)
                        for (s = 0; s < S; ++s) {
                            const int ih = h + r - 1, iw = w + s - 1;
                            if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
                            for (c = 0; c < C; ++c)
                                acc += (double)x[((size_t)(n * H + ih) * W + iw) * C + c] * (double)v[((size_t)k * C + c) * R * S + r * S + s];
                        }
                    {
                        const double g = y[((size_t)(n * H + h) * W + w) * K + k];
                        err2 += (g - acc) * (g - acc);
                        ref2 += acc * acc;
                    }
                }
    printf("planned convolution %d x %d x %d, %d -> %d, 3 x 3: form \"%s\", relative L2 error %.2e\n", N, H, W, C, K, mq_form_name(p.form),
           sqrt(err2 / ref2));
    hipFree(dx); hipFree(dv); hipFree(dwk); hipFree(dwc); hipFree(db); hipFree(dy);
    if (ws) hipFree(ws);
    free(x); free(v); free(b); free(y);
    return !(sqrt(err2 / ref2) < 1e-5);
}

static int exchange(hipStream_t st)
{
    const size_t n = 1u << 16;
    unsigned char id[MQ_COMM_ID_BYTES];
    mqComm *comm = NULL;
    float *h = malloc(sizeof(float) * n), *g = malloc(sizeof(float) * n), *dbuf;
    hipStream_t side;
    hipEvent_t ready, done;
    size_t i;
    int bad = 0;
    for (i = 0; i < n; ++i) h[i] = rnd();
    HIP(hipMalloc((void **)&dbuf, sizeof(float) * n));
    HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    MQ(mq_comm_unique_id(id));                       /* rank 0; the bytes travel to the other ranks over any host channel */
    MQ(mq_comm_init(&comm, 1, 0, id));
    HIP(hipMemcpyAsync(dbuf, h, sizeof(float) * n, hipMemcpyHostToDevice, st));   /* "backward" wrote the bucket on the compute stream */
    HIP(hipEventRecord(ready, st));
    HIP(hipStreamWaitEvent(side, ready, 0));
    MQ(mq_allreduce_bucket(comm, dbuf, n, side));    /* the exchange runs on the side stream */
    HIP(hipEventRecord(done, side));
    HIP(hipStreamWaitEvent(st, done, 0));            /* clip / Adam would follow on the compute stream */
    HIP(hipMemcpyAsync(g, dbuf, sizeof(float) * n, hipMemcpyDeviceToHost, st));
    HIP(hipStreamSynchronize(st));
    for (i = 0; i < n; ++i) bad += g[i] != h[i];
    MQ(mq_comm_destroy(comm));
    printf("gradient-bucket all-reduce over RCCL, one rank, %zu floats on a side stream: %s\n", n, bad ? "CHANGED" : "unchanged (sum over one rank)");
    hipEventDestroy(ready); hipEventDestroy(done); hipStreamDestroy(side); hipFree(dbuf);
    free(h); free(g);
    return bad != 0;
}

int main(void)
{
    hipStream_t st;
    if (mq_abi_version() != 11) {
        fprintf(stderr, "libmqgan_hip.so has ABI %d, this program was written against 11\n", mq_abi_version());
        return 1;
    }
    HIP(hipSetDevice(0));
    HIP(hipStreamCreate(&st));
    if (search(st) || conv(st) || exchange(st)) {
        fprintf(stderr, "c_consumer FAILED\n");
        return 1;
    }
    HIP(hipStreamDestroy(st));
    printf("c_consumer OK\n");
    return 0;
}
