// Algorithm choice and execution of one convolution layer behind the C ABI (include/mqgan_hip.h: mq_conv_plan,
// mq_conv_fprop_planned, mq_conv_backward_planned).
//
// Which layers of the reference take which form (reference preencoder.py:86-130: the refiner's 3x3 / stride-1 blocks;
// discriminators.py:196-242, 292-311: the patch discriminator's 5x5 / stride-2 and 3x3 / stride-(2, 1) layers, the multi-bin
// discriminator's (3 x k) layers; attentions.py:95-130: the encoder / decoder 1-D convolutions) is decided HERE, once: a caller
// that binds the header gets the Winograd / polyphase / nested forms by the same rule the Python package applies - it asks for a
// plan, allocates the two buffers the plan sizes (workspace; optionally the forward -> backward "keep" buffer) and calls the
// planned entry points.  Every form is the batch-grouped GEMM kernels of igemm.hip between the HBM-bound transforms of
// winogradn.hip / wino1d.hip / winow.hip; the orchestration (which transform, which GEMM descriptor, which buffer) lives in this
// file only.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace {

using mq::fail;

mqLaunchHook g_hook = nullptr;
mqPassHook g_pass = nullptr;

inline size_t al(size_t n) { return (n + 255) & ~(size_t)255; }
inline int cdivi(int a, int b) { return (a + b - 1) / b; }

int env_int(const char *name, int dflt, int *was_set = nullptr)
{
    const char *v = getenv(name);
    if (was_set) *was_set = v != nullptr;
    return v ? atoi(v) : dflt;
}

bool f32_result(const mqConvDesc &d) { return d.precision == MQ_PRECISION_F32 || d.precision == MQ_PRECISION_F32X3; }

double gemm_flops(const mqConvDesc &g) { return 2.0 * g.N * g.P * g.Q * (double)g.K * g.C * g.R * g.S; }

mqConvDesc make_desc(int N, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, int groups, int P, int Q, int precision)
{
    mqConvDesc g;
    g.N = N; g.H = H; g.W = W; g.C = C; g.K = K; g.R = R; g.S = S; g.P = P; g.Q = Q;
    g.stride_h = sh; g.stride_w = sw; g.pad_h = ph; g.pad_w = pw; g.dil_h = 1; g.dil_w = 1; g.groups = groups; g.precision = precision;
    return g;
}

// ---------------------------------------------------------------------------------------------------- eligibility
// Output tile of F(mh x mw, 3x3): the one with the fewest products among 3x3 (25 per 9 outputs), 4x4 (36 per 16) and 4x5 (42 per
// 20) for this image; ties go to the smaller tile (smaller error).  cfg.wino_tile 3 / 4 / 45 force one.
void wino2d_tile(int H, int W, int forced, int *mh, int *mw)
{
    if (forced == 3) { *mh = 3; *mw = 3; return; }
    if (forced == 4) { *mh = 4; *mw = 4; return; }
    if (forced == 45) { *mh = 4; *mw = 5; return; }
    if (forced == 6) { *mh = 6; *mw = 6; return; }
    const int th[3] = {3, 4, 4}, tw[3] = {3, 4, 5};
    long best = -1;
    for (int i = 0; i < 3; ++i) {
        const long cost = (long)cdivi(H, th[i]) * cdivi(W, tw[i]) * (th[i] + 2) * (tw[i] + 2);
        if (best < 0 || cost < best) { best = cost; *mh = th[i]; *mw = tw[i]; }
    }
}

// 3x3, stride 1, pad 1, one group on an f32-result engine.  Worth it when the GEMM work saved (proportional to C K) outweighs the
// transforms (proportional to C + K): C K / (C + K) >= 45 for F(3x3) (192 -> 64 pays, 64 -> 64 does not), >= 32 for the 4-wide
// tiles; an explicit minimum (MQGAN_WINO_MINC) is a plain bound on min(C, K).  K <= 1024: the fused dy pass.
bool wino2d_ok(const mqConvDesc &d, const mqPlanCfg &c, int *mh, int *mw)
{
    if (!(c.wino_minc > 0 && d.R == 3 && d.S == 3 && d.stride_h == 1 && d.stride_w == 1 && d.pad_h == 1 && d.pad_w == 1 && d.dil_h == 1 &&
          d.dil_w == 1 && d.groups == 1 && d.P == d.H && d.Q == d.W && f32_result(d) && d.C % 16 == 0 && d.K % 16 == 0 && d.K <= 1024))
        return false;
    wino2d_tile(d.H, d.W, c.wino_tile, mh, mw);
    if (!c.wino_minc_explicit) {
        const long need = (*mh == 3 ? 45 : 32);
        return std::min(d.C, d.K) >= 64 && (long)d.C * d.K >= need * ((long)d.C + d.K);
    }
    return std::min(d.C, d.K) >= c.wino_minc;
}

// 5x5, stride 2, pad 2, one group, even image: a 3x3 / stride-1 convolution over the four polyphase components of the input
bool s2d_ok(const mqConvDesc &d, const mqPlanCfg &c, int *mh, int *mw)
{
    if (!(c.s2d_minc > 0 && d.R == 5 && d.S == 5 && d.stride_h == 2 && d.stride_w == 2 && d.pad_h == 2 && d.pad_w == 2 && d.dil_h == 1 &&
          d.dil_w == 1 && d.groups == 1 && d.H % 2 == 0 && d.W % 2 == 0 && d.P == d.H / 2 && d.Q == d.W / 2 && f32_result(d) &&
          d.C % 16 == 0 && d.K % 16 == 0 && d.K <= 1024 && std::min(d.C, d.K) >= c.s2d_minc))
        return false;
    wino2d_tile(d.P, d.Q, c.wino_tile, mh, mw);
    return true;
}

// 3 taps, stride 1, pad 1 along the rows (any taps, stride 1 or 2 along the columns), batch groups: F(m, 3) along the rows
bool wino1d_ok(const mqConvDesc &d, const mqPlanCfg &c)
{
    return c.wino1d_minc > 0 && d.R == 3 && d.stride_h == 1 && d.pad_h == 1 && d.dil_h == 1 && d.dil_w == 1 && d.P == d.H && d.H >= 2 &&
           (d.stride_w == 1 || d.stride_w == 2) && f32_result(d) && d.C % 16 == 0 && d.K % 16 == 0 && d.K <= 1024 &&
           std::min(d.C, d.K) >= c.wino1d_minc && d.groups > 0 && d.N % d.groups == 0;
}

// 3, 5 or 7 stride-1 taps along the columns (any taps / stride along the rows, any left padding, batch groups)
bool winow_ok(const mqConvDesc &d, const mqPlanCfg &c)
{
    return c.winow_minc > 0 && (d.S == 3 || d.S == 5 || d.S == 7) && d.stride_w == 1 && d.dil_w == 1 && d.dil_h == 1 && d.Q >= 1 &&
           f32_result(d) && d.C % 16 == 0 && d.K % 16 == 0 && std::min(d.C, d.K) >= c.winow_minc && d.groups > 0 && d.N % d.groups == 0;
}

// the positions of F(m, S) along the columns as (m + S - 1) x groups batch groups of (H x tiles) images with (R x 1) filters
mqConvDesc winow_desc(const mqConvDesc &d)
{
    const int m = mq_winow_tile(d.S), tw = cdivi(d.Q, m), np = m + d.S - 1;
    return make_desc(np * d.N, d.H, tw, d.C, d.K, d.R, 1, d.stride_h, 1, d.pad_h, 0, np * d.groups, d.P, tw, d.precision);
}

// the positions of F(mh x mw, 3x3) over an (N, H, W) grid as that many groups of one-row "images" (one per tile row: small images
// keep the GEMM kernels' 32-bit image-relative offsets far from their limits) with batch-grouped 1x1 filters
mqConvDesc wino2d_desc(int N, int H, int W, int cin, int cout, int mh, int mw, int precision)
{
    const int np = (mh + 2) * (mw + 2), th = cdivi(H, mh), tw = cdivi(W, mw);
    return make_desc(np * N * th, 1, tw, cin, cout, 1, 1, 1, 1, 0, 0, np, 1, tw, precision);
}

// ---------------------------------------------------------------------------------------------------- buffers
// Carves one caller-provided buffer.  With base == NULL it only measures (peak): mq_conv_plan sizes the workspaces by running the
// SAME code paths dry, so the sizes it reports and the bytes the planned calls touch cannot drift apart.
struct Carver {
    char *base;
    size_t cap, used = 0, peak = 0;
    Carver(void *p, size_t capacity) : base(static_cast<char *>(p)), cap(capacity) {}
    bool over() const { return peak > cap; }      // checked before every launch (MQ_TRY): nothing runs on a pointer past the end
    void *take_bytes(size_t bytes)
    {
        char *p = base ? base + used : nullptr;
        used += al(bytes);
        peak = std::max(peak, used);
        return p;
    }
    float *take(size_t floats) { return static_cast<float *>(take_bytes(floats * sizeof(float))); }
    size_t mark() const { return used; }
    void rewind(size_t m) { used = m; }
};

struct Sizes {          // element counts of one layer's intermediates
    size_t V = 0, U = 0, Uc = 0, M = 0;      // forward
    size_t V2 = 0, U2 = 0, Uc2 = 0, M2 = 0;  // nested column form of the row form's GEMM
};

Sizes sizes_of(const mqConvPlan &p)
{
    Sizes s;
    const mqConvDesc &d = p.desc;
    switch (p.form) {
    case MQ_FORM_WINO2D:
        s.V = (size_t)p.npos * p.tiles * d.C; s.U = (size_t)p.npos * d.K * d.C; s.M = (size_t)p.npos * p.tiles * d.K;
        break;
    case MQ_FORM_S2D:
        s.V = (size_t)p.npos * p.tiles * 4 * d.C; s.U = (size_t)p.npos * d.K * 4 * d.C; s.M = (size_t)p.npos * p.tiles * d.K;
        break;
    case MQ_FORM_WINOW: {
        const mqConvDesc &g = p.gemm;                  // (np N, H, tw, C) -> (np N, P, tw, K), filters (np G K, R, C)
        s.V = (size_t)g.N * g.H * g.W * g.C; s.U = (size_t)g.groups * g.K * g.R * g.C; s.Uc = s.U; s.M = (size_t)g.N * g.P * g.Q * g.K;
        break;
    }
    case MQ_FORM_WINO1D: {
        const mqConvDesc &g = p.gemm;                  // (np N th, 1, W, C) -> (.., 1, Q, K), filters (np G K, S, C)
        s.V = (size_t)g.N * g.W * g.C; s.U = (size_t)g.groups * g.K * g.S * g.C; s.Uc = s.U; s.M = (size_t)g.N * g.Q * g.K;
        if (p.nested) {
            const mqConvDesc &h = p.inner;
            s.V2 = (size_t)h.N * h.H * h.W * h.C; s.U2 = (size_t)h.groups * h.K * h.R * h.C; s.Uc2 = s.U2; s.M2 = (size_t)h.N * h.P * h.Q * h.K;
        }
        break;
    }
    default: break;
    }
    return s;
}

// what forward hands to backward: the transformed input the weight gradient multiplies with and the CRSK transformed filters the
// input gradient multiplies with (the two-dimensional forms transform the rotated filter in the backward pass instead)
void keep_layout(const mqConvPlan &p, const Sizes &s, size_t *v_floats, size_t *uc_floats)
{
    if (p.form == MQ_FORM_WINO1D && p.nested) { *v_floats = s.V2; *uc_floats = s.Uc2; }
    else { *v_floats = s.V; *uc_floats = s.Uc; }
}

// every launch of the planned paths goes through MQ_TRY: skipped when the path runs dry (workspace sizing)
#define MQ_TRY(call)                                                                                              \
    do {                                                                                                          \
        if (!dry) {                                                                                               \
            if (ws.over()) return fail("planned call: workspace carved past its end (%zu > %zu bytes)", ws.peak, ws.cap); \
            if ((call) != 0) return 1;                                                                            \
        }                                                                                                         \
    } while (0)

// one HBM-bound transform launch of a reduced form, bracketed for the pass hook (bench.py times the family with HIP events)
#define MQ_PASS(call)                                                      \
    do {                                                                   \
        if (!dry && g_pass) g_pass(0, 0.0, "winograd_transforms");         \
        MQ_TRY(call);                                                      \
        if (!dry && g_pass) g_pass(1, 0.0, "winograd_transforms");         \
    } while (0)
// the layer's ALGORITHMIC activation bytes for this pass (what a fused kernel would have to move: every activation tensor the
// pass reads or writes, once), reported once per planned call of a reduced form
#define MQ_LAYER_BYTES(bytes)                                              \
    do {                                                                   \
        if (!dry && g_pass) g_pass(2, (double)(bytes), mq_form_name(p->form)); \
    } while (0)

// one GEMM launch, reported to the launch hook (bench.py's live HIP-event timing) with its algorithmic flops
template <class F>
int gemm_call(const mqConvDesc &g, F &&f)
{
    if (g_hook) g_hook(0, gemm_flops(g), "");
    const int rc = f();
    if (g_hook) g_hook(1, gemm_flops(g), mq_last_kernel());
    return rc;
}

int fprop_impl(const mqConvPlan *p, const float *x, const float *w_krsc, const float *bias, int act, float slope, const uint8_t *mask,
               int64_t d1, int64_t d2, int64_t d3, float *y, void *keep, void *workspace, size_t workspace_bytes, mqStream st, bool dry,
               size_t *peak);
int backward_impl(const mqConvPlan *p, const float *x, const void *keep, const float *w_krsc, const float *w_crsk, const float *dy,
                  const float *gate, float gate_slope, const uint8_t *gate_mask, int64_t g1, int64_t g2, int64_t g3, float *dx, float *dw,
                  float *db, void *workspace, size_t workspace_bytes, mqStream st, bool dry, size_t *peak);

}  // namespace

// ---------------------------------------------------------------------------------------------------- C ABI
extern "C" void mq_debug_set_launch_hook(mqLaunchHook hook) { g_hook = hook; }
extern "C" void mq_debug_set_pass_hook(mqPassHook hook) { g_pass = hook; }

extern "C" void mq_plan_cfg_default(mqPlanCfg *c)
{
    c->wino_minc = env_int("MQGAN_WINO_MINC", 128, &c->wino_minc_explicit);
    const char *t = getenv("MQGAN_WINO_TILE");
    c->wino_tile = (t && strcmp(t, "auto") != 0) ? atoi(t) : 0;
    c->s2d_minc = env_int("MQGAN_WINO_S2D_MINC", 128);
    c->wino1d_minc = env_int("MQGAN_WINO1D_MINC", 128);
    c->wino1d_tile = env_int("MQGAN_WINO1D_TILE", 0);
    c->winow_minc = env_int("MQGAN_WINOW_MINC", 128);
    c->keep_mib = env_int("MQGAN_WINO_KEEP_MIB", 6144);
}

extern "C" const char *mq_form_name(int form)
{
    switch (form) {
    case MQ_FORM_DIRECT: return "direct";
    case MQ_FORM_WINO2D: return "winograd2d";
    case MQ_FORM_S2D: return "polyphase-winograd2d";
    case MQ_FORM_WINO1D: return "winograd-rows";
    case MQ_FORM_WINOW: return "winograd-columns";
    default: return "?";
    }
}

extern "C" int mq_conv_plan(const mqConvDesc *desc, int leaky, const mqPlanCfg *cfg_in, mqConvPlan *p)
{
    MQ_REQUIRE(desc && p, "mq_conv_plan: null argument");
    mqPlanCfg cfg;
    if (cfg_in) cfg = *cfg_in;
    else mq_plan_cfg_default(&cfg);
    MQ_REQUIRE(cfg.wino_tile == 0 || cfg.wino_tile == 3 || cfg.wino_tile == 4 || cfg.wino_tile == 45 || cfg.wino_tile == 6,
               "mq_conv_plan: wino_tile must be 0 (auto), 3, 4, 45 or 6, got %d (F(2x2, 3x3) left the library in ABI 8)", cfg.wino_tile);
    const mqConvDesc &d = *desc;
    memset(p, 0, sizeof(*p));
    p->desc = d;
    p->leaky = leaky != 0;
    p->form = MQ_FORM_DIRECT;
    p->gemm = d;
    const bool vec = d.C % 4 == 0;             // the fused LeakyReLU epilogue and every transform need the vector path
    int mh = 0, mw = 0;
    if (leaky && vec && wino1d_ok(d, cfg)) {
        const int m = (cfg.wino1d_tile == 2 || cfg.wino1d_tile == 4 || cfg.wino1d_tile == 5) ? cfg.wino1d_tile : mq_wino1d_tile(d.H);
        const int th = cdivi(d.H, m), np = m + 2;
        p->form = MQ_FORM_WINO1D;
        p->mh = m;
        p->npos = np;
        p->tiles = (int64_t)d.N * th;
        p->gemm = make_desc(np * d.N * th, 1, d.W, d.C, d.K, 1, d.S, 1, d.stride_w, 0, d.pad_w, np * d.groups, 1, d.Q, d.precision);
        if (winow_ok(p->gemm, cfg)) {
            p->nested = 1;
            p->mw = mq_winow_tile(d.S);
            p->inner = winow_desc(p->gemm);
        }
    } else if (leaky && vec && s2d_ok(d, cfg, &mh, &mw)) {
        p->form = MQ_FORM_S2D;
        p->mh = mh; p->mw = mw;
        p->npos = (mh + 2) * (mw + 2);
        p->tiles = (int64_t)d.N * cdivi(d.P, mh) * cdivi(d.Q, mw);
        p->gemm = wino2d_desc(d.N, d.P, d.Q, 4 * d.C, d.K, mh, mw, d.precision);
    } else if (!leaky && wino2d_ok(d, cfg, &mh, &mw)) {
        p->form = MQ_FORM_WINO2D;
        p->mh = mh; p->mw = mw;
        p->npos = (mh + 2) * (mw + 2);
        p->tiles = (int64_t)d.N * cdivi(d.H, mh) * cdivi(d.W, mw);
        p->gemm = wino2d_desc(d.N, d.H, d.W, d.C, d.K, mh, mw, d.precision);
    } else if (vec && winow_ok(d, cfg)) {
        p->form = MQ_FORM_WINOW;
        p->mw = mq_winow_tile(d.S);
        p->npos = p->mw + d.S - 1;
        p->tiles = (int64_t)d.N * d.P * cdivi(d.Q, p->mw);
        p->gemm = winow_desc(d);
    }
    const mqConvDesc &mult = (p->form == MQ_FORM_WINO1D && p->nested) ? p->inner : p->gemm;
    p->flops = gemm_flops(mult);
    const Sizes s = sizes_of(*p);
    size_t v = 0, uc = 0;
    keep_layout(*p, s, &v, &uc);
    p->keep_bytes = (p->form != MQ_FORM_DIRECT && v * 4 <= ((size_t)cfg.keep_mib << 20)) ? al(v * 4) + al(uc * 4) : 0;
    p->keep_uc_bytes = (p->form == MQ_FORM_WINO1D || p->form == MQ_FORM_WINOW) ? al(uc * 4) : 0;
    // workspace sizes: the planned paths themselves, run dry (no launches, a measuring carver) in their most demanding
    // configuration - no keep buffer, every gradient wanted
    size_t peak = 0;
    float dummy = 0.f;       // non-NULL stand-ins for "wanted"
    if (fprop_impl(p, &dummy, &dummy, nullptr, leaky ? 2 : 0, 0.2f, nullptr, 1, 0, 1, &dummy, nullptr, nullptr, 0, nullptr, true, &peak)) return 1;
    p->ws_fprop = peak;
    if (backward_impl(p, &dummy, nullptr, &dummy, &dummy, &dummy, nullptr, 1.f, nullptr, 1, 0, 1, &dummy, &dummy, &dummy, nullptr, 0,
                      nullptr, true, &peak))
        return 1;
    p->ws_backward = peak;
    return 0;
}

// y = epilogue(conv(x, w) + bias).  act 0: none; act 2: LeakyReLU(slope), then zero where mask[(m / d1) * d2 + m % d3] (m = output
// pixel; mask may be NULL).  keep: NULL or plan->keep_bytes bytes that mq_conv_backward_planned will read.
extern "C" int mq_conv_fprop_planned(const mqConvPlan *p, const float *x, const float *w_krsc, const float *bias, int act, float slope,
                                     const uint8_t *mask, int64_t d1, int64_t d2, int64_t d3, float *y, void *keep, void *workspace,
                                     size_t workspace_bytes, mqStream st)
{
    MQ_REQUIRE(p && x && w_krsc && y, "mq_conv_fprop_planned: null argument");
    MQ_REQUIRE(act == 0 || act == 2, "mq_conv_fprop_planned: act 0 or 2");
    MQ_REQUIRE(act == 2 || !mask, "mq_conv_fprop_planned: a mask needs act 2");
    MQ_REQUIRE((act == 2) == (p->leaky != 0), "mq_conv_fprop_planned: the plan was made for %s", p->leaky ? "a LeakyReLU epilogue (act 2)" : "a plain store (act 0)");
    MQ_REQUIRE(p->ws_fprop == 0 || (workspace && workspace_bytes >= p->ws_fprop), "mq_conv_fprop_planned: workspace of %zu bytes needed, %zu given",
               p->ws_fprop, workspace_bytes);
    MQ_REQUIRE(((uintptr_t)workspace | (uintptr_t)keep) % 256 == 0, "mq_conv_fprop_planned: workspace and keep buffer must be 256-byte aligned");
    return fprop_impl(p, x, w_krsc, bias, act, slope, mask, d1, d2, d3, y, keep, workspace, workspace_bytes, st, false, nullptr);
}

namespace {
int fprop_impl(const mqConvPlan *p, const float *x, const float *w_krsc, const float *bias, int act, float slope, const uint8_t *mask,
               int64_t d1, int64_t d2, int64_t d3, float *y, void *keep, void *workspace, size_t workspace_bytes, mqStream st, bool dry,
               size_t *peak)
{
    if (!mask) { d1 = 1; d2 = 0; d3 = 1; }
    const mqConvDesc &d = p->desc;
    if (peak) *peak = 0;
    const bool fonly = p->keep_filters_only != 0 && p->keep_uc_bytes > 0;      // the keep buffer holds the CRSK filters only
    Carver ws(workspace, workspace_bytes), kp(keep, keep ? (fonly ? p->keep_uc_bytes : p->keep_bytes) : 0);
    if (p->form == MQ_FORM_DIRECT) {
        if (act == 2) MQ_TRY(mq_conv_fprop_leaky(&d, x, w_krsc, bias, mask, d1, d2, d3, slope, y, st));
        else MQ_TRY(mq_conv_fprop(&d, x, w_krsc, bias, y, st));
        return 0;
    }
    MQ_LAYER_BYTES(4.0 * ((double)d.N * d.H * d.W * d.C + (double)d.N * d.P * d.Q * d.K));
    const Sizes s = sizes_of(*p);
    size_t kv = 0, kuc = 0;
    keep_layout(*p, s, &kv, &kuc);
    struct PeakOut {
        Carver &c; size_t *out;
        ~PeakOut() { if (out) *out = c.peak; }
    } peak_out{ws, peak};
    const bool keeping = keep != nullptr && (fonly || p->keep_bytes > 0);
    float *keepV = (keeping && !fonly) ? kp.take(kv) : nullptr, *keepUc = keeping ? kp.take(kuc) : nullptr;
    const int mh = p->mh, mw = p->mw;
    switch (p->form) {
    case MQ_FORM_WINO2D: {
        float *V = keepV ? keepV : ws.take(s.V), *U = ws.take(s.U), *M = ws.take(s.M);
        MQ_PASS(mq_winon_input_transform(x, d.N, d.H, d.W, d.C, V, mh, mw, st));
        MQ_PASS(mq_winon_weight_transform(w_krsc, d.K, d.C, 0, U, mh, mw, st));
        MQ_TRY(gemm_call(p->gemm, [&] { return mq_conv_fprop(&p->gemm, V, U, nullptr, M, st); }));
        MQ_PASS(mq_winon_output_transform(M, bias, d.N, d.H, d.W, d.K, act, slope, mask, d1, d2, d3, y, mh, mw, st));
        return 0;
    }
    case MQ_FORM_S2D: {
        float *V = keepV ? keepV : ws.take(s.V), *U = ws.take(s.U), *M = ws.take(s.M);
        MQ_PASS(mq_winon_input_transform_s2d(x, d.N, d.H, d.W, d.C, V, mh, mw, st));
        MQ_PASS(mq_winon_weight_transform_s2d(w_krsc, d.K, d.C, 0, U, mh, mw, st));
        MQ_TRY(gemm_call(p->gemm, [&] { return mq_conv_fprop(&p->gemm, V, U, nullptr, M, st); }));
        MQ_PASS(mq_winon_output_transform(M, bias, d.N, d.P, d.Q, d.K, act, slope, mask, d1, d2, d3, y, mh, mw, st));
        return 0;
    }
    case MQ_FORM_WINOW: {
        float *V = keepV ? keepV : ws.take(s.V), *U = ws.take(s.U), *Uc = keepUc ? keepUc : ws.take(s.Uc), *M = ws.take(s.M);
        MQ_PASS(mq_winow_weight_transform(w_krsc, d.groups, d.K, d.C, d.R, d.S, U, Uc, st));
        MQ_PASS(mq_winow_input_transform(x, d.N, d.H, d.W, d.C, d.S, d.Q, d.pad_w, V, st));
        MQ_TRY(gemm_call(p->gemm, [&] { return mq_conv_fprop(&p->gemm, V, U, nullptr, M, st); }));
        MQ_PASS(mq_winow_output_transform(M, bias, d.N, d.P, d.Q, d.K, d.S, d.groups, act, slope, mask, d1, d2, d3, y, st));
        return 0;
    }
    case MQ_FORM_WINO1D: {
        const int m = mh;
        const mqConvDesc &g = p->gemm;
        float *V = (keepV && !p->nested) ? keepV : ws.take(s.V), *U = ws.take(s.U), *Uc = (keepUc && !p->nested) ? keepUc : ws.take(s.Uc);
        float *M = ws.take(s.M);
        MQ_PASS(mq_wino1d_input_transform(x, d.N, d.H, d.W, d.C, m, V, st));
        MQ_PASS(mq_wino1d_weight_transform(w_krsc, d.groups, d.K, d.C, d.S, m, U, Uc, st));
        if (p->nested) {      // the row convolutions themselves as F(m', S) along the columns
            const mqConvDesc &h = p->inner;
            float *V2 = keepV ? keepV : ws.take(s.V2), *U2 = ws.take(s.U2), *Uc2 = keepUc ? keepUc : ws.take(s.Uc2), *M2 = ws.take(s.M2);
            MQ_PASS(mq_winow_weight_transform(U, g.groups, g.K, g.C, g.R, g.S, U2, Uc2, st));
            MQ_PASS(mq_winow_input_transform(V, g.N, g.H, g.W, g.C, g.S, g.Q, g.pad_w, V2, st));
            MQ_TRY(gemm_call(h, [&] { return mq_conv_fprop(&h, V2, U2, nullptr, M2, st); }));
            MQ_PASS(mq_winow_output_transform(M2, nullptr, g.N, g.P, g.Q, g.K, g.S, g.groups, 0, 0.f, nullptr, 1, 0, 1, M, st));
        } else {
            MQ_TRY(gemm_call(g, [&] { return mq_conv_fprop(&g, V, U, nullptr, M, st); }));
        }
        MQ_PASS(mq_wino1d_output_transform(M, bias, d.N, d.H, d.Q, d.K, d.groups, m, act, slope, mask, d1, d2, d3, y, st));
        return 0;
    }
    default: return fail("mq_conv_fprop_planned: unknown form %d", p->form);
    }
}
}  // namespace

// Gradients of one layer from dy, the gradient with respect to the convolution's output BEFORE its activation (a LeakyReLU layer's
// caller applies LeakyReLU'(y) and the mask first - mq_pointwise_bwd - unless the consumer of y already did: the gated chains of
// DESIGN 3.5).  dx / dw / db: NULL = not wanted.  x: the layer's input (needed when `keep` is NULL and dw is wanted, and by the
// direct form).  w_crsk: the direct form's input gradient only.  gate (may be NULL): x itself when x came out of a LeakyReLU(gate_slope)
// + mask layer that expects its gradient with respect to ITS pre-activation: dx *= x > 0 ? 1 : gate_slope, zero where
// gate_mask[(m / g1) * g2 + m % g3] (m = input pixel).
extern "C" int mq_conv_backward_planned(const mqConvPlan *p, const float *x, const void *keep, const float *w_krsc, const float *w_crsk,
                                        const float *dy, const float *gate, float gate_slope, const uint8_t *gate_mask, int64_t g1,
                                        int64_t g2, int64_t g3, float *dx, float *dw, float *db, void *workspace, size_t workspace_bytes,
                                        mqStream st)
{
    MQ_REQUIRE(p && dy && w_krsc, "mq_conv_backward_planned: null argument");
    MQ_REQUIRE(!gate_mask || gate, "mq_conv_backward_planned: a gate mask needs the gate tensor");
    MQ_REQUIRE(p->ws_backward == 0 || (workspace && workspace_bytes >= p->ws_backward),
               "mq_conv_backward_planned: workspace of %zu bytes needed, %zu given", p->ws_backward, workspace_bytes);
    MQ_REQUIRE(((uintptr_t)workspace | (uintptr_t)keep) % 256 == 0, "mq_conv_backward_planned: workspace and keep buffer must be 256-byte aligned");
    MQ_REQUIRE(p->form != MQ_FORM_DIRECT || !dx || w_crsk, "mq_conv_backward_planned: the direct input gradient needs the CRSK weights");
    MQ_REQUIRE(!dw || x || (keep && p->keep_bytes > 0 && !p->keep_filters_only), "mq_conv_backward_planned: the weight gradient needs the full keep buffer or x");
    MQ_REQUIRE(p->form != MQ_FORM_WINO2D || !gate, "mq_conv_backward_planned: the plain 3x3 form has no gated input gradient (its layers feed APTx, not LeakyReLU)");
    return backward_impl(p, x, keep, w_krsc, w_crsk, dy, gate, gate_slope, gate_mask, g1, g2, g3, dx, dw, db, workspace, workspace_bytes, st, false,
                         nullptr);
}

namespace {
int backward_impl(const mqConvPlan *p, const float *x, const void *keep, const float *w_krsc, const float *w_crsk, const float *dy,
                  const float *gate, float gate_slope, const uint8_t *gate_mask, int64_t g1, int64_t g2, int64_t g3, float *dx, float *dw,
                  float *db, void *workspace, size_t workspace_bytes, mqStream st, bool dry, size_t *peak)
{
    if (!gate_mask) { g1 = 1; g2 = 0; g3 = 1; }
    if (!gate) gate_slope = 1.f;
    const mqConvDesc &d = p->desc;
    Carver ws(workspace, workspace_bytes);
    struct PeakOut {
        Carver &c; size_t *out;
        ~PeakOut() { if (out) *out = c.peak; }
    } peak_out{ws, peak};
    const size_t rowsK = (size_t)d.groups * d.K;
    if (p->form == MQ_FORM_DIRECT) {
        if (dx) {
            if (gate) MQ_TRY(gemm_call(d, [&] { return mq_conv_dgrad_gated(&d, dy, w_crsk, gate, gate_mask, g1, g2, g3, gate_slope, dx, st); }));
            else MQ_TRY(gemm_call(d, [&] { return mq_conv_dgrad(&d, dy, w_crsk, dx, st); }));
        }
        if (dw) {
            const size_t nb = std::max(mq_conv_wgrad_workspace(&d), mq_conv_wgrad_bias_workspace(&d));
            void *w = ws.take_bytes(nb);
            if (db) MQ_TRY(gemm_call(d, [&] { return mq_conv_wgrad_bias(&d, x, dy, dw, db, w, mq_conv_wgrad_bias_workspace(&d), st); }));
            else MQ_TRY(gemm_call(d, [&] { return mq_conv_wgrad(&d, x, dy, dw, w, mq_conv_wgrad_workspace(&d), st); }));
        } else if (db) {
            // one group's column sums at a time: the scratch is reused, so it is 1024 x K floats whatever the group count (the
            // sizing pass runs with dw wanted and reserves at least that much)
            float *cw = ws.take(1024 * (size_t)d.K);
            const int64_t rows = (int64_t)(d.N / d.groups) * d.P * d.Q;
            for (int g = 0; g < d.groups; ++g) MQ_TRY(mq_colsum(dy + (size_t)g * rows * d.K, rows, d.K, db + (size_t)g * d.K, cw, st));
        }
        return 0;
    }
    MQ_LAYER_BYTES(4.0 * ((double)d.N * d.P * d.Q * d.K + ((dx ? 1.0 : 0.0) + (dw ? 1.0 : 0.0)) * (double)d.N * d.H * d.W * d.C));
    const Sizes s = sizes_of(*p);
    size_t kv = 0, kuc = 0;
    keep_layout(*p, s, &kv, &kuc);
    const bool fonly = p->keep_filters_only != 0 && p->keep_uc_bytes > 0;      // the keep buffer holds the CRSK filters only
    const bool kept = keep != nullptr && (fonly || p->keep_bytes > 0);
    const float *keepV = (kept && !fonly) ? static_cast<const float *>(keep) : nullptr;
    const float *keepUc = kept ? reinterpret_cast<const float *>(static_cast<const char *>(keep) + (fonly ? 0 : al(kv * 4))) : nullptr;
    const int mh = p->mh, mw = p->mw;
    float *colws = ws.take(1024 * rowsK);
    auto bias_from = [&](const float *part, size_t part_rows) -> int {
        if (part && part_rows) return mq_colsum(part, (int64_t)part_rows, (int)rowsK, db, colws, st);
        const int64_t rows = (int64_t)(d.N / d.groups) * d.P * d.Q;
        for (int g = 0; g < d.groups; ++g) MQ_TRY(mq_colsum(dy + (size_t)g * rows * d.K, rows, d.K, db + (size_t)g * d.K, colws, st));
        return 0;
    };

    if (p->form == MQ_FORM_WINO2D || p->form == MQ_FORM_S2D) {
        const bool s2d = p->form == MQ_FORM_S2D;
        const int GH = s2d ? d.P : d.H, GW = s2d ? d.Q : d.W;          // the grid the tiles cover
        const int cin = (s2d ? 4 : 1) * d.C;
        const size_t prow = mq_winon_dy_fused_rows(d.N, GH, GW, d.K, mh, mw);
        // decisions by what is WANTED, never by the carved pointers: those are NULL in the dry (sizing) run
        const bool need_Z = dw != nullptr, need_Vd = dx != nullptr;
        const bool need_part = db != nullptr && prow > 0 && need_Z;        // the column sums ride on the pass that writes Z
        float *Z = need_Z ? ws.take(s.M) : nullptr;
        float *part = need_part ? ws.take(prow * d.K) : nullptr;
        const size_t m0 = ws.mark();
        float *Vd = need_Vd ? ws.take(s.M) : nullptr;
        // one pass over dy: Z = A dY A^T (weight gradient), Vd = B^T dy B (input gradient), per-block column sums (bias)
        if (need_Z) MQ_PASS(mq_winon_dy_fused(dy, d.N, GH, GW, d.K, Z, Vd, part, mh, mw, st));
        else if (need_Vd) MQ_PASS(mq_winon_input_transform(dy, d.N, GH, GW, d.K, Vd, mh, mw, st));
        if (dx) {
            float *Ut = ws.take(s.U), *Mt = ws.take((size_t)p->npos * p->tiles * cin);
            const mqConvDesc gt = wino2d_desc(d.N, GH, GW, d.K, cin, mh, mw, d.precision);
            if (s2d) MQ_PASS(mq_winon_weight_transform_s2d(w_krsc, d.K, d.C, 1, Ut, mh, mw, st));
            else MQ_PASS(mq_winon_weight_transform(w_krsc, d.K, d.C, 1, Ut, mh, mw, st));
            MQ_TRY(gemm_call(gt, [&] { return mq_conv_fprop(&gt, Vd, Ut, nullptr, Mt, st); }));
            if (s2d) MQ_PASS(mq_winon_output_transform_d2s(Mt, d.N, d.H, d.W, d.C, gate, gate_slope, gate_mask, g1, g2, g3, dx, mh, mw, st));
            else MQ_PASS(mq_winon_output_transform(Mt, nullptr, d.N, d.H, d.W, d.C, 0, 0.f, nullptr, 1, 0, 1, dx, mh, mw, st));
        }
        ws.rewind(m0);
        if (dw) {
            const float *V = keepV;
            if (!V) {
                float *Vn = ws.take(s.V);
                if (s2d) MQ_PASS(mq_winon_input_transform_s2d(x, d.N, d.H, d.W, d.C, Vn, mh, mw, st));
                else MQ_PASS(mq_winon_input_transform(x, d.N, d.H, d.W, d.C, Vn, mh, mw, st));
                V = Vn;
            }
            float *P = ws.take(s.U);
            const size_t nb = mq_conv_wgrad_workspace(&p->gemm);
            void *w = ws.take_bytes(nb);
            MQ_TRY(gemm_call(p->gemm, [&] { return mq_conv_wgrad(&p->gemm, V, Z, P, w, nb, st); }));
            if (s2d) MQ_PASS(mq_winon_wgrad_inverse_s2d(P, d.K, d.C, dw, mh, mw, st));
            else MQ_PASS(mq_winon_wgrad_inverse(P, d.K, d.C, dw, mh, mw, st));
        }
        if (db) MQ_TRY(bias_from(part, need_part ? prow : 0));
        return 0;
    }

    if (p->form == MQ_FORM_WINOW) {
        const mqConvDesc &g = p->gemm;
        const size_t prow = db ? mq_winow_dy_rows(d.N, d.P, d.Q, d.K, d.S, d.groups) : 0;
        float *Z = ws.take(s.M), *part = prow ? ws.take(prow * rowsK) : nullptr;
        MQ_PASS(mq_winow_dy_transform(dy, d.N, d.P, d.Q, d.K, d.S, d.groups, Z, part, st));
        const size_t m0 = ws.mark();
        const float *V = keepV, *Uc = keepUc;
        if ((dx && !Uc) || (dw && !V)) {           // no keep buffer: transform again
            float *U = ws.take(s.U), *Ucn = ws.take(s.Uc);
            MQ_PASS(mq_winow_weight_transform(w_krsc, d.groups, d.K, d.C, d.R, d.S, U, Ucn, st));
            Uc = Ucn;
            if (dw) {
                float *Vn = ws.take(s.V);
                MQ_PASS(mq_winow_input_transform(x, d.N, d.H, d.W, d.C, d.S, d.Q, d.pad_w, Vn, st));
                V = Vn;
            }
        }
        const size_t m1 = ws.mark();
        if (dx) {
            float *dV = ws.take(s.V);
            MQ_TRY(gemm_call(g, [&] { return mq_conv_dgrad(&g, Z, Uc, dV, st); }));
            MQ_PASS(mq_winow_dx_transform(dV, d.N, d.H, d.W, d.C, d.S, d.Q, d.pad_w, gate, gate_slope, gate_mask, g1, g2, g3, dx, st));
        }
        ws.rewind(m1);
        if (dw) {
            float *dU = ws.take(s.U);
            const size_t nb = mq_conv_wgrad_workspace(&g);
            void *w = ws.take_bytes(nb);
            MQ_TRY(gemm_call(g, [&] { return mq_conv_wgrad(&g, V, Z, dU, w, nb, st); }));
            MQ_PASS(mq_winow_wgrad_inverse(dU, d.groups, d.K, d.C, d.R, d.S, dw, st));
        }
        (void)m0;
        if (db) MQ_TRY(bias_from(part, prow));
        return 0;
    }

    if (p->form == MQ_FORM_WINO1D) {
        const int m = mh;
        const mqConvDesc &g = p->gemm, &h = p->inner;
        const size_t prow = db ? mq_wino1d_dy_rows(d.N, d.H, d.Q, d.K, d.groups, m) : 0;
        float *Z = ws.take(s.M), *part = prow ? ws.take(prow * rowsK) : nullptr;
        MQ_PASS(mq_wino1d_dy_transform(dy, d.N, d.H, d.Q, d.K, d.groups, m, Z, part, st));
        float *Z2 = nullptr;
        if (p->nested) {
            Z2 = ws.take(s.M2);
            MQ_PASS(mq_winow_dy_transform(Z, g.N, g.P, g.Q, g.K, g.S, g.groups, Z2, nullptr, st));
        }
        const float *V = keepV, *Uc = keepUc;              // nested: the column-transformed ones
        if ((dx && !Uc) || (dw && !V)) {
            float *U1 = ws.take(s.U), *Uc1 = ws.take(s.Uc), *V1 = nullptr;
            MQ_PASS(mq_wino1d_weight_transform(w_krsc, d.groups, d.K, d.C, d.S, m, U1, Uc1, st));
            if (dw) {
                V1 = ws.take(s.V);
                MQ_PASS(mq_wino1d_input_transform(x, d.N, d.H, d.W, d.C, m, V1, st));
            }
            if (p->nested) {
                float *U2 = ws.take(s.U2), *Uc2 = ws.take(s.Uc2);
                MQ_PASS(mq_winow_weight_transform(U1, g.groups, g.K, g.C, g.R, g.S, U2, Uc2, st));
                Uc = Uc2;
                if (dw) {
                    float *V2 = ws.take(s.V2);
                    MQ_PASS(mq_winow_input_transform(V1, g.N, g.H, g.W, g.C, g.S, g.Q, g.pad_w, V2, st));
                    V = V2;
                }
            } else {
                Uc = Uc1;
                V = V1;
            }
        }
        const size_t m1 = ws.mark();
        if (dx) {
            float *dV = ws.take(s.V);
            if (p->nested) {
                float *dV2 = ws.take(s.V2);
                MQ_TRY(gemm_call(h, [&] { return mq_conv_dgrad(&h, Z2, Uc, dV2, st); }));
                MQ_PASS(mq_winow_dx_transform(dV2, g.N, g.H, g.W, g.C, g.S, g.Q, g.pad_w, nullptr, 1.f, nullptr, 1, 0, 1, dV, st));
            } else {
                MQ_TRY(gemm_call(g, [&] { return mq_conv_dgrad(&g, Z, Uc, dV, st); }));
            }
            MQ_PASS(mq_wino1d_dx_transform(dV, d.N, d.H, d.W, d.C, m, gate, gate_slope, gate_mask, g1, g2, g3, dx, st));
        }
        ws.rewind(m1);
        if (dw) {
            float *dU = ws.take(s.U);
            if (p->nested) {
                float *dU2 = ws.take(s.U2);
                const size_t nb = mq_conv_wgrad_workspace(&h);
                void *w = ws.take_bytes(nb);
                MQ_TRY(gemm_call(h, [&] { return mq_conv_wgrad(&h, V, Z2, dU2, w, nb, st); }));
                MQ_PASS(mq_winow_wgrad_inverse(dU2, g.groups, g.K, g.C, g.R, g.S, dU, st));
            } else {
                const size_t nb = mq_conv_wgrad_workspace(&g);
                void *w = ws.take_bytes(nb);
                MQ_TRY(gemm_call(g, [&] { return mq_conv_wgrad(&g, V, Z, dU, w, nb, st); }));
            }
            MQ_PASS(mq_wino1d_wgrad_inverse(dU, d.groups, d.K, d.C, d.S, m, dw, st));
        }
        if (db) MQ_TRY(bias_from(part, prow));
        return 0;
    }
    return fail("mq_conv_backward_planned: unknown form %d", p->form);
}
}  // namespace
