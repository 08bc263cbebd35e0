// Transform matrices of one-dimensional Winograd F(m, 3), m = 4 and 5:  y = A^T [(G g) . (B^T d)]  (d: m + 2 inputs, g: 3
// taps, y: m outputs).  Generated as exact rationals by tools/wino1d_matrices.py (Toom-Cook through the transposition
// principle); the point sets are the ones with the smallest f32 error among those it surveys:
//     F(4, 3): 0, 1, -1, 2, -1/2, inf        F(5, 3): 0, 1, -1, 2, -2, 1/2, inf
// In both, the first input of a tile enters position 0 only and the last input the last position only, with coefficient 1.
// Used along the rows by wino1d.hip and along both axes (nested) by winogradn.hip.
#pragma once
#include "common.h"

namespace mq {
namespace wt {
// F(3, 3) on the points 0, 1, -1, 2, inf: 25 products per 3 x 3 output tile when nested (2.78 per output); transform constants no
// larger than 4.  The tile ops picks for images too small for the 4-wide tiles (winogradn.hip).
namespace f3 {
constexpr float BT[5][5] = {{2, -1, -2, 1, 0}, {0, -2, -1, 1, 0}, {0, 2, -3, 1, 0}, {0, -1, 0, 1, 0}, {0, 2, -1, -2, 1}};
constexpr float G[5][3] = {{0.5f, 0, 0}, {-0.5f, -0.5f, -0.5f}, {-1.0f / 6, 1.0f / 6, -1.0f / 6}, {1.0f / 6, 1.0f / 3, 2.0f / 3}, {0, 0, 1}};
constexpr float AT[3][5] = {{1, 1, 1, 1, 0}, {0, 1, -1, 2, 0}, {0, 1, 1, 4, 1}};
}  // namespace f3
namespace f4 {
constexpr float BT[6][6] = {{1, 1.5f, -2, -1.5f, 1, 0},
                            {0, 1.0f / 3, 5.0f / 6, 1.0f / 6, -1.0f / 3, 0},
                            {0, 1.0f / 3, 1.0f / 6, -5.0f / 6, 1.0f / 3, 0},
                            {0, -1.0f / 30, -1.0f / 15, 1.0f / 30, 1.0f / 15, 0},
                            {0, -32.0f / 15, 16.0f / 15, 32.0f / 15, -16.0f / 15, 0},
                            {0, 1, 1.5f, -2, -1.5f, 1}};
constexpr float G[6][3] = {{1, 0, 0}, {1, 1, 1}, {1, -1, 1}, {1, 2, 4}, {1, -0.5f, 0.25f}, {0, 0, 1}};
constexpr float AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -0.5f, 0}, {0, 1, 1, 4, 0.25f, 0}, {0, 1, -1, 8, -0.125f, 1}};
}  // namespace f4
namespace f5 {
constexpr float BT[7][7] = {{1, -2, -1.25f, 2.5f, 0.25f, -0.5f, 0},
                            {0, -2.0f / 3, 2.0f / 3, 1.5f, -1.0f / 6, -1.0f / 3, 0},
                            {0, -2.0f / 9, 2.0f / 3, -7.0f / 18, -1.0f / 6, 1.0f / 9, 0},
                            {0, 1.0f / 36, -1.0f / 24, -1.0f / 18, 1.0f / 24, 1.0f / 36, 0},
                            {0, 1.0f / 60, -1.0f / 24, 0, 1.0f / 24, -1.0f / 60, 0},
                            {0, 128.0f / 45, 0, -32.0f / 9, 0, 32.0f / 45, 0},
                            {0, -2, 4, 2.5f, -5, -0.5f, 1}};
constexpr float G[7][3] = {{1, 0, 0}, {1, 1, 1}, {1, -1, 1}, {1, 2, 4}, {1, -2, 4}, {1, 0.5f, 0.25f}, {0, 0, 1}};
constexpr float AT[5][7] = {{1, 1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0.5f, 0}, {0, 1, 1, 4, 4, 0.25f, 0},
                            {0, 1, -1, 8, -8, 0.125f, 0}, {0, 1, 1, 16, 16, 0.0625f, 1}};
}  // namespace f5
// eight points 0, 1, -1, 2, -2, 1/2, -1/2, inf: F(4, 5) and F(2, 7) (the same B^T: it depends on the points only).  f32 error
// against float64 at 512 channels: F(4, 5) 9.7e-7 rms / 7.4e-6 max, F(2, 7) 1.1e-6 / 7.1e-6 (direct: 4.6e-7 / 3.3e-6); with ten
// points (F(4, 7), F(6, 5)) it is 4e-6 / 4e-5, which is why the 7-tap filter gets the 2-output tile.
namespace p8 {
constexpr float BT[8][8] = {{1, 0, -21.0f / 4, 0, 21.0f / 4, 0, -1, 0},
                            {0, -2.0f / 9, -2.0f / 9, 17.0f / 18, 17.0f / 18, -2.0f / 9, -2.0f / 9, 0},
                            {0, 2.0f / 9, -2.0f / 9, -17.0f / 18, 17.0f / 18, 2.0f / 9, -2.0f / 9, 0},
                            {0, 1.0f / 180, 1.0f / 360, -1.0f / 36, -1.0f / 72, 1.0f / 45, 1.0f / 90, 0},
                            {0, -1.0f / 180, 1.0f / 360, 1.0f / 36, -1.0f / 72, -1.0f / 45, 1.0f / 90, 0},
                            {0, 64.0f / 45, 128.0f / 45, -16.0f / 9, -32.0f / 9, 16.0f / 45, 32.0f / 45, 0},
                            {0, -64.0f / 45, 128.0f / 45, 16.0f / 9, -32.0f / 9, -16.0f / 45, 32.0f / 45, 0},
                            {0, -1, 0, 21.0f / 4, 0, -21.0f / 4, 0, 1}};
constexpr float G[8][7] = {{1, 0, 0, 0, 0, 0, 0},
                           {1, 1, 1, 1, 1, 1, 1},
                           {1, -1, 1, -1, 1, -1, 1},
                           {1, 2, 4, 8, 16, 32, 64},
                           {1, -2, 4, -8, 16, -32, 64},
                           {1, 0.5f, 0.25f, 0.125f, 0.0625f, 0.03125f, 0.015625f},
                           {1, -0.5f, 0.25f, -0.125f, 0.0625f, -0.03125f, 0.015625f},
                           {0, 0, 0, 0, 0, 0, 1}};          // F(r taps): columns 0 .. r-2 as they stand, the row of inf has its 1 at r-1
constexpr float AT[4][8] = {{1, 1, 1, 1, 1, 1, 1, 0},
                            {0, 1, -1, 2, -2, 0.5f, -0.5f, 0},
                            {0, 1, 1, 4, 4, 0.25f, 0.25f, 0},
                            {0, 1, -1, 8, -8, 0.125f, -0.125f, 0}};  // F(m outputs): rows 0 .. m-1, the column of inf has its 1 in row m-1
// F(6, 3) on the same eight points: A^T has two more rows of powers, G is the first three columns.  Nested as F(6x6, 3x3): 64
// products per 36 outputs = 1.78 per output; f32 error against float64 at 512 channels 4.6e-6 rms / 4.9e-5 max (F(4x5): 2.4e-6 /
// 2.2e-5; of the point sets tools/wino1d_matrices.py surveys for 8 points this one is the most accurate)
constexpr float AT6[6][8] = {{1, 1, 1, 1, 1, 1, 1, 0},
                             {0, 1, -1, 2, -2, 0.5f, -0.5f, 0},
                             {0, 1, 1, 4, 4, 0.25f, 0.25f, 0},
                             {0, 1, -1, 8, -8, 0.125f, -0.125f, 0},
                             {0, 1, 1, 16, 16, 0.0625f, 0.0625f, 0},
                             {0, 1, -1, 32, -32, 0.03125f, -0.03125f, 1}};
}  // namespace p8

// F(M, S) along one axis, S taps: (4, 3) on the six points of f4, (4, 5) and (2, 7) on the eight points of p8
template <int M, int S> __device__ __forceinline__ constexpr float bt2(int i, int j)
{
    if constexpr (S == 3) return f4::BT[i][j]; else return p8::BT[i][j];
}
template <int M, int S> __device__ __forceinline__ constexpr float g2(int i, int j)
{
    if constexpr (S == 3) return f4::G[i][j];
    else return i == M + S - 2 ? (j == S - 1 ? 1.f : 0.f) : p8::G[i][j];
}
template <int M, int S> __device__ __forceinline__ constexpr float at2(int i, int j)
{
    if constexpr (S == 3) return f4::AT[i][j];
    else return j == M + S - 2 ? (i == M - 1 ? 1.f : 0.f) : p8::AT[i][j];
}

static_assert(f4::BT[0][0] == 1.f && f4::BT[5][5] == 1.f && f5::BT[0][0] == 1.f && f5::BT[6][6] == 1.f, "shared-row coefficients");

template <int M> __device__ __forceinline__ constexpr float bt(int i, int j)
{
    if constexpr (M == 3) return f3::BT[i][j]; else if constexpr (M == 4) return f4::BT[i][j];
    else if constexpr (M == 5) return f5::BT[i][j]; else return p8::BT[i][j];
}
template <int M> __device__ __forceinline__ constexpr float g(int i, int j)
{
    if constexpr (M == 3) return f3::G[i][j]; else if constexpr (M == 4) return f4::G[i][j];
    else if constexpr (M == 5) return f5::G[i][j]; else return i == 7 ? (j == 2 ? 1.f : 0.f) : p8::G[i][j];
}
template <int M> __device__ __forceinline__ constexpr float at(int i, int j)
{
    if constexpr (M == 3) return f3::AT[i][j]; else if constexpr (M == 4) return f4::AT[i][j];
    else if constexpr (M == 5) return f5::AT[i][j]; else return p8::AT6[i][j];
}

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// acc += c * v for clang extended vectors (winogradn.hip's channel vectors): elementwise, packed where the width allows
template <class V> __device__ __forceinline__ void axpyv(V &acc, float c, V v)
{
    if (c == 0.f) return;
    if (c == 1.f) { acc += v; return; }
    if (c == -1.f) { acc -= v; return; }
    acc += c * v;
}
// acc += c * v with the constant known after unrolling: zero terms vanish, +-1 become adds
__device__ __forceinline__ void axpy(float4 &acc, float c, float4 v)
{
    if (c == 0.f) return;
    if (c == 1.f) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; return; }
    if (c == -1.f) { acc.x -= v.x; acc.y -= v.y; acc.z -= v.z; acc.w -= v.w; return; }
    acc.x += c * v.x; acc.y += c * v.y; acc.z += c * v.z; acc.w += c * v.w;
}
__device__ __forceinline__ void axpy1(float &acc, float c, float v)
{
    if (c == 0.f) return;
    acc += c * v;
}
}  // namespace wt
}  // namespace mq
