// Shared device helpers for libpnr_hip.so (gfx950 only; wavefront = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pnr.h"

#define PNR_WAVE 64

#define PNR_HIP_CHECK(expr)                         \
    do {                                            \
        hipError_t _e = (expr);                     \
        if (_e != hipSuccess) return (int32_t)_e;   \
    } while (0)

#define PNR_LAUNCH_CHECK()                          \
    do {                                            \
        hipError_t _e = hipGetLastError();          \
        if (_e != hipSuccess) return (int32_t)_e;   \
    } while (0)

namespace pnr {

// ---------------------------------------------------------------- counter-based RNG (Philox4x32-10)
// Keyed by (seed, global ray index, draw id); the same ray gets the same draws on any GPU / any
// sharding, so an N-GPU frame is bit-identical to the 1-GPU frame.
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // [0,1)

enum { DRAW_COARSE = 0, DRAW_U = 1, DRAW_R = 2, DRAW_G = 3 };

__device__ __forceinline__ float rng_uniform(uint64_t seed, int64_t ray, int draw, int idx) {
    u32x4 v = philox4x32(seed, (uint32_t)ray, (uint32_t)((uint64_t)ray >> 32), (uint32_t)draw, (uint32_t)idx);
    return u01(v.x);
}
__device__ __forceinline__ float rng_normal(uint64_t seed, int64_t ray, int draw, int idx) {
    u32x4 v = philox4x32(seed, (uint32_t)ray, (uint32_t)((uint64_t)ray >> 32), (uint32_t)draw, (uint32_t)idx);
    float a = 1.0f - u01(v.x);                    // (0,1]
    float b = u01(v.y);
    return sqrtf(-2.0f * logf(a)) * cosf(6.28318530717958647692f * b);
}

// ---------------------------------------------------------------- wave-level primitives
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive scans over the 64 lanes
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v *= t;
    }
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// ---------------------------------------------------------------- cameras / projection
// One source view after PixelNeRFNet.encode (models.py.backup2:121-150).
struct Cam {
    float R[9];     // world -> camera rotation, row-major
    float t[3];
    float fx, fy;   // fy already negated
    float cx, cy;
};

__device__ __forceinline__ Cam load_cam(const pnr_views& vw, int view) {
    Cam c;
    const float* p = vw.w2c + (size_t)view * 12;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        c.R[i * 3 + 0] = p[i * 4 + 0]; c.R[i * 3 + 1] = p[i * 4 + 1]; c.R[i * 3 + 2] = p[i * 4 + 2];
        c.t[i] = p[i * 4 + 3];
    }
    const float* f = vw.focal + (vw.n_focal > 1 ? (size_t)view * 2 : 0);
    const float* pc = vw.c + (vw.n_c > 1 ? (size_t)view * 2 : 0);
    c.fx = f[0]; c.fy = f[1]; c.cx = pc[0]; c.cy = pc[1];
    return c;
}

__device__ __forceinline__ void rot3(const float* R, const float* v, float* o) {
    o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}

// uv in image pixels: -x_cam.xy / x_cam.z * (fx, fy) + c   (models.py.backup2:215-221)
__device__ __forceinline__ void project(const Cam& c, const float* xrot, float& u, float& v) {
    float xc = xrot[0] + c.t[0], yc = xrot[1] + c.t[1], zc = xrot[2] + c.t[2];
    u = (-xc / zc) * c.fx + c.cx;
    v = (-yc / zc) * c.fy + c.cy;
}

// Bilinear tap set of SpatialEncoder.index (encoder.py:152-205) + ATen grid_sampler_2d with
// align_corners=True, padding_mode=border: normalise by the LATENT size, unnormalise, clip,
// floor.  Returns the 4 tap offsets (y*W+x, clamped in-bounds) and weights (0 for out-of-bounds taps).
struct Taps { int off[4]; float w[4]; };

__device__ __forceinline__ Taps bilinear_taps(float u, float v, int W, int H) {
    float gx = (u / (float)(W - 1)) * 2.0f - 1.0f;
    float gy = (v / (float)(H - 1)) * 2.0f - 1.0f;
    float ix = ((gx + 1.0f) * 0.5f) * (float)(W - 1);
    float iy = ((gy + 1.0f) * 0.5f) * (float)(H - 1);
    ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));   // NaN -> 0 here (fmaxf), ATen's std::max/min also drops NaN to a bound
    iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
    float x0 = floorf(ix), y0 = floorf(iy);
    float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    Taps t;
    t.w[0] = (x1 - ix) * (y1 - iy);
    t.w[1] = (ix - x0) * (y1 - iy);
    t.w[2] = (x1 - ix) * (iy - y0);
    t.w[3] = (ix - x0) * (iy - y0);
    int xi0 = (int)x0, yi0 = (int)y0, xi1 = xi0 + 1, yi1 = yi0 + 1;
    bool bx1 = xi1 <= W - 1, by1 = yi1 <= H - 1;          // x0,y0 are always in bounds after the clip
    if (!bx1) { t.w[1] = 0.0f; t.w[3] = 0.0f; xi1 = xi0; }
    if (!by1) { t.w[2] = 0.0f; t.w[3] = 0.0f; yi1 = yi0; }
    t.off[0] = yi0 * W + xi0; t.off[1] = yi0 * W + xi1;
    t.off[2] = yi1 * W + xi0; t.off[3] = yi1 * W + xi1;
    return t;
}

// Derivatives of the tap weights with respect to the texel coordinate (ATen grid_sampler_2d_backward,
// align_corners=True, padding_mode=border): d w_i / d ix, d w_i / d iy; zero for out-of-bounds taps and
// zero altogether along an axis whose coordinate was clipped (clip_coordinates_set_grad: the gradient
// is dropped for ix <= 0 or ix >= W-1).  d ix / d u = 1 (the two normalisations cancel).
struct TapsGrad { float dx[4]; float dy[4]; };

__device__ __forceinline__ TapsGrad bilinear_taps_grad(float u, float v, int W, int H) {
    float gx = (u / (float)(W - 1)) * 2.0f - 1.0f;
    float gy = (v / (float)(H - 1)) * 2.0f - 1.0f;
    float rx = ((gx + 1.0f) * 0.5f) * (float)(W - 1);
    float ry = ((gy + 1.0f) * 0.5f) * (float)(H - 1);
    float mx = (rx > 0.0f && rx < (float)(W - 1)) ? 1.0f : 0.0f;
    float my = (ry > 0.0f && ry < (float)(H - 1)) ? 1.0f : 0.0f;
    float ix = fminf((float)(W - 1), fmaxf(rx, 0.0f));
    float iy = fminf((float)(H - 1), fmaxf(ry, 0.0f));
    float x0 = floorf(ix), y0 = floorf(iy);
    float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    bool bx1 = (int)x0 + 1 <= W - 1, by1 = (int)y0 + 1 <= H - 1;
    TapsGrad t;
    t.dx[0] = -(y1 - iy) * mx;                      t.dy[0] = -(x1 - ix) * my;
    t.dx[1] = bx1 ? (y1 - iy) * mx : 0.0f;          t.dy[1] = bx1 ? -(ix - x0) * my : 0.0f;
    t.dx[2] = by1 ? -(iy - y0) * mx : 0.0f;         t.dy[2] = by1 ? (x1 - ix) * my : 0.0f;
    t.dx[3] = (bx1 && by1) ? (iy - y0) * mx : 0.0f; t.dy[3] = (bx1 && by1) ? (ix - x0) * my : 0.0f;
    return t;
}

// Positional encoding element j (0-based) of PositionalEncoding.forward (code.py:30-46) for input
// vector x of dimension d (3 or 6): layout [x(d), sin(f0 x)(d), sin(f0 x + pi/2)(d), sin(f1 x)(d), ...].
__device__ __forceinline__ float posenc_elem(const float* x, int d, int j, float freq_factor) {
    if (j < d) return x[j];
    int q = (j - d) / d, i = (j - d) % d;       // q = 2*k + phase
    float f = freq_factor * (float)(1 << (q >> 1));
    float ph = (q & 1) ? 1.57079637050628662109375f : 0.0f;   // fp32(pi/2), as the reference buffer holds it
    return sinf(fmaf(x[i], f, ph));              // addcmul(phase, x, freq)
}

// Where a point comes from (see pnr_point_mlp in pnr.h).
struct PointSrc {
    const float* rays; const float* z; int K;       // rays mode
    const float* xyz; const float* dirs;            // explicit mode
};
__device__ __forceinline__ void fetch_point(const PointSrc& s, int64_t g, float* p, float* d) {
    if (s.rays) {
        int64_t ray = g / s.K;
        const float* r = s.rays + ray * 8;
        float z = s.z[g];
        d[0] = r[3]; d[1] = r[4]; d[2] = r[5];
        p[0] = r[0] + z * d[0]; p[1] = r[1] + z * d[1]; p[2] = r[2] + z * d[2];
    } else {
        p[0] = s.xyz[g * 3 + 0]; p[1] = s.xyz[g * 3 + 1]; p[2] = s.xyz[g * 3 + 2];
        d[0] = s.dirs[g * 3 + 0]; d[1] = s.dirs[g * 3 + 1]; d[2] = s.dirs[g * 3 + 2];
    }
}

}  // namespace pnr
