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
#define PNR_TRY(expr) do { int32_t _rc = (expr); if (_rc) return _rc; } while (0)

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

// Global index of a call's local ray r (the generator's key; pnr_noise.ray_index_obj_stride in pnr.h): base + r for a call
// whose objects follow each other, base + (r / rays_per_obj) * stride + r % rays_per_obj for a shard that holds a range of
// every object's rays (extra = stride - rays_per_obj; the division only runs on that path).
struct RayKey { int64_t base; int64_t extra; int64_t rays_per_obj; };
__device__ __forceinline__ int64_t global_ray(const RayKey& k, int64_t r) {
    return k.base + r + (k.extra != 0 ? (r / k.rays_per_obj) * k.extra : 0);
}

// Uniform draw `idx` of (seed, ray, draw): word idx & 3 of the Philox block idx >> 2 — four consecutive draws share one
// block, so a thread that produces four samples of a ray (k_sample_coarse) pays for one block, and every kernel that asks
// for a single draw gets the same number.
__device__ __forceinline__ u32x4 rng_block(uint64_t seed, int64_t ray, int draw, int blk) {
    return philox4x32(seed, (uint32_t)ray, (uint32_t)((uint64_t)ray >> 32), (uint32_t)draw, (uint32_t)blk);
}
__device__ __forceinline__ float rng_uniform(uint64_t seed, int64_t ray, int draw, int idx) {
    const u32x4 v = rng_block(seed, ray, draw, idx >> 2);
    const int w = idx & 3;
    return u01(w == 0 ? v.x : w == 1 ? v.y : w == 2 ? v.z : v.w);
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
// The same reductions with the lane index handed in (bit-identical results: same exchanges, same order).  The fused render
// launch passes a lane index the optimiser cannot trace to the hardware lane id, so the six permute addresses are formed where
// they are used instead of being hoisted to the kernel's entry and kept (spilled) across the MFMA blocks.
__device__ __forceinline__ float shfl_at(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ float wave_sum_l(float v, int lane) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += shfl_at(v, lane ^ o);
    return v;
}
__device__ __forceinline__ float wave_scan_mul_l(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = shfl_at(v, lane >= o ? lane - o : lane);
        if (lane >= o) v *= t;
    }
    return v;
}

// The compositing's scan and sums on DPP row shifts / row broadcasts (gfx9 wave64: shifts by 1, 2, 4, 8 inside each row of 16
// lanes, then lane 15 of a row into the next row, then lane 31 into the upper half) — VALU moves with no LDS crossbar
// traffic and no lane index at all.  Lanes a step has no source for keep `ident`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float ident, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_scan_mul_dpp(float v) {        // inclusive product over lanes 0 .. l
    v *= dpp_f<0x111, 0xf>(1.0f, v);      // row_shr:1
    v *= dpp_f<0x112, 0xf>(1.0f, v);      // row_shr:2
    v *= dpp_f<0x114, 0xf>(1.0f, v);      // row_shr:4
    v *= dpp_f<0x118, 0xf>(1.0f, v);      // row_shr:8
    v *= dpp_f<0x142, 0xa>(1.0f, v);      // row_bcast:15 into rows 1, 3
    v *= dpp_f<0x143, 0xc>(1.0f, v);      // row_bcast:31 into rows 2, 3
    return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {             // the sum of all 64 lanes, in every lane
    v += dpp_f<0x111, 0xf>(0.0f, v);
    v += dpp_f<0x112, 0xf>(0.0f, v);
    v += dpp_f<0x114, 0xf>(0.0f, v);
    v += dpp_f<0x118, 0xf>(0.0f, v);
    v += dpp_f<0x142, 0xa>(0.0f, v);
    v += dpp_f<0x143, 0xc>(0.0f, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
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

// Explicit fma chains here and in the per-ray stages below: under hipcc's default -ffp-contract=fast a sum of products may be
// fused either way round, and which way depends on the surrounding code — the same source line inlined into two kernels
// then differs in the last bit.  The fused render launch is bit-identical to the stage kernels only with the order spelled out.
__device__ __forceinline__ void rot3(const float* R, const float* v, float* o) {
    o[0] = fmaf(R[2], v[2], fmaf(R[1], v[1], R[0] * v[0]));
    o[1] = fmaf(R[5], v[2], fmaf(R[4], v[1], R[3] * v[0]));
    o[2] = fmaf(R[8], v[2], fmaf(R[7], v[1], R[6] * v[0]));
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

// pnr_views.uv_scale_{x,y}[lvl]: 0 = 1.0 = the fork's mapping (texel coordinate = image-pixel coordinate); a
// product with exactly 1.0f returns its operand bit for bit (NaN and inf included), so the default stays the parity path.
__device__ __forceinline__ float uv_sx(const pnr_views& vw, int lvl) { const float s = vw.uv_scale_x[lvl]; return s != 0.f ? s : 1.0f; }
__device__ __forceinline__ float uv_sy(const pnr_views& vw, int lvl) { const float s = vw.uv_scale_y[lvl]; return s != 0.f ? s : 1.0f; }

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

// ---------------------------------------------------------------- per-ray stages as device functions
// (shared by the stage kernels of stage_kernels.hip and by the fused render launch of point_mfma.hip)
// sample_coarse (nerf.py:98-118): z = near(1-t)+far*t  (or 1/((1-t)/near + t/far)),  t = linspace(0,1-1/Kc,Kc)[k] + U*(1/Kc)
__device__ __forceinline__ float z_from_t(float t, float near, float far, int lindisp) {
    if (!lindisp) return fmaf(far, t, near * (1.0f - t));
    return 1.0f / fmaf(1.0f / far, t, (1.0f / near) * (1.0f - t));
}
__device__ __forceinline__ float linspace_k(int k, int n) {
    // torch.linspace(0, 1-step, n): start + k*(end-start)/(n-1), mirrored from the end in the upper half
    float step = 1.0f / (float)n;
    float end = 1.0f - step;
    if (n == 1) return 0.0f;
    float inc = end / (float)(n - 1);
    return (k < n / 2) ? inc * (float)k : fmaf(-inc, (float)(n - 1 - k), end);
}

// gen_rays (util.py:118-148,243-281): pinhole camera looking down -z, pixel -> world ray
struct RayCam { float R[9]; float o[3]; float fx, fy, cx, cy, zn, zf; int W, H; };
inline RayCam make_ray_cam(const float* c2w /* host, row-major 4x4 */, int W, int H, float fx, float fy, float cx, float cy,
                           float z_near, float z_far) {
    RayCam c;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c.R[i * 3 + j] = c2w[i * 4 + j];
        c.o[i] = c2w[i * 4 + 3];
    }
    c.fx = fx; c.fy = fy; c.cx = cx; c.cy = cy; c.zn = z_near; c.zf = z_far; c.W = W; c.H = H;
    return c;
}
__device__ __forceinline__ void pinhole_ray(const RayCam& c, int pix, float* d) {
    float y = (float)(pix / c.W), x = (float)(pix % c.W);
    float X = (x - c.cx) / c.fx, Y = (y - c.cy) / c.fy;
    float nrm = sqrtf(fmaf(X, X, fmaf(Y, Y, 1.0f)));
    float v[3] = {X / nrm, -Y / nrm, -1.0f / nrm};
    rot3(c.R, v, d);
}

// Loads that another wave of the SAME workgroup may have just stored (the fused launch composites what its own waves wrote a
// moment ago; a 128-byte line can hold samples of a ray composited earlier, whose stale copy would still sit in the CU's L1):
// NT = true bypasses the L1 (MI355X_MICROARCH.md, visibility table: nt / sc1 loads are L2-served).
template <bool NT> __device__ __forceinline__ float ld_f(const float* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ float4 ld_f4(const float4* p) {
    if (!NT) return *p;
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v v = __builtin_nontemporal_load((const f4v*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}

// composite (nerf.py:178-182,223-249), one WAVE per ray: alpha = 1-exp(-delta*relu(sigma)); T = exclusive
// cumprod(1-alpha+1e-10); w = alpha*T; rgb = sum w c (+ 1 - sum w if white background); depth = sum w z.  The transmittance
// product is a wave-wide multiplicative scan per 64-sample segment with a carry.  Returns (rgb, depth) in lane 0.
// LZ(k) -> z_k, LC(k) -> (r, g, b, sigma)_k of this ray: global memory (stage kernel, the fused launch's memory route) or the
// workgroup's LDS ring (the fused launch's on-chip route) — ONE body, so every route is the same arithmetic in the same order.
template <typename LZ, typename LC>
__device__ __forceinline__ float4 composite_ray_t(LZ lz, LC lc, int K, float far, int white_bkgd, float* wr /* or null */, int lane) {
    float carry = 1.0f;
    float ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f, aw = 0.f;
    // the next 64-sample segment's loads are issued before this segment's scan (same arithmetic, one round trip hidden)
    // z_{k+1} is the neighbouring lane's z_k (one DPP wave shift, no second load); lane 63's neighbour is lane 0 of the next segment
    int k = lane;
    bool act = k < K;
    float zk = act ? lz(k) : 0.f;
    float4 c = act ? lc(k) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k2 = k0 + 64 + lane;
        const bool act2 = k2 < K;
        float zk2 = 0.f;
        float4 c2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k0 + 64 < K) {
            zk2 = act2 ? lz(k2) : 0.f;
            c2 = act2 ? lc(k2) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float zn = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(zk), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
        const float z_next_seg = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zk2), 0));
        if (lane == 63) zn = z_next_seg;
        if (!(k + 1 < K)) zn = far;                             // delta_K = far - z_K
        float delta = zn - zk;
        float alpha = act ? 1.0f - expf(-delta * fmaxf(c.w, 0.0f)) : 0.0f;
        float tr = act ? (1.0f - alpha) + 1e-10f : 1.0f;
        float incl = wave_scan_mul_dpp(tr);
        float excl = dpp_f<0x138 /* wave_shr:1 */, 0xf>(1.0f, incl);       // lane 0 has no source: 1
        float w = alpha * (carry * excl);
        carry *= __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), 63));
        if (act && wr) wr[k] = w;
        ar = fmaf(w, c.x, ar); ag = fmaf(w, c.y, ag); ab = fmaf(w, c.z, ab); ad = fmaf(w, zk, ad); aw += w;
        k = k2; act = act2; zk = zk2; c = c2;
    }
    ar = wave_sum_dpp(ar); ag = wave_sum_dpp(ag); ab = wave_sum_dpp(ab); ad = wave_sum_dpp(ad);
    aw = wave_sum_dpp(aw);
    if (white_bkgd) { float bg = 1.0f - aw; ar = ar + bg; ag = ag + bg; ab = ab + bg; }
    return make_float4(ar, ag, ab, ad);
}
template <bool NT>
__device__ __forceinline__ float4 composite_ray(const float* zr, const float4* cr, int K, float far, int white_bkgd,
                                                float* wr /* or null */, int lane) {
    return composite_ray_t([zr](int k) { return ld_f<NT>(zr + k); }, [cr](int k) { return ld_f4<NT>(cr + k); }, K, far, white_bkgd, wr, lane);
}

// sample_fine + sample_fine_depth + cat + sort (nerf.py:120-161,285-295), one WAVE per ray with wave-private LDS scratch
// `cdf` (Kc+2 floats) and `buf` (P2 = pow2 >= Kc+n_imp+n_dep floats):
//   cdf[0..Kc]  = [0, cumsum((w+1e-5)/sum(w+1e-5))]
//   importance: i = #(cdf <= u) - 1 clamped at 0 (NO upper clamp), t = (i + r)/Kc -> z
//   depth:      z = clamp(depth + g*depth_std, near, far)
//   merged with z_coarse and sorted ascending (bitonic network, +inf padding).
// SYNC() orders the wave's LDS accesses between the phases (__syncthreads in the stage kernel; a wave-level fence in the fused one).
struct FineArgs { int Kc, n_imp, n_dep, P2, lindisp; float depth_std; const float* un; const float* rn; const float* gn; uint64_t seed; };
template <bool NT, typename Sync>
__device__ __forceinline__ void sample_fine_ray(const FineArgs& f, const float* zc /* this ray */, const float* w /* this ray */,
                                                float depth, float near, float far, int64_t ray_local, int64_t ray_global,
                                                bool live, float* cdf, float* buf, float* z_out /* this ray */, int lane,
                                                Sync sync) {
    const int Kc = f.Kc, n_imp = f.n_imp, n_dep = f.n_dep, P2 = f.P2, Kt = Kc + n_imp + n_dep;
    float carry = 0.f;
    if (live && n_imp > 0) {
        float s = 0.f;
        for (int k = lane; k < Kc; k += 64) s += ld_f<NT>(w + k) + 1e-5f;
        s = wave_sum(s);
        if (lane == 0) cdf[0] = 0.f;
        for (int k0 = 0; k0 < Kc; k0 += 64) {
            int k = k0 + lane;
            float p = (k < Kc) ? (ld_f<NT>(w + k) + 1e-5f) / s : 0.f;
            float inc = wave_scan_add(p, lane) + carry;
            if (k < Kc) cdf[k + 1] = inc;
            carry = __shfl(inc, 63, 64);
        }
    }
    for (int k = lane; k < P2; k += 64) buf[k] = (live && k < Kc) ? ld_f<NT>(zc + k) : __builtin_inff();
    sync();
    if (live) {
        for (int j = lane; j < n_imp; j += 64) {
            float u = f.un ? f.un[ray_local * n_imp + j] : rng_uniform(f.seed, ray_global, DRAW_U, j);
            float r = f.rn ? f.rn[ray_local * n_imp + j] : rng_uniform(f.seed, ray_global, DRAW_R, j);
            int cnt = 0;
            for (int k = 0; k <= Kc; ++k) cnt += (cdf[k] <= u) ? 1 : 0;     // searchsorted(right=True)
            float ind = fmaxf((float)cnt - 1.0f, 0.0f);
            float t = (ind + r) / (float)Kc;
            buf[Kc + j] = z_from_t(t, near, far, f.lindisp);
        }
        for (int j = lane; j < n_dep; j += 64) {
            float g = f.gn ? f.gn[ray_local * n_dep + j] : rng_normal(f.seed, ray_global, DRAW_G, j);
            float zz = fmaf(g, f.depth_std, depth);
            buf[Kc + n_imp + j] = fmaxf(fminf(zz, far), near);
        }
    }
    sync();
    for (int sz = 2; sz <= P2; sz <<= 1) {
        for (int st = sz >> 1; st > 0; st >>= 1) {
            for (int i = lane; i < (P2 >> 1); i += 64) {
                int lo = ((i / st) * (st << 1)) + (i % st);
                int hi = lo + st;
                bool asc = ((lo & sz) == 0);
                float a = buf[lo], b = buf[hi];
                bool sw = asc ? (a > b) : (a < b);
                if (sw) { buf[lo] = b; buf[hi] = a; }
            }
            sync();
        }
    }
    if (live)
        for (int k = lane; k < Kt; k += 64) z_out[k] = buf[k];
}

// The per-ray work a point launch of the MFMA kernel also does (the fused render launch of pnr_render, point_mfma.hip):
// workgroups own whole rays; sample positions are generated and the finished rays composited inside the launch.
struct RayJob {
    int on;                         // 0: plain point launch (pnr_point_mlp)
    int K;                          // samples per ray of this pass
    int gen_z;                      // 1: coarse positions generated in the kernel (and written to z_out); 0: PointSrc.z
    int lindisp, white_bkgd;
    int from_cam;                   // 1: ray r is pixel pix0 + r of `cam` (no ray tensor)
    int rays_per_wg;
    // cmp_lds (set by point_mfma): a finished ray is composited from the workgroup's LDS ring — the tile epilogue leaves the
    // point's (rgb, sigma) and z there instead of in global memory (K <= CMP_MAX_K, ring of CMP_RING points).  z_needed: the
    // coarse positions are also wanted in z_out (a fine pass or the caller reads them); without cmp_lds z_out is always written.
    int cmp_lds, z_needed;
    int64_t n_rays;
    const float* noise_c; uint64_t seed; RayKey key;
    float* z_out; float* w_out; float* rgb_out; float* depth_out;      // w_out may be NULL
    int rgb_stride, depth_stride, w_stride;                            // floats per ray (pnr_outputs strides; dense: 3, 1, K)
    RayCam cam; int pix0;
    // fine pass: the launch first resamples its own rays (sample_fine .. sort, nerf.py:120-161,285-295) from the coarse pass's
    // outputs into z_fine (= PointSrc.z of this launch), every workgroup for the rays it owns
    int resample;
    FineArgs fine;
    const float* zc; const float* wc; const float* depth_c;
    int wc_stride, dc_stride;       // floats per ray of the coarse weights / depth the resampling reads (they may be strided outputs)
    float* z_fine;
};

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
        p[0] = fmaf(z, d[0], r[0]); p[1] = fmaf(z, d[1], r[1]); p[2] = fmaf(z, d[2], r[2]);
    } else {
        p[0] = s.xyz[g * 3 + 0]; p[1] = s.xyz[g * 3 + 1]; p[2] = s.xyz[g * 3 + 2];
        d[0] = s.dirs[g * 3 + 0]; d[1] = s.dirs[g * 3 + 1]; d[2] = s.dirs[g * 3 + 2];
    }
}

}  // namespace pnr
