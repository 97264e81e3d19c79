// Per-ray sampling / compositing stages of NeRFRenderer (reference render/nerf.py).  HBM-bound,
// ray-major: one wavefront (64 lanes) per ray, lanes stride the samples so every global access is a
// coalesced 256-B (float) or 1-KiB (float4) row segment.
#include "pnr_common.h"

namespace pnr {

// ------------------------------------------------------------------ sample_coarse  (nerf.py:98-118)
// z[i,k] = near(1-t)+far*t  (or 1/((1-t)/near + t/far)),  t = linspace(0,1-1/Kc,Kc)[k] + U*(1/Kc)
__device__ __forceinline__ float z_from_t(float t, float near, float far, int lindisp) {
    if (!lindisp) return near * (1.0f - t) + far * t;
    return 1.0f / (1.0f / near * (1.0f - t) + 1.0f / far * t);
}

__device__ __forceinline__ float linspace_k(int k, int n) {
    // torch.linspace(0, 1-step, n): start + k*(end-start)/(n-1), mirrored from the end in the upper half
    float step = 1.0f / (float)n;
    float end = 1.0f - step;
    if (n == 1) return 0.0f;
    float inc = end / (float)(n - 1);
    return (k < n / 2) ? inc * (float)k : end - inc * (float)(n - 1 - k);
}

__global__ void k_sample_coarse(const float* __restrict__ rays, int64_t n_rays, int Kc, int lindisp,
                                const float* __restrict__ noise, uint64_t seed, int64_t ray_base,
                                float* __restrict__ z_out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rays * Kc) return;
    int64_t ray = idx / Kc;
    int k = (int)(idx % Kc);
    float near = rays[ray * 8 + 6], far = rays[ray * 8 + 7];
    float u = noise ? noise[idx] : rng_uniform(seed, ray_base + ray, DRAW_COARSE, k);
    float t = linspace_k(k, Kc) + u * (1.0f / (float)Kc);
    z_out[idx] = z_from_t(t, near, far, lindisp);
}

// ------------------------------------------------------------------ composite  (nerf.py:178-182,223-249)
// One wave per ray.  alpha = 1-exp(-delta*relu(sigma)); T = exclusive cumprod(1-alpha+1e-10);
// w = alpha*T; rgb = sum w c (+ 1 - sum w if white background); depth = sum w z.
// The transmittance product is a wave-wide multiplicative scan per 64-sample segment with a carry.
__global__ void __launch_bounds__(256) k_composite(const float* __restrict__ rays, const float* __restrict__ z,
                                                   const float4* __restrict__ rgbs, int64_t n_rays, int K,
                                                   int white_bkgd, float* __restrict__ w_out,
                                                   float* __restrict__ rgb_out, float* __restrict__ depth_out) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;                      // whole wave exits together
    const float far = rays[ray * 8 + 7];
    const float* zr = z + ray * K;
    const float4* cr = rgbs + ray * K;
    float carry = 1.0f;
    float ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f, aw = 0.f;
    for (int k0 = 0; k0 < K; k0 += 64) {
        int k = k0 + lane;
        bool act = k < K;
        float zk = act ? zr[k] : 0.f;
        float zn = (k + 1 < K) ? zr[k + 1] : far;   // delta_K = far - z_K
        float4 c = act ? cr[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        float delta = zn - zk;
        float alpha = act ? 1.0f - expf(-delta * fmaxf(c.w, 0.0f)) : 0.0f;
        float tr = act ? (1.0f - alpha) + 1e-10f : 1.0f;
        float incl = wave_scan_mul(tr, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        float w = alpha * (carry * excl);
        carry *= __shfl(incl, 63, 64);
        if (act && w_out) w_out[ray * K + k] = w;
        ar += w * c.x; ag += w * c.y; ab += w * c.z; ad += w * zk; aw += w;
    }
    ar = wave_sum(ar); ag = wave_sum(ag); ab = wave_sum(ab); ad = wave_sum(ad); aw = wave_sum(aw);
    if (lane == 0) {
        if (white_bkgd) { float bg = 1.0f - aw; ar = ar + bg; ag = ag + bg; ab = ab + bg; }
        rgb_out[ray * 3 + 0] = ar; rgb_out[ray * 3 + 1] = ag; rgb_out[ray * 3 + 2] = ab;
        depth_out[ray] = ad;
    }
}

// ------------------------------------------------------------------ sample_fine + sample_fine_depth + cat + sort
// (nerf.py:120-161,285-295).  One wave per ray, wave-private LDS:
//   cdf[0..Kc]  = [0, cumsum((w+1e-5)/sum(w+1e-5))]
//   importance: i = #(cdf <= u) - 1 clamped at 0 (NO upper clamp), t = (i + r)/Kc -> z
//   depth:      z = clamp(depth + g*depth_std, near, far)
//   merged with z_coarse and sorted ascending (bitonic network over the next power of two, +inf padding).
__global__ void __launch_bounds__(256) k_sample_fine(
    const float* __restrict__ rays, const float* __restrict__ zc, const float* __restrict__ weights,
    const float* __restrict__ depth, int64_t n_rays, int Kc, int n_imp, int n_dep, float depth_std, int lindisp,
    const float* __restrict__ un, const float* __restrict__ rn, const float* __restrict__ gn,
    uint64_t seed, int64_t ray_base, float* __restrict__ z_out, int P2 /* pow2 >= Kc+n_imp+n_dep */) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    const bool live = ray < n_rays;                  // wave-uniform; no early return (barriers below)
    const int Kt = Kc + n_imp + n_dep;
    float* cdf = smem + (size_t)wv * (P2 + Kc + 2);  // Kc+1 entries
    float* buf = cdf + Kc + 2;                       // P2 entries
    float near = 0.f, far = 0.f;
    if (live) { near = rays[ray * 8 + 6]; far = rays[ray * 8 + 7]; }

    // --- cdf
    float carry = 0.f;
    if (live && n_imp > 0) {
        float s = 0.f;
        for (int k = lane; k < Kc; k += 64) s += weights[ray * Kc + k] + 1e-5f;
        s = wave_sum(s);
        if (lane == 0) cdf[0] = 0.f;
        for (int k0 = 0; k0 < Kc; k0 += 64) {
            int k = k0 + lane;
            float p = (k < Kc) ? (weights[ray * Kc + k] + 1e-5f) / s : 0.f;
            float inc = wave_scan_add(p, lane) + carry;
            if (k < Kc) cdf[k + 1] = inc;
            carry = __shfl(inc, 63, 64);
        }
    }
    // --- coarse samples + padding
    for (int k = lane; k < P2; k += 64) buf[k] = (live && k < Kc) ? zc[ray * Kc + k] : __builtin_inff();
    __syncthreads();
    if (live) {
        // --- importance samples
        for (int j = lane; j < n_imp; j += 64) {
            float u = un ? un[ray * n_imp + j] : rng_uniform(seed, ray_base + ray, DRAW_U, j);
            float r = rn ? rn[ray * n_imp + j] : rng_uniform(seed, ray_base + ray, DRAW_R, j);
            int cnt = 0;
            for (int k = 0; k <= Kc; ++k) cnt += (cdf[k] <= u) ? 1 : 0;     // searchsorted(right=True)
            float ind = fmaxf((float)cnt - 1.0f, 0.0f);
            float t = (ind + r) / (float)Kc;
            buf[Kc + j] = z_from_t(t, near, far, lindisp);
        }
        // --- depth samples
        float dpt = n_dep > 0 ? depth[ray] : 0.f;
        for (int j = lane; j < n_dep; j += 64) {
            float g = gn ? gn[ray * n_dep + j] : rng_normal(seed, ray_base + ray, DRAW_G, j);
            float zz = dpt + g * depth_std;
            buf[Kc + n_imp + j] = fmaxf(fminf(zz, far), near);
        }
    }
    __syncthreads();
    // --- bitonic sort of buf[0..P2) (ascending); one wave per array, barriers keep LDS ordered
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
            __syncthreads();
        }
    }
    if (live)
        for (int k = lane; k < Kt; k += 64) z_out[ray * Kt + k] = buf[k];
}

// ------------------------------------------------------------------ gen_rays  (util.py:118-148,243-281)
struct RayCam { float R[9]; float o[3]; float fx, fy, cx, cy, zn, zf; int W, H; };
__global__ void k_gen_rays(RayCam c, int64_t pix0, int64_t n, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t pix = pix0 + i;
    float y = (float)(pix / c.W), x = (float)(pix % c.W);
    float X = (x - c.cx) / c.fx, Y = (y - c.cy) / c.fy;
    float v[3] = {X, -Y, -1.0f};
    float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] /= nrm; v[1] /= nrm; v[2] /= nrm;
    float d[3];
    rot3(c.R, v, d);
    float* o = out + i * 8;
    o[0] = c.o[0]; o[1] = c.o[1]; o[2] = c.o[2];
    o[3] = d[0]; o[4] = d[1]; o[5] = d[2];
    o[6] = c.zn; o[7] = c.zf;
}

}  // namespace pnr

using namespace pnr;

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

extern "C" int32_t pnr_sample_coarse(const float* rays, int64_t n_rays, int32_t n_coarse, int32_t lindisp,
                                     const float* noise_c, uint64_t seed, int64_t ray_index_base,
                                     float* z_out, void* stream) {
    if (!rays || !z_out) return PNR_E_NULL;
    if (n_rays < 0 || n_coarse <= 0) return PNR_E_SHAPE;
    if (n_rays == 0) return PNR_OK;
    int64_t tot = n_rays * n_coarse;
    hipLaunchKernelGGL(k_sample_coarse, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rays, n_rays, n_coarse, lindisp, noise_c, seed, ray_index_base, z_out);
    PNR_LAUNCH_CHECK();
    return PNR_OK;
}

extern "C" int32_t pnr_composite(const float* rays, const float* z, const float* rgbsigma, int64_t n_rays,
                                 int32_t K, int32_t white_bkgd, float* weights_out, float* rgb_out,
                                 float* depth_out, void* stream) {
    if (!rays || !z || !rgbsigma || !rgb_out || !depth_out) return PNR_E_NULL;
    if (n_rays < 0 || K <= 0) return PNR_E_SHAPE;
    if (((uintptr_t)rgbsigma & 15) != 0) return PNR_E_ALIGN;
    if (n_rays == 0) return PNR_OK;
    hipLaunchKernelGGL(k_composite, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       rays, z, (const float4*)rgbsigma, n_rays, K, white_bkgd, weights_out, rgb_out, depth_out);
    PNR_LAUNCH_CHECK();
    return PNR_OK;
}

extern "C" int32_t pnr_sample_fine(const float* rays, const float* z_coarse, const float* weights,
                                   const float* depth, int64_t n_rays, int32_t n_coarse, int32_t n_fine,
                                   int32_t n_fine_depth, float depth_std, int32_t lindisp, const float* u,
                                   const float* r, const float* g, uint64_t seed, int64_t ray_index_base,
                                   float* z_out, void* stream) {
    if (!rays || !z_coarse || !z_out) return PNR_E_NULL;
    if (n_rays < 0 || n_coarse <= 0 || n_fine < 0 || n_fine_depth < 0 || n_fine_depth > n_fine) return PNR_E_SHAPE;
    int n_imp = n_fine - n_fine_depth;
    if (n_imp > 0 && !weights) return PNR_E_NULL;
    if (n_fine_depth > 0 && !depth) return PNR_E_NULL;
    int Kt = n_coarse + n_fine;
    int P2 = next_pow2(Kt);
    if (P2 > 4096) return PNR_E_UNSUPPORTED;
    if (n_rays == 0) return PNR_OK;
    size_t lds = (size_t)4 * (P2 + n_coarse + 2) * sizeof(float);
    if (lds > 160 * 1024) return PNR_E_UNSUPPORTED;
    if (lds > 64 * 1024)
        PNR_HIP_CHECK(hipFuncSetAttribute((const void*)k_sample_fine, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_sample_fine, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), lds, (hipStream_t)stream,
                       rays, z_coarse, weights, depth, n_rays, n_coarse, n_imp, n_fine_depth, depth_std, lindisp,
                       u, r, g, seed, ray_index_base, z_out, P2);
    PNR_LAUNCH_CHECK();
    return PNR_OK;
}

extern "C" int32_t pnr_gen_rays(const float* c2w, int32_t W, int32_t H, float fx, float fy, float cx, float cy,
                                float z_near, float z_far, int64_t pix0, int64_t n, float* rays_out, void* stream) {
    if (!c2w || !rays_out) return PNR_E_NULL;
    if (W <= 0 || H <= 0 || n < 0 || pix0 < 0 || pix0 + n > (int64_t)W * H) return PNR_E_SHAPE;
    if (n == 0) return PNR_OK;
    RayCam c;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c.R[i * 3 + j] = c2w[i * 4 + j];
        c.o[i] = c2w[i * 4 + 3];
    }
    c.fx = fx; c.fy = fy; c.cx = cx; c.cy = cy; c.zn = z_near; c.zf = z_far; c.W = W; c.H = H;
    hipLaunchKernelGGL(k_gen_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c, pix0, n, rays_out);
    PNR_LAUNCH_CHECK();
    return PNR_OK;
}
