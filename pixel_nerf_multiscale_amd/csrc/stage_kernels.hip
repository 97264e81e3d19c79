// Per-ray sampling / compositing stages of NeRFRenderer (reference render/nerf.py).  HBM-bound,
// ray-major: one wavefront (64 lanes) per ray, lanes stride the samples so every global access is a
// coalesced 256-B (float) or 1-KiB (float4) row segment.
#include "pnr_common.h"

namespace pnr {

// ------------------------------------------------------------------ sample_coarse  (nerf.py:98-118): z_from_t / linspace_k in pnr_common.h
// One thread = four consecutive samples of a ray: one Philox block (rng_block), one 16-byte store where the row is
// aligned — the stage is bound by its 4K B/ray of output once the generator is amortised.
__global__ void k_sample_coarse(const float* __restrict__ rays, int64_t n_rays, int Kc, int lindisp,
                                const float* __restrict__ noise, uint64_t seed, RayKey key,
                                float* __restrict__ z_out) {
    const int q_per_ray = (Kc + 3) >> 2;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tot = n_rays * q_per_ray;
    if (idx >= tot) return;
    // (ray, quad) of the thread: the kernel is VALU-bound (Philox + the sample arithmetic), and a 64-bit division and
    // remainder per thread were a quarter of its instructions — shift for a power of two, 32-bit division below 2^31 threads
    int64_t ray;
    int kq;
    if ((q_per_ray & (q_per_ray - 1)) == 0) {
        const int sh = 31 - __builtin_clz((unsigned)q_per_ray);
        ray = idx >> sh;
        kq = (int)(idx & (q_per_ray - 1));
    } else if (tot < 0x7fffffffLL) {
        const uint32_t r32 = (uint32_t)idx / (uint32_t)q_per_ray;
        ray = r32;
        kq = (int)((uint32_t)idx - r32 * (uint32_t)q_per_ray);
    } else {
        ray = idx / q_per_ray;
        kq = (int)(idx - ray * q_per_ray);
    }
    const int k0 = kq * 4;
    const float near = rays[ray * 8 + 6], far = rays[ray * 8 + 7];
    float u[4];
    if (noise) {
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = (k0 + j < Kc) ? noise[ray * Kc + k0 + j] : 0.f;
    } else {
        const u32x4 v = rng_block(seed, global_ray(key, ray), DRAW_COARSE, k0 >> 2);
        u[0] = u01(v.x); u[1] = u01(v.y); u[2] = u01(v.z); u[3] = u01(v.w);
    }
    float z[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float t = fmaf(u[j], 1.0f / (float)Kc, linspace_k(k0 + j < Kc ? k0 + j : Kc - 1, Kc));
        z[j] = z_from_t(t, near, far, lindisp);
    }
    float* o = z_out + ray * Kc + k0;
    if ((Kc & 3) == 0 && (((uintptr_t)z_out) & 15) == 0) {
        *(float4*)o = make_float4(z[0], z[1], z[2], z[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < Kc) o[j] = z[j];
    }
}

// ------------------------------------------------------------------ composite  (nerf.py:178-182,223-249): composite_ray in pnr_common.h
__global__ void __launch_bounds__(256) k_composite(const float* __restrict__ rays, const float* __restrict__ z,
                                                   const float4* __restrict__ rgbs, int64_t n_rays, int K,
                                                   int white_bkgd, float* __restrict__ w_out,
                                                   float* __restrict__ rgb_out, float* __restrict__ depth_out,
                                                   int w_stride, int rgb_stride, int depth_stride /* floats per ray */) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;                      // whole wave exits together
    const float4 r = composite_ray<false>(z + ray * K, rgbs + ray * K, K, rays[ray * 8 + 7], white_bkgd,
                                          w_out ? w_out + ray * w_stride : nullptr, lane);
    if (lane == 0) {
        float* po = rgb_out + ray * rgb_stride;
        po[0] = r.x; po[1] = r.y; po[2] = r.z;
        depth_out[ray * depth_stride] = r.w;
    }
}

// ------------------------------------------------------------------ sample_fine + sample_fine_depth + cat + sort
// (nerf.py:120-161,285-295): sample_fine_ray in pnr_common.h, one wave per ray, wave-private LDS.
__global__ void __launch_bounds__(256) k_sample_fine(
    const float* __restrict__ rays, const float* __restrict__ zc, const float* __restrict__ weights,
    const float* __restrict__ depth, int64_t n_rays, int Kc, int n_imp, int n_dep, float depth_std, int lindisp,
    const float* __restrict__ un, const float* __restrict__ rn, const float* __restrict__ gn,
    uint64_t seed, RayKey key, float* __restrict__ z_out, int P2 /* pow2 >= Kc+n_imp+n_dep */,
    float near_all, float far_all /* the bounds of every ray when rays == NULL (rays of one camera) */,
    int w_stride, int d_stride /* floats per ray of `weights` / `depth` */) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    const bool live = ray < n_rays;                  // wave-uniform; no early return (barriers below)
    float* cdf = smem + (size_t)wv * (P2 + Kc + 2);  // Kc+1 entries
    float* buf = cdf + Kc + 2;                       // P2 entries
    const int64_t rr = live ? ray : 0;
    const FineArgs f{Kc, n_imp, n_dep, P2, lindisp, depth_std, un, rn, gn, seed};
    sample_fine_ray<false>(f, zc + rr * Kc, weights ? weights + rr * w_stride : nullptr, (live && n_dep > 0) ? depth[rr * d_stride] : 0.f,
                           !rays ? near_all : live ? rays[rr * 8 + 6] : 0.f, !rays ? far_all : live ? rays[rr * 8 + 7] : 0.f,
                           rr, global_ray(key, rr), live, cdf, buf,
                           z_out + rr * (Kc + n_imp + n_dep), lane, [] { __syncthreads(); });
}

// ------------------------------------------------------------------ gen_rays  (util.py:118-148,243-281)
__global__ void k_gen_rays(RayCam c, int64_t pix0, int64_t n, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float d[3];
    pinhole_ray(c, (int)(pix0 + i), d);
    float* o = out + i * 8;
    o[0] = c.o[0]; o[1] = c.o[1]; o[2] = c.o[2];
    o[3] = d[0]; o[4] = d[1]; o[5] = d[2];
    o[6] = c.zn; o[7] = c.zf;
}

}  // namespace pnr

using namespace pnr;


namespace pnr {
int32_t sample_coarse_launch(const float* rays, int64_t n_rays, int32_t n_coarse, int32_t lindisp, const float* noise_c,
                             uint64_t seed, RayKey key, float* z_out, void* stream) {
    if (!rays || !z_out) return PNR_E_NULL;
    if (n_rays < 0 || n_coarse <= 0) return PNR_E_SHAPE;
    if (n_rays == 0) return PNR_OK;
    int64_t tot = n_rays * ((n_coarse + 3) / 4);
    hipLaunchKernelGGL(k_sample_coarse, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rays, n_rays, n_coarse, lindisp, noise_c, seed, key, z_out);
    PNR_LAUNCH_CHECK();
    return PNR_OK;
}

// strides in floats per ray (0 = dense)
int32_t composite_launch(const float* rays, const float* z, const float* rgbsigma, int64_t n_rays, int32_t K,
                         int32_t white_bkgd, float* weights_out, float* rgb_out, float* depth_out, int w_stride,
                         int rgb_stride, int depth_stride, void* stream) {
    if (!rays || !z || !rgbsigma || !rgb_out || !depth_out) return PNR_E_NULL;
    if (n_rays < 0 || K <= 0) return PNR_E_SHAPE;
    if (((uintptr_t)rgbsigma & 15) != 0) return PNR_E_ALIGN;
    if (n_rays == 0) return PNR_OK;
    hipLaunchKernelGGL(k_composite, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       rays, z, (const float4*)rgbsigma, n_rays, K, white_bkgd, weights_out, rgb_out, depth_out,
                       w_stride ? w_stride : K, rgb_stride ? rgb_stride : 3, depth_stride ? depth_stride : 1);
    PNR_LAUNCH_CHECK();
    return PNR_OK;
}
}  // namespace pnr

extern "C" int32_t pnr_sample_coarse(const float* rays, int64_t n_rays, int32_t n_coarse, int32_t lindisp,
                                     const float* noise_c, uint64_t seed, int64_t ray_index_base,
                                     float* z_out, void* stream) {
    return sample_coarse_launch(rays, n_rays, n_coarse, lindisp, noise_c, seed, RayKey{ray_index_base, 0, 1}, z_out, stream);
}

extern "C" int32_t pnr_composite(const float* rays, const float* z, const float* rgbsigma, int64_t n_rays,
                                 int32_t K, int32_t white_bkgd, float* weights_out, float* rgb_out,
                                 float* depth_out, void* stream) {
    return composite_launch(rays, z, rgbsigma, n_rays, K, white_bkgd, weights_out, rgb_out, depth_out, 0, 0, 0, stream);
}

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

namespace pnr {
// rays == NULL: every ray has the bounds (near_all, far_all) — the rays of one camera (pnr_render_camera)
int32_t sample_fine_launch(const float* rays, float near_all, float far_all, const float* z_coarse, const float* weights,
                           const float* depth, int64_t n_rays, int32_t n_coarse, int32_t n_fine, int32_t n_fine_depth,
                           float depth_std, int32_t lindisp, const float* u, const float* r, const float* g, uint64_t seed,
                           RayKey key, float* z_out, void* stream, int w_stride, int d_stride) {
    if (!z_coarse || !z_out) return PNR_E_NULL;
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
                       u, r, g, seed, key, z_out, P2, near_all, far_all, w_stride ? w_stride : n_coarse, d_stride ? d_stride : 1);
    PNR_LAUNCH_CHECK();
    return PNR_OK;
}
}  // namespace pnr

extern "C" int32_t pnr_sample_fine(const float* rays, const float* z_coarse, const float* weights,
                                   const float* depth, int64_t n_rays, int32_t n_coarse, int32_t n_fine,
                                   int32_t n_fine_depth, float depth_std, int32_t lindisp, const float* u,
                                   const float* r, const float* g, uint64_t seed, int64_t ray_index_base,
                                   float* z_out, void* stream) {
    if (!rays) return PNR_E_NULL;
    return sample_fine_launch(rays, 0.f, 0.f, z_coarse, weights, depth, n_rays, n_coarse, n_fine, n_fine_depth, depth_std,
                              lindisp, u, r, g, seed, RayKey{ray_index_base, 0, 1}, z_out, stream, 0, 0);
}

extern "C" int32_t pnr_gen_rays(const float* c2w, int32_t W, int32_t H, float fx, float fy, float cx, float cy,
                                float z_near, float z_far, int64_t pix0, int64_t n, float* rays_out, void* stream) {
    if (!c2w || !rays_out) return PNR_E_NULL;
    if (W <= 0 || H <= 0 || n < 0 || pix0 < 0 || pix0 + n > (int64_t)W * H || (int64_t)W * H > 0x7fffffffLL) return PNR_E_SHAPE;
    if (n == 0) return PNR_OK;
    const RayCam c = make_ray_cam(c2w, W, H, fx, fy, cx, cy, z_near, z_far);
    hipLaunchKernelGGL(k_gen_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c, pix0, n, rays_out);
    PNR_LAUNCH_CHECK();
    return PNR_OK;
}
