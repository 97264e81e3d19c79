// fp32 kernels shared by the inference-precision path (point_f32.hip) and the training path
// (train_f32.hip).  Included into both translation units; every kernel is static.
#pragma once
#include "pnr_common.h"

namespace pnr {

// Channels-last fp32 copies of the latent maps, [view][texel][C] per level (round 4): the feature build's lanes walk the
// channels, and in the reference's NCHW layout that is a stride of H*W floats — 64 different cache lines per load
// instruction (k_features_f32 ran at 0.4-0.5 TB/s, bound by the texture addresser).  With the channel innermost a tap is one
// 256-byte row.  p[0] == NULL: none (the NCHW reads stay).  Built once per call into the caller's workspace / tape.
struct LatCL { const float* p[PNR_MAX_LEVELS]; };
static inline uint64_t latent_cl_bytes(const pnr_views& vw) {
    uint64_t n = 0;
    for (int l = 0; l < vw.n_levels; ++l)
        n += (((uint64_t)vw.n_objs * vw.n_views * vw.lat_c[l] * vw.lat_h[l] * vw.lat_w[l] * 4) + 255) & ~(uint64_t)255;
    return n;
}
static __global__ void k_latent_to_cl(const float* __restrict__ src, int64_t n_views, int C, int T, float* __restrict__ dst) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // over (view, texel, channel): coalesced stores
    if (e >= n_views * C * T) return;
    const int c = (int)(e % C);
    const int64_t r = e / C;
    const int t = (int)(r % T);
    const int64_t v = r / T;
    dst[e] = src[(v * C + c) * T + t];
}
static inline LatCL latent_cl_build(const pnr_views& vw, void* base, hipStream_t s) {
    LatCL cl{};
    uint8_t* p = (uint8_t*)base;
    for (int l = 0; l < vw.n_levels; ++l) {
        if (!vw.latent[l]) return LatCL{};
        const int64_t nv = (int64_t)vw.n_objs * vw.n_views, n = nv * vw.lat_c[l] * vw.lat_h[l] * vw.lat_w[l];
        hipLaunchKernelGGL(k_latent_to_cl, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, vw.latent[l], nv, vw.lat_c[l],
                           vw.lat_h[l] * vw.lat_w[l], (float*)p);
        cl.p[l] = (const float*)p;
        p += ((uint64_t)n * 4 + 255) & ~(uint64_t)255;
    }
    return cl;
}

// zx[col][0..L) = bilinear latent sample, zx[col][L..L+d_in) = pos-enc(x_rot) ++ R*viewdir
// col = v*CH + pl for point g0+pl of the chunk and source view v (view index obj*NS+v).
// One wave per (view, point) column: the geometry (point, camera, rotation, projection, the four taps of every level) is formed
// once per wave — the element-per-thread form redid it (and a 64-bit div / mod) for each of the ~300 row elements — and the lanes
// walk the row, 64 consecutive elements per step (coalesced stores).  Same arithmetic per element: bit-identical rows.
static __global__ void __launch_bounds__(256) k_features_f32(pnr_views vw, PointSrc src, int64_t g0, int CH, int64_t pts_per_obj,
                               int L, int d_in, int use_code_viewdirs, int num_freqs, float freq_factor,
                               float* __restrict__ zx, int ldz /* row stride of zx, >= L + d_in */, LatCL cl) {
    const int E = L + d_in;
    const int lane = threadIdx.x & 63;
    const int64_t col = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= (int64_t)CH * vw.n_views) return;
    const int v = (int)(col / CH);
    const int pl = (int)(col - (int64_t)v * CH);
    const int64_t g = g0 + pl;
    const int obj = (int)(g / pts_per_obj);
    const int view = obj * vw.n_views + v;
    float p[3], d[3], xr[3], dr[3];
    fetch_point(src, g, p, d);
    const Cam cam = load_cam(vw, view);
    rot3(cam.R, p, xr);
    rot3(cam.R, d, dr);
    float u, w;
    project(cam, xr, u, w);
    float* row = zx + (size_t)col * ldz;
    // latent part, level by level (the taps are per level)
    int e0 = 0;
    for (int lvl = 0; lvl < vw.n_levels; ++lvl) {
        const int W = vw.lat_w[lvl], H = vw.lat_h[lvl], C = vw.lat_c[lvl];
        const Taps t = bilinear_taps(u * uv_sx(vw, lvl), w * uv_sy(vw, lvl), W, H);
        if (cl.p[lvl]) {          // channels-last copy: a tap is C contiguous floats (same products, same order)
            const float* base = cl.p[lvl] + (size_t)view * (size_t)(H * W) * C;
            for (int ch = lane; ch < C; ch += 64) {
                float val = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) val += base[(size_t)t.off[i] * C + ch] * t.w[i];
                row[e0 + ch] = val;
            }
        } else {
            const float* base = vw.latent[lvl] + (size_t)view * C * (size_t)(H * W);
            for (int ch = lane; ch < C; ch += 64) {
                const float* bc = base + (size_t)ch * (size_t)(H * W);
                float val = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) val += bc[t.off[i]] * t.w[i];
                row[e0 + ch] = val;
            }
        }
        e0 += C;
    }
    // code part
    const float x6[6] = {xr[0], xr[1], xr[2], dr[0], dr[1], dr[2]};
    const int dcode = 3 + 6 * num_freqs;
    for (int j = lane; j < E - L; j += 64) {
        float val;
        if (use_code_viewdirs) val = posenc_elem(x6, 6, j, freq_factor);
        else val = (j < dcode) ? posenc_elem(xr, 3, j, freq_factor) : dr[j - dcode];
        row[L + j] = val;
    }
}

// launch: 4 columns (waves) per block
// The kernel writes sum(lat_c) latent columns per row whatever L says: the MLP's d_latent must BE that sum (an MLP without a
// latent input, or coarse / fine MLPs of different d_latent, against encoded maps would overrun into the code columns).
static inline bool latent_width_matches(const pnr_views& vw, int L) {
    int sum = 0;
    for (int i = 0; i < vw.n_levels; ++i) sum += vw.lat_c[i];
    return sum == L;
}
static inline int32_t features_launch(const pnr_views& vw, const PointSrc& src, int64_t g0, int CH, int64_t pts_per_obj, int L, int d_in,
                                      int use_code_viewdirs, int num_freqs, float freq_factor, float* zx, int ldz, hipStream_t s,
                                      LatCL cl = LatCL{}) {
    if (!latent_width_matches(vw, L)) return PNR_E_SHAPE;
    const int64_t cols = (int64_t)CH * vw.n_views;
    if (cols == 0) return PNR_OK;
    hipLaunchKernelGGL(k_features_f32, dim3((unsigned)((cols + 3) / 4)), dim3(256), 0, s, vw, src, g0, CH, pts_per_obj, L, d_in,
                       use_code_viewdirs, num_freqs, freq_factor, zx, ldz, cl);
    return PNR_OK;
}

// x (NS, CH, H) -> (CH, H): mean or max over the view axis (util.combine_interleaved, util.py:466-476)
static __global__ void k_combine_f32(const float* __restrict__ x, int NS, int64_t per_view, int combine_type,
                              float* __restrict__ y) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_view) return;
    float a = x[i];
    if (combine_type == PNR_COMBINE_MAX) {
        for (int v = 1; v < NS; ++v) a = fmaxf(a, x[i + v * per_view]);
    } else {
        for (int v = 1; v < NS; ++v) a += x[i + v * per_view];
        a = a / (float)NS;
    }
    y[i] = a;
}

// sigmoid(rgb), relu(sigma)  (models.py.backup2:276-281)
static __global__ void k_out_act(const float* __restrict__ o, int64_t n, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = ((const float4*)o)[i];
    v.x = 1.0f / (1.0f + expf(-v.x));
    v.y = 1.0f / (1.0f + expf(-v.y));
    v.z = 1.0f / (1.0f + expf(-v.z));
    v.w = fmaxf(v.w, 0.f);
    ((float4*)out)[i] = v;
}

}  // namespace pnr
