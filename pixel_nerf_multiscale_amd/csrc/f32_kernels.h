// fp32 kernels shared by the inference-precision path (point_f32.hip) and the training path
// (train_f32.hip).  Included into both translation units; every kernel is static.
#pragma once
#include "pnr_common.h"

namespace pnr {

// zx[col][0..L) = bilinear latent sample, zx[col][L..L+d_in) = pos-enc(x_rot) ++ R*viewdir
// col = v*CH + pl for point g0+pl of the chunk and source view v (view index obj*NS+v).
static __global__ void k_features_f32(pnr_views vw, PointSrc src, int64_t g0, int CH, int64_t pts_per_obj,
                               int L, int d_in, int use_code_viewdirs, int num_freqs, float freq_factor,
                               float* __restrict__ zx, int ldz /* row stride of zx, >= L + d_in */) {
    const int E = L + d_in;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)CH * vw.n_views * E;
    if (idx >= total) return;
    int e = (int)(idx % E);
    int64_t col = idx / E;
    int v = (int)(col / CH);
    int pl = (int)(col % CH);
    int64_t g = g0 + pl;
    int obj = (int)(g / pts_per_obj);
    int view = obj * vw.n_views + v;
    float p[3], d[3], xr[3];
    fetch_point(src, g, p, d);
    Cam cam = load_cam(vw, view);
    rot3(cam.R, p, xr);
    float val;
    if (e < L) {
        float u, w;
        project(cam, xr, u, w);
        int lvl = 0, ch = e;
        while (ch >= vw.lat_c[lvl]) { ch -= vw.lat_c[lvl]; ++lvl; }
        int W = vw.lat_w[lvl], H = vw.lat_h[lvl], C = vw.lat_c[lvl];
        Taps t = bilinear_taps(u, w, W, H);
        const float* base = vw.latent[lvl] + ((size_t)view * C + ch) * (size_t)(H * W);
        val = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) val += base[t.off[i]] * t.w[i];
    } else {
        int j = e - L;
        float dr[3];
        rot3(cam.R, d, dr);
        if (use_code_viewdirs) {
            float x6[6] = {xr[0], xr[1], xr[2], dr[0], dr[1], dr[2]};
            val = posenc_elem(x6, 6, j, freq_factor);
        } else {
            int dcode = 3 + 6 * num_freqs;
            val = (j < dcode) ? posenc_elem(xr, 3, j, freq_factor) : dr[j - dcode];
        }
    }
    zx[(size_t)col * ldz + e] = val;
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
