// fp32 reference-precision point path: PixelNeRFNet.forward (models.py.backup2:155-282) as plain fp32
// kernels (feature build -> tiled fp32 linear layers -> view combine -> output activation).  Any
// d_hidden / d_latent / d_in.  This is the 1e-4-tolerance parity path (PNR_F32); the production
// path is the fused MFMA kernel in point_mfma.hip.
#include "f32_kernels.h"

namespace pnr {

// train_f32.hip: y (+)= act(x) W^T + b on the fp32 MFMA tile kernel (v_mfma_f32_32x32x2_f32, 128 x 128 tiles)
int32_t linear_f32_mfma(const float* X, int ldx, const float* W, int ldw, const float* b, bool relu_in, bool accum, float* Y,
                        int ldy, int64_t M, int N, int K, hipStream_t s);

template <bool RELU_IN, bool ACCUM>
static int32_t linear(const float* X, int ldx, const float* W, const float* b, float* Y, int ldy, int M, int N, int K,
                      hipStream_t s) {
    return linear_f32_mfma(X, ldx, W, K, b, RELU_IN, ACCUM, Y, ldy, M, N, K, s);
}

// points per chunk: 49152 x NS rows = 384 x NS row tiles x 4 column tiles per GEMM launch — two full rounds of the 768 workgroup
// slots (3 per CU).  With 16384 a launch was 512 workgroups, 2 per CU and nothing behind them: 0.57 of the fp32 MFMA peak
// against 0.64 (24576: 0.62; 98304 / 196608: 0.64 / 0.65).  Workspace: ~5.3 KB per point and view.
static const int F32_CHUNK = 49152;

uint64_t point_f32_workspace_bytes(const pnr_mlp* mlp, const pnr_views* vw) {
    uint64_t per_pt = (uint64_t)vw->n_views * (((mlp->d_latent + mlp->d_in + 3) & ~3) + 2ull * mlp->d_hidden) + 4;
    return per_pt * F32_CHUNK * sizeof(float) + 512 + latent_cl_bytes(*vw);      // + the channels-last latent copies of a call
}

// ResnetFC.forward (resnetfc.py:203-236) on M = CH * NS assembled rows zx (row = view * CH + point, [z | x], row stride E):
// lin_in, per block [view reduction at combine_layer] + lin_z + fc_0 / fc_1, lin_out -> o (CH, d_out), no activation.
static int32_t chain_f32(const pnr_mlp* mlp, const float* zx, int E, int CH, int NS, float* x, float* h, float* o, hipStream_t s) {
    const int L = mlp->d_latent, Din = mlp->d_in, H = mlp->d_hidden;
    const int n_lin_z = mlp->combine_layer < mlp->n_blocks ? mlp->combine_layer : mlp->n_blocks;
    int M = CH * NS;
    int32_t rc;
    if ((rc = linear<false, false>(zx + L, E, mlp->lin_in_w, mlp->lin_in_b, x, H, M, H, Din, s))) return rc;
    for (int b = 0; b < mlp->n_blocks; ++b) {
        if (b == mlp->combine_layer && NS > 1) {
            int64_t per_view = (int64_t)CH * H;
            hipLaunchKernelGGL(k_combine_f32, dim3((unsigned)((per_view + 255) / 256)), dim3(256), 0, s, x, NS,
                               per_view, mlp->combine_type, x);
            PNR_LAUNCH_CHECK();
            M = CH;
        }
        if (L > 0 && b < n_lin_z)
            if ((rc = linear<false, true>(zx, E, mlp->lin_z_w[b], mlp->lin_z_b[b], x, H, M, H, L, s))) return rc;
        if ((rc = linear<true, false>(x, H, mlp->fc0_w[b], mlp->fc0_b[b], h, H, M, H, H, s))) return rc;
        if ((rc = linear<true, true>(h, H, mlp->fc1_w[b], mlp->fc1_b[b], x, H, M, H, H, s))) return rc;
    }
    if (M != CH) return PNR_E_UNSUPPORTED;   // combine_layer >= n_blocks with NS > 1: the reference never reduces
    return linear<true, false>(x, H, mlp->lin_out_w, mlp->lin_out_b, o, mlp->d_out, M, mlp->d_out, H, s);
}

int32_t point_f32(const pnr_params* prm, const pnr_mlp* mlp, const pnr_views* vw, PointSrc src, int64_t n_points,
                  int64_t pts_per_obj, float* out, void* workspace, uint64_t ws_bytes, hipStream_t s) {
    if (ws_bytes < point_f32_workspace_bytes(mlp, vw)) return PNR_E_WORKSPACE;
    if (mlp->d_out != 4) return PNR_E_SHAPE;
    const int NS = vw->n_views, L = mlp->d_latent, Din = mlp->d_in, H = mlp->d_hidden;
    const int E = (L + Din + 3) & ~3;              // row stride of zx: 16-byte rows for the GEMM's vector loads
    float* zx = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float* x = zx + (size_t)F32_CHUNK * NS * E;
    float* h = x + (size_t)F32_CHUNK * NS * H;
    float* o4 = h + (size_t)F32_CHUNK * NS * H;
    // channels-last copies of the maps, once per call (a frame is many chunks): the feature build then reads whole rows
    void* cl_base = (void*)(((uintptr_t)(o4 + (size_t)F32_CHUNK * 4) + 255) & ~(uintptr_t)255);
    const LatCL cl = (L > 0 && n_points >= 4096) ? latent_cl_build(*vw, cl_base, s) : LatCL{};
    PNR_LAUNCH_CHECK();
    for (int64_t g0 = 0; g0 < n_points; g0 += F32_CHUNK) {
        int CH = (int)((n_points - g0 < F32_CHUNK) ? (n_points - g0) : F32_CHUNK);
        PNR_TRY(features_launch(*vw, src, g0, CH, pts_per_obj, L, Din, prm->use_code_viewdirs, prm->num_freqs, prm->freq_factor, zx, E, s, cl));
        PNR_LAUNCH_CHECK();
        int32_t rc;
        if ((rc = chain_f32(mlp, zx, E, CH, NS, x, h, o4, s))) return rc;
        hipLaunchKernelGGL(k_out_act, dim3((unsigned)((CH + 255) / 256)), dim3(256), 0, s, o4, (int64_t)CH, out + g0 * 4);
        PNR_LAUNCH_CHECK();
    }
    return PNR_OK;
}

// ResnetFC.forward on caller-assembled rows (pnr_resnetfc_forward): zx (outer, NS, B, E) -> out (outer, B, d_out).
uint64_t resnetfc_f32_workspace_bytes(const pnr_mlp* mlp, int NS) {
    uint64_t per_pt = (uint64_t)NS * (mlp->d_latent + mlp->d_in + 2ull * mlp->d_hidden);
    return per_pt * F32_CHUNK * sizeof(float) + 256;
}
int32_t resnetfc_f32(const pnr_mlp* mlp, const float* zx, int64_t outer, int NS, int64_t B, float* out, void* workspace,
                     uint64_t ws_bytes, hipStream_t s) {
    if (ws_bytes < resnetfc_f32_workspace_bytes(mlp, NS)) return PNR_E_WORKSPACE;
    const int H = mlp->d_hidden, E = mlp->d_latent + mlp->d_in;
    float* zc = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float* x = zc + (size_t)F32_CHUNK * NS * E;
    float* h = x + (size_t)F32_CHUNK * NS * H;
    for (int64_t o = 0; o < outer; ++o)
        for (int64_t p0 = 0; p0 < B; p0 += F32_CHUNK) {
            const int CH = (int)((B - p0 < F32_CHUNK) ? (B - p0) : F32_CHUNK);
            const float* src = zx + ((size_t)o * NS * B + p0) * E;
            const float* rows = src;
            if (NS > 1 && CH != B) {        // the views' row blocks of this chunk are not contiguous: gather them
                PNR_HIP_CHECK(hipMemcpy2DAsync(zc, (size_t)CH * E * 4, src, (size_t)B * E * 4, (size_t)CH * E * 4, NS,
                                               hipMemcpyDeviceToDevice, s));
                rows = zc;
            }
            int32_t rc;
            if ((rc = chain_f32(mlp, rows, E, CH, NS, x, h, out + ((size_t)o * B + p0) * mlp->d_out, s))) return rc;
        }
    return PNR_OK;
}

// SpatialEncoder.index (encoder.py:138-205): uv (V, N, 2) image points -> out (V, L, N), every level sampled with its own
// latent-size normalisation and concatenated along the channels.  One thread per (view, channel, point).
static __global__ void k_index_latent(pnr_views vw, const float* __restrict__ uv, int64_t N, int uv_views, int L,
                                      float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t V = (int64_t)vw.n_objs * vw.n_views;
    if (idx >= V * L * N) return;
    const int64_t n = idx % N;
    const int e = (int)((idx / N) % L);
    const int64_t view = idx / (N * L);
    const float* q = uv + ((uv_views == 1 ? 0 : view) * N + n) * 2;          // uv of one view broadcasts (encoder.py:148-149)
    int lvl = 0, ch = e;
    while (ch >= vw.lat_c[lvl]) { ch -= vw.lat_c[lvl]; ++lvl; }
    const int W = vw.lat_w[lvl], H = vw.lat_h[lvl], C = vw.lat_c[lvl];
    const Taps t = bilinear_taps(q[0] * uv_sx(vw, lvl), q[1] * uv_sy(vw, lvl), W, H);
    const float* base = vw.latent[lvl] + ((size_t)view * C + ch) * (size_t)(H * W);
    float val = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) val += base[t.off[i]] * t.w[i];
    out[idx] = val;
}
int32_t index_latent_f32(const pnr_views* vw, const float* uv, int64_t N, int uv_views, float* out, hipStream_t s) {
    int L = 0;
    for (int i = 0; i < vw->n_levels; ++i) L += vw->lat_c[i];
    const int64_t tot = (int64_t)vw->n_objs * vw->n_views * L * N;
    if (tot == 0) return PNR_OK;
    hipLaunchKernelGGL(k_index_latent, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, *vw, uv, N, uv_views, L, out);
    PNR_LAUNCH_CHECK();
    return PNR_OK;
}

}  // namespace pnr
