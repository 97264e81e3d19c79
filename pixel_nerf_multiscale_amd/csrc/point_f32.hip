// fp32 reference-precision point path: PixelNeRFNet.forward (models.py.backup2:155-282) as plain fp32
// kernels (feature build -> tiled fp32 linear layers -> view combine -> output activation).  Any
// d_hidden / d_latent / d_in.  This is the 1e-4-tolerance parity path (PNR_F32); the production
// path is the fused MFMA kernel in point_mfma.hip.
#include "f32_kernels.h"

namespace pnr {

template <bool RELU_IN, bool ACCUM>
static int32_t linear(const float* X, int ldx, const float* W, const float* b, float* Y, int ldy, int M, int N, int K,
                      hipStream_t s) {
    dim3 grid((M + 63) / 64, (N + 63) / 64);
    hipLaunchKernelGGL((k_linear_f32<RELU_IN, ACCUM>), grid, dim3(256), 0, s, X, ldx, W, b, Y, ldy, M, N, K);
    PNR_LAUNCH_CHECK();
    return PNR_OK;
}

static const int F32_CHUNK = 16384;   // points per chunk

uint64_t point_f32_workspace_bytes(const pnr_mlp* mlp, const pnr_views* vw) {
    uint64_t per_pt = (uint64_t)vw->n_views * (mlp->d_latent + mlp->d_in + 2ull * mlp->d_hidden) + 4;
    return per_pt * F32_CHUNK * sizeof(float) + 256;
}

int32_t point_f32(const pnr_params* prm, const pnr_mlp* mlp, const pnr_views* vw, PointSrc src, int64_t n_points,
                  int64_t pts_per_obj, float* out, void* workspace, uint64_t ws_bytes, hipStream_t s) {
    if (ws_bytes < point_f32_workspace_bytes(mlp, vw)) return PNR_E_WORKSPACE;
    const int NS = vw->n_views, L = mlp->d_latent, Din = mlp->d_in, H = mlp->d_hidden, E = L + Din;
    float* zx = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float* x = zx + (size_t)F32_CHUNK * NS * E;
    float* h = x + (size_t)F32_CHUNK * NS * H;
    float* o4 = h + (size_t)F32_CHUNK * NS * H;
    const int n_lin_z = mlp->combine_layer < mlp->n_blocks ? mlp->combine_layer : mlp->n_blocks;
    for (int64_t g0 = 0; g0 < n_points; g0 += F32_CHUNK) {
        int CH = (int)((n_points - g0 < F32_CHUNK) ? (n_points - g0) : F32_CHUNK);
        int64_t tot = (int64_t)CH * NS * E;
        hipLaunchKernelGGL(k_features_f32, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, *vw, src, g0, CH,
                           pts_per_obj, L, Din, prm->use_code_viewdirs, prm->num_freqs, prm->freq_factor, zx, E);
        PNR_LAUNCH_CHECK();
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
        if (M != CH) {   // combine_layer >= n_blocks with NS > 1: the reference never reduces; out would be per view
            return PNR_E_UNSUPPORTED;
        }
        if ((rc = linear<true, false>(x, H, mlp->lin_out_w, mlp->lin_out_b, o4, 4, M, 4, H, s))) return rc;
        hipLaunchKernelGGL(k_out_act, dim3((unsigned)((CH + 255) / 256)), dim3(256), 0, s, o4, (int64_t)CH, out + g0 * 4);
        PNR_LAUNCH_CHECK();
    }
    return PNR_OK;
}

}  // namespace pnr
