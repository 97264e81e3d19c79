// C-ABI entry points: argument validation, workspace carving, stage sequencing on the caller's stream.
#include "pnr_common.h"

namespace pnr {
// point_f32.hip
uint64_t point_f32_workspace_bytes(const pnr_mlp* mlp, const pnr_views* vw);
int32_t point_f32(const pnr_params* prm, const pnr_mlp* mlp, const pnr_views* vw, PointSrc src, int64_t n_points,
                  int64_t pts_per_obj, float* out, void* workspace, uint64_t ws_bytes, hipStream_t s);
uint64_t resnetfc_f32_workspace_bytes(const pnr_mlp* mlp, int NS);
int32_t resnetfc_f32(const pnr_mlp* mlp, const float* zx, int64_t outer, int NS, int64_t B, float* out, void* workspace,
                     uint64_t ws_bytes, hipStream_t s);
int32_t index_latent_f32(const pnr_views* vw, const float* uv, int64_t N, int uv_views, float* out, hipStream_t s);
// stage_kernels.hip
int32_t sample_fine_launch(const float* rays, float near_all, float far_all, const float* z_coarse, const float* weights,
                           const float* depth, int64_t n_rays, int32_t n_coarse, int32_t n_fine, int32_t n_fine_depth,
                           float depth_std, int32_t lindisp, const float* u, const float* r, const float* g, uint64_t seed,
                           RayKey key, float* z_out, void* stream, int w_stride = 0, int d_stride = 0);
int32_t sample_coarse_launch(const float* rays, int64_t n_rays, int32_t n_coarse, int32_t lindisp, const float* noise_c,
                             uint64_t seed, RayKey key, float* z_out, void* stream);
int32_t composite_launch(const float* rays, const float* z, const float* rgbsigma, int64_t n_rays, int32_t K,
                         int32_t white_bkgd, float* weights_out, float* rgb_out, float* depth_out, int w_stride,
                         int rgb_stride, int depth_stride, void* stream);
// point_mfma.hip
uint64_t point_mfma_workspace_bytes(const pnr_mlp* mlp, const pnr_views* vw);
int32_t point_mfma(const pnr_params* prm, const pnr_mlp* mlp, const pnr_views* vw, PointSrc src, int64_t n_points,
                   int64_t pts_per_obj, float* out, void* workspace, uint64_t ws_bytes, hipStream_t s,
                   const RayJob* job = nullptr);
// train_f32.hip
uint64_t train_tape_bytes(const pnr_mlp* mlp, const pnr_views* vw, int64_t P);
uint64_t train_tape_bytes_p(const pnr_params* prm, const pnr_mlp* mlp, const pnr_views* vw, int64_t P);
uint64_t train_bwd_workspace_bytes(const pnr_mlp* mlp, const pnr_views* vw, int64_t P);
int32_t point_train_fwd(const pnr_params* prm, const pnr_mlp* mlp, const pnr_views* vw, PointSrc src, int64_t P,
                        int64_t pts_per_obj, float* out, void* tape, uint64_t tape_bytes, hipStream_t s);
int32_t point_bwd(const pnr_params* prm, const pnr_mlp* mlp, const pnr_views* vw, PointSrc src, int64_t P,
                  int64_t pts_per_obj, const float* out, const float* d_out, void* tape, uint64_t tape_bytes,
                  const pnr_mlp_grads* gr, float* const* d_latent, float* d_xyz, float* d_z, void* workspace,
                  uint64_t ws_bytes, hipStream_t s);
}  // namespace pnr

using namespace pnr;

static inline uint64_t align256(uint64_t v) { return (v + 255) & ~(uint64_t)255; }

extern "C" int32_t pnr_version(void) { return PNR_VERSION; }

extern "C" const char* pnr_error_string(int32_t code) {
    switch (code) {
        case PNR_OK: return "ok";
        case PNR_E_NULL: return "required pointer is NULL";
        case PNR_E_SHAPE: return "inconsistent or out-of-range sizes";
        case PNR_E_UNSUPPORTED: return "unsupported configuration";
        case PNR_E_WORKSPACE: return "workspace too small";
        case PNR_E_ALIGN: return "pointer must be 16-byte aligned";
        case PNR_E_PACKED: return "packed weights/latents missing or of the wrong dtype";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

static int32_t check_model(const pnr_params* prm, const pnr_mlp* mlp, const pnr_views* vw) {
    if (!prm || !mlp || !vw) return PNR_E_NULL;
    if (!mlp->lin_in_w || !mlp->lin_in_b || !mlp->lin_out_w || !mlp->lin_out_b) return PNR_E_NULL;
    if (mlp->d_out != 4 || mlp->d_hidden <= 0 || mlp->n_blocks < 1 || mlp->n_blocks > PNR_MAX_BLOCKS) return PNR_E_SHAPE;
    if (mlp->combine_type != PNR_COMBINE_AVERAGE && mlp->combine_type != PNR_COMBINE_MAX) return PNR_E_UNSUPPORTED;
    if (prm->num_freqs < 1 || prm->num_freqs > 16) return PNR_E_SHAPE;
    int d_in = prm->use_code_viewdirs ? 6 + 12 * prm->num_freqs : 3 + 6 * prm->num_freqs + 3;
    if (mlp->d_in != d_in) return PNR_E_SHAPE;
    if (vw->n_objs < 1 || vw->n_views < 1 || !vw->w2c || !vw->focal || !vw->c) return PNR_E_NULL;
    int nv = vw->n_objs * vw->n_views;
    if ((vw->n_focal != 1 && vw->n_focal != nv) || (vw->n_c != 1 && vw->n_c != nv)) return PNR_E_SHAPE;
    if (vw->n_levels < 1 || vw->n_levels > PNR_MAX_LEVELS) return PNR_E_SHAPE;
    int L = 0;
    for (int i = 0; i < vw->n_levels; ++i) {
        if (prm->precision == PNR_F32 ? !vw->latent[i] : (!vw->latent[i] && !vw->latent_packed[i])) return PNR_E_NULL;
        if (vw->lat_c[i] < 1 || vw->lat_h[i] < 2 || vw->lat_w[i] < 2) return PNR_E_SHAPE;  // (W-1) divides uv
        L += vw->lat_c[i];
    }
    if (L != mlp->d_latent) return PNR_E_SHAPE;
    int n_lin_z = mlp->combine_layer < mlp->n_blocks ? mlp->combine_layer : mlp->n_blocks;
    if (n_lin_z < 0) return PNR_E_SHAPE;
    for (int b = 0; b < mlp->n_blocks; ++b) {
        if (!mlp->fc0_w[b] || !mlp->fc0_b[b] || !mlp->fc1_w[b] || !mlp->fc1_b[b]) return PNR_E_NULL;
        if (b < n_lin_z && (!mlp->lin_z_w[b] || !mlp->lin_z_b[b])) return PNR_E_NULL;
    }
    if (vw->n_views > 1 && mlp->combine_layer >= mlp->n_blocks) return PNR_E_UNSUPPORTED;
    if (prm->precision != PNR_F32 && prm->precision != PNR_BF16 && prm->precision != PNR_F16 && prm->precision != PNR_BF16X3)
        return PNR_E_UNSUPPORTED;
    return PNR_OK;
}

static uint64_t point_workspace_bytes(const pnr_params* prm, const pnr_mlp* mlp, const pnr_views* vw) {
    return prm->precision == PNR_F32 ? point_f32_workspace_bytes(mlp, vw) : point_mfma_workspace_bytes(mlp, vw);
}

static int32_t point_dispatch(const pnr_params* prm, const pnr_mlp* mlp, const pnr_views* vw, PointSrc src,
                              int64_t n_points, int64_t pts_per_obj, float* out, void* ws, uint64_t ws_bytes,
                              hipStream_t s) {
    if (n_points == 0) return PNR_OK;
    if (prm->precision == PNR_F32) return point_f32(prm, mlp, vw, src, n_points, pts_per_obj, out, ws, ws_bytes, s);
    return point_mfma(prm, mlp, vw, src, n_points, pts_per_obj, out, ws, ws_bytes, s);
}

extern "C" int32_t pnr_point_mlp(const pnr_params* params, const pnr_mlp* mlp, const pnr_views* views,
                                 const float* rays, const float* z, int32_t K, const float* xyz,
                                 const float* viewdirs, int64_t n_points, int64_t points_per_obj, float* out,
                                 void* workspace, uint64_t workspace_bytes, void* stream) {
    int32_t rc = check_model(params, mlp, views);
    if (rc) return rc;
    if (!out) return PNR_E_NULL;
    if (((uintptr_t)out & 15) != 0) return PNR_E_ALIGN;
    PointSrc src{};
    if (rays) {
        if (!z || K <= 0) return PNR_E_NULL;
        src.rays = rays; src.z = z; src.K = K;
        if (n_points % K != 0) return PNR_E_SHAPE;
    } else {
        if (!xyz || !viewdirs) return PNR_E_NULL;
        src.xyz = xyz; src.dirs = viewdirs; src.K = 1;
    }
    if (n_points < 0 || points_per_obj <= 0 || n_points != points_per_obj * views->n_objs) return PNR_E_SHAPE;
    if (!workspace && n_points > 0) return PNR_E_NULL;
    return point_dispatch(params, mlp, views, src, n_points, points_per_obj, out, workspace, workspace_bytes,
                          (hipStream_t)stream);
}

// ------------------------------------------------------------------ module-level stage calls (fp32)
static int32_t check_mlp_f32(const pnr_mlp* mlp) {
    if (!mlp) return PNR_E_NULL;
    if (mlp->d_in <= 0 || mlp->d_latent < 0 || mlp->d_hidden <= 0 || mlp->d_out <= 0 || mlp->n_blocks < 0 ||
        mlp->n_blocks > PNR_MAX_BLOCKS) return PNR_E_SHAPE;
    if (!mlp->lin_in_w || !mlp->lin_in_b || !mlp->lin_out_w || !mlp->lin_out_b) return PNR_E_NULL;
    const int n_lin_z = mlp->combine_layer < mlp->n_blocks ? mlp->combine_layer : mlp->n_blocks;
    for (int b = 0; b < mlp->n_blocks; ++b) {
        if (!mlp->fc0_w[b] || !mlp->fc0_b[b] || !mlp->fc1_w[b] || !mlp->fc1_b[b]) return PNR_E_NULL;
        if (mlp->d_latent > 0 && b < n_lin_z && (!mlp->lin_z_w[b] || !mlp->lin_z_b[b])) return PNR_E_NULL;
    }
    return PNR_OK;
}

extern "C" uint64_t pnr_resnetfc_workspace_bytes(const pnr_mlp* mlp, int32_t n_inner_views) {
    if (!mlp || n_inner_views < 1) return 0;
    return resnetfc_f32_workspace_bytes(mlp, n_inner_views);
}

extern "C" int32_t pnr_resnetfc_forward(const pnr_mlp* mlp, const float* zx, int64_t n_outer, int32_t n_inner_views,
                                        int64_t n_inner_points, float* out, void* workspace, uint64_t workspace_bytes,
                                        void* stream) {
    int32_t rc = check_mlp_f32(mlp);
    if (rc) return rc;
    if (!zx || !out) return PNR_E_NULL;
    if (n_outer < 0 || n_inner_views < 1 || n_inner_points < 0) return PNR_E_SHAPE;
    if (n_outer == 0 || n_inner_points == 0) return PNR_OK;
    if (!workspace) return PNR_E_NULL;
    return resnetfc_f32(mlp, zx, n_outer, n_inner_views, n_inner_points, out, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int32_t pnr_index_latent(const pnr_views* views, const float* uv, int64_t n_points, int32_t uv_views,
                                    float* out, void* stream) {
    if (!views || !uv || !out) return PNR_E_NULL;
    if (views->n_levels < 1 || views->n_levels > PNR_MAX_LEVELS || views->n_objs < 1 || views->n_views < 1) return PNR_E_SHAPE;
    if (n_points < 0 || (uv_views != 1 && uv_views != views->n_objs * views->n_views)) return PNR_E_SHAPE;
    for (int i = 0; i < views->n_levels; ++i) {
        if (!views->latent[i]) return PNR_E_NULL;
        if (views->lat_c[i] < 1 || views->lat_h[i] < 1 || views->lat_w[i] < 1) return PNR_E_SHAPE;
    }
    return index_latent_f32(views, uv, n_points, uv_views, out, (hipStream_t)stream);
}

// ------------------------------------------------------------------ training entry points (train_f32.hip)
static int32_t point_args(const pnr_params* params, const pnr_mlp* mlp, const pnr_views* views, const float* rays,
                          const float* z, int32_t K, const float* xyz, const float* viewdirs, int64_t n_points,
                          int64_t points_per_obj, PointSrc* src) {
    int32_t rc = check_model(params, mlp, views);
    if (rc) return rc;
    for (int i = 0; i < views->n_levels; ++i)
        if (!views->latent[i]) return PNR_E_NULL;      // the training path reads the fp32 maps
    *src = PointSrc{};
    if (rays) {
        if (!z || K <= 0) return PNR_E_NULL;
        src->rays = rays; src->z = z; src->K = K;
        if (n_points % K != 0) return PNR_E_SHAPE;
    } else {
        if (!xyz || !viewdirs) return PNR_E_NULL;
        src->xyz = xyz; src->dirs = viewdirs; src->K = 1;
    }
    if (n_points < 0 || points_per_obj <= 0 || n_points != points_per_obj * views->n_objs) return PNR_E_SHAPE;
    return PNR_OK;
}

extern "C" uint64_t pnr_train_tape_bytes(const pnr_mlp* mlp, const pnr_views* views, int64_t n_points) {
    if (!mlp || !views || n_points < 0) return 0;
    return train_tape_bytes(mlp, views, n_points);
}

extern "C" uint64_t pnr_train_tape_bytes_for(const pnr_params* params, const pnr_mlp* mlp, const pnr_views* views,
                                             int64_t n_points) {
    if (!params || !mlp || !views || n_points < 0) return 0;
    return train_tape_bytes_p(params, mlp, views, n_points);
}

extern "C" uint64_t pnr_train_bwd_workspace_bytes(const pnr_mlp* mlp, const pnr_views* views, int64_t n_points) {
    if (!mlp || !views || n_points < 0) return 0;
    return train_bwd_workspace_bytes(mlp, views, n_points);
}

extern "C" int32_t pnr_point_mlp_train_fwd(const pnr_params* params, const pnr_mlp* mlp, const pnr_views* views,
                                           const float* rays, const float* z, int32_t K, const float* xyz,
                                           const float* viewdirs, int64_t n_points, int64_t points_per_obj,
                                           float* out, void* tape, uint64_t tape_bytes, void* stream) {
    PointSrc src;
    int32_t rc = point_args(params, mlp, views, rays, z, K, xyz, viewdirs, n_points, points_per_obj, &src);
    if (rc) return rc;
    if (!out || (!tape && n_points > 0)) return PNR_E_NULL;
    if (((uintptr_t)out & 15) != 0) return PNR_E_ALIGN;
    if (n_points == 0) return PNR_OK;
    return point_train_fwd(params, mlp, views, src, n_points, points_per_obj, out, tape, tape_bytes, (hipStream_t)stream);
}

extern "C" int32_t pnr_point_mlp_bwd(const pnr_params* params, const pnr_mlp* mlp, const pnr_views* views,
                                     const float* rays, const float* z, int32_t K, const float* xyz,
                                     const float* viewdirs, int64_t n_points, int64_t points_per_obj,
                                     const float* out, const float* d_out, void* tape, uint64_t tape_bytes,
                                     const pnr_mlp_grads* grads, float* const* d_latent, float* d_xyz, float* d_z,
                                     void* workspace, uint64_t workspace_bytes, void* stream) {
    PointSrc src;
    int32_t rc = point_args(params, mlp, views, rays, z, K, xyz, viewdirs, n_points, points_per_obj, &src);
    if (rc) return rc;
    if (!out || !d_out || !grads || ((!tape || !workspace) && n_points > 0)) return PNR_E_NULL;
    if ((((uintptr_t)out | (uintptr_t)d_out) & 15) != 0) return PNR_E_ALIGN;
    if (rays ? d_xyz != nullptr : d_z != nullptr) return PNR_E_SHAPE;   // d_z goes with rays, d_xyz with explicit points
    if (n_points == 0) return PNR_OK;
    return point_bwd(params, mlp, views, src, n_points, points_per_obj, out, d_out, tape, tape_bytes, grads, d_latent,
                     d_xyz, d_z, workspace, workspace_bytes, (hipStream_t)stream);
}

struct RenderWs { uint64_t zc, zf, rgbs, w, rgb, depth, rays, point, total; };

// `fine` (may be NULL): the point workspace is shared by both passes, so it is sized for the larger of the two MLPs
static RenderWs carve(const pnr_params* prm, const pnr_mlp* mlp, const pnr_views* vw, int64_t n, const pnr_mlp* fine = nullptr) {
    RenderWs r;
    uint64_t Kc = prm->n_coarse, Kt = (uint64_t)prm->n_coarse + prm->n_fine, off = 0;
    r.zc = off; off += align256(n * Kc * 4);
    r.zf = off; off += align256(n * Kt * 4);
    r.rgbs = off; off += align256(n * Kt * 16);
    r.w = off; off += align256(n * Kt * 4);
    r.rgb = off; off += align256(n * 12);
    r.depth = off; off += align256(n * 4);
    r.rays = off; off += align256(n * 32);          // pnr_render_camera on the fp32 path materialises its rays here
    uint64_t pw = point_workspace_bytes(prm, mlp, vw);
    if (fine) { uint64_t pf = point_workspace_bytes(prm, fine, vw); if (pf > pw) pw = pf; }
    r.point = off; off += align256(pw);
    r.total = off + 256;
    return r;
}

extern "C" uint64_t pnr_workspace_bytes(const pnr_params* params, const pnr_mlp* mlp, const pnr_views* views,
                                        int64_t n_rays) {
    if (!params || !mlp || !views || n_rays < 0) return 0;
    return carve(params, mlp, views, n_rays).total;
}

// pnr_render / pnr_render_camera: `rays` or `cam` (+ first pixel)
static int32_t render_impl(const pnr_params* params, const pnr_mlp* coarse, const pnr_mlp* fine, const pnr_views* views,
                           const float* rays, const RayCam* cam, int64_t pix0, int64_t n_rays, int64_t rays_per_obj,
                           const pnr_noise* noise, uint64_t seed, int64_t ray_index_base, const pnr_outputs* outputs,
                           void* workspace, uint64_t workspace_bytes, void* stream) {
    int32_t rc = check_model(params, coarse, views);
    if (rc) return rc;
    if (fine && (rc = check_model(params, fine, views))) return rc;
    if ((!rays && !cam) || !outputs) return PNR_E_NULL;
    if (params->n_coarse < 1 || params->n_fine < 0 || params->n_fine_depth < 0 ||
        params->n_fine_depth > params->n_fine) return PNR_E_SHAPE;
    if (n_rays < 0 || rays_per_obj <= 0 || n_rays != rays_per_obj * views->n_objs) return PNR_E_SHAPE;
    if (n_rays == 0) return PNR_OK;
    if (!workspace) return PNR_E_NULL;
    RenderWs cw = carve(params, coarse, views, n_rays, params->n_fine > 0 ? fine : nullptr);
    if (workspace_bytes < cw.total) return PNR_E_WORKSPACE;     // pnr_workspace_bytes of BOTH MLPs, the larger one
    char* base = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    hipStream_t s = (hipStream_t)stream;
    const int Kc = params->n_coarse, Kf = params->n_fine, Kfd = params->n_fine_depth, Kt = Kc + Kf;
    float* zc = outputs->z_coarse ? outputs->z_coarse : (float*)(base + cw.zc);
    float* zf = outputs->z_fine ? outputs->z_fine : (float*)(base + cw.zf);
    float* rgbs = (float*)(base + cw.rgbs);
    float* w_c = outputs->coarse_weights ? outputs->coarse_weights : (float*)(base + cw.w);
    float* rgb_c = outputs->coarse_rgb ? outputs->coarse_rgb : (float*)(base + cw.rgb);
    float* dep_c = outputs->coarse_depth ? outputs->coarse_depth : (float*)(base + cw.depth);
    void* pws = base + cw.point;
    uint64_t pws_bytes = workspace_bytes - (uint64_t)((char*)pws - (char*)workspace);
    const pnr_noise nz = noise ? *noise : pnr_noise{nullptr, nullptr, nullptr, nullptr, 0};
    if (Kf > 0 && (!outputs->fine_rgb || !outputs->fine_depth)) return PNR_E_NULL;
    // the generator's key (pnr_noise.ray_index_obj_stride) and the output row strides (pnr_outputs); the coarse-pass
    // outputs that only live in the workspace are dense
    if (nz.ray_index_obj_stride < 0 || (nz.ray_index_obj_stride != 0 && nz.ray_index_obj_stride < rays_per_obj)) return PNR_E_SHAPE;
    const RayKey key{ray_index_base, nz.ray_index_obj_stride ? nz.ray_index_obj_stride - rays_per_obj : 0, rays_per_obj};
    if (outputs->rgb_stride < 0 || outputs->depth_stride < 0 || outputs->coarse_weights_stride < 0 ||
        outputs->fine_weights_stride < 0) return PNR_E_SHAPE;
    const int s_rgb = outputs->rgb_stride ? outputs->rgb_stride : 3, s_dep = outputs->depth_stride ? outputs->depth_stride : 1;
    const int s_rgb_c = outputs->coarse_rgb ? s_rgb : 3, s_dep_c = outputs->coarse_depth ? s_dep : 1;
    const int s_w_c = (outputs->coarse_weights && outputs->coarse_weights_stride) ? outputs->coarse_weights_stride : Kc;
    const int s_w_f = outputs->fine_weights_stride ? outputs->fine_weights_stride : Kt;

    if (params->precision != PNR_F32) {
        // The MFMA kernel renders a pass in ONE launch: rays from the camera (if any) and coarse positions generated in the tile
        // prologue, the network, and the compositing of every finished ray by the workgroup that evaluated it
        // (point_mfma.hip).  Between the passes one small launch resamples (a 256-element sort per ray has no place
        // between two MFMA tiles).
        RayJob job{};
        job.on = 1; job.K = Kc; job.gen_z = 1; job.lindisp = params->lindisp; job.white_bkgd = params->white_bkgd;
        job.n_rays = n_rays; job.noise_c = nz.noise_c; job.seed = seed; job.key = key;
        job.z_out = zc; job.rgb_out = rgb_c; job.depth_out = dep_c;
        job.z_needed = (Kf > 0 || outputs->z_coarse) ? 1 : 0;        // the resampling / the caller read the coarse positions
        job.rgb_stride = s_rgb_c; job.depth_stride = s_dep_c; job.w_stride = s_w_c;
        job.w_out = (Kf > 0 || outputs->coarse_weights) ? w_c : nullptr;      // only the resampling and the caller read them
        if (cam) { job.from_cam = 1; job.cam = *cam; job.pix0 = (int)pix0; }
        PointSrc src{rays, nullptr, Kc, nullptr, nullptr};
        if (outputs->ev_point_begin) PNR_HIP_CHECK(hipEventRecord((hipEvent_t)outputs->ev_point_begin, s));
        if ((rc = point_mfma(params, coarse, views, src, n_rays * Kc, rays_per_obj * Kc, rgbs, pws, pws_bytes, s, &job))) return rc;
        if (outputs->ev_point_end) PNR_HIP_CHECK(hipEventRecord((hipEvent_t)outputs->ev_point_end, s));
        if (Kf == 0) return PNR_OK;
        // Small batches: the fine launch resamples its own rays first (sample_fine .. sort, one wave per ray in the wave's LDS
        // buffer) — no third launch.  Whole frames: a workgroup owns ~64 rays and would sort them 4 at a time in front of its
        // first tile (measured +1.3 % frame time at 16384 rays x 64+32), against 0.5 % for one chip-wide resampling launch.
        int P2 = 1;
        while (P2 < Kt) P2 <<= 1;
        const bool in_kernel = (uint64_t)(P2 + Kc + 2) * 4 <= 16384 && n_rays <= 2048;
        if (!in_kernel) {
            if ((rc = sample_fine_launch(rays, cam ? cam->zn : 0.f, cam ? cam->zf : 0.f, zc, w_c, dep_c, n_rays, Kc, Kf, Kfd,
                                         params->depth_std, params->lindisp, nz.u, nz.r, nz.g, seed, key, zf, stream, s_w_c, s_dep_c)))
                return rc;
        }
        job.resample = in_kernel ? 1 : 0;
        job.fine = FineArgs{Kc, Kf - Kfd, Kfd, P2, params->lindisp, params->depth_std, nz.u, nz.r, nz.g, seed};
        job.zc = zc; job.wc = w_c; job.depth_c = dep_c; job.z_fine = zf;
        job.wc_stride = s_w_c; job.dc_stride = s_dep_c;
        job.K = Kt; job.gen_z = 0; job.noise_c = nullptr; job.z_out = nullptr;
        job.w_out = outputs->fine_weights; job.rgb_out = outputs->fine_rgb; job.depth_out = outputs->fine_depth;
        job.rgb_stride = s_rgb; job.depth_stride = s_dep; job.w_stride = s_w_f;
        PointSrc srcf{rays, zf, Kt, nullptr, nullptr};
        return point_mfma(params, fine ? fine : coarse, views, srcf, n_rays * Kt, rays_per_obj * Kt, rgbs, pws, pws_bytes, s, &job);
    }

    // fp32 path: the stages as separate launches
    if (!rays) {
        float* rbuf = (float*)(base + cw.rays);
        // c2w back from the RayCam for the stage entry point
        const float m[16] = {cam->R[0], cam->R[1], cam->R[2], cam->o[0], cam->R[3], cam->R[4], cam->R[5], cam->o[1],
                             cam->R[6], cam->R[7], cam->R[8], cam->o[2], 0.f, 0.f, 0.f, 1.f};
        if ((rc = pnr_gen_rays(m, cam->W, cam->H, cam->fx, cam->fy, cam->cx, cam->cy, cam->zn, cam->zf, pix0, n_rays, rbuf, stream)))
            return rc;
        rays = rbuf;
    }
    // coarse pass (nerf.py:273-282)
    if ((rc = sample_coarse_launch(rays, n_rays, Kc, params->lindisp, nz.noise_c, seed, key, zc, stream))) return rc;
    PointSrc src{rays, zc, Kc, nullptr, nullptr};
    if (outputs->ev_point_begin) PNR_HIP_CHECK(hipEventRecord((hipEvent_t)outputs->ev_point_begin, s));
    if ((rc = point_dispatch(params, coarse, views, src, n_rays * Kc, rays_per_obj * Kc, rgbs, pws, pws_bytes, s))) return rc;
    if (outputs->ev_point_end) PNR_HIP_CHECK(hipEventRecord((hipEvent_t)outputs->ev_point_end, s));
    if ((rc = composite_launch(rays, zc, rgbs, n_rays, Kc, params->white_bkgd, w_c, rgb_c, dep_c, s_w_c, s_rgb_c, s_dep_c, stream))) return rc;
    if (Kf == 0) return PNR_OK;

    // fine pass (nerf.py:284-301); mlp_fine=None falls back to the coarse MLP (backup2:258)
    if ((rc = sample_fine_launch(rays, 0.f, 0.f, zc, w_c, dep_c, n_rays, Kc, Kf, Kfd, params->depth_std, params->lindisp,
                                 nz.u, nz.r, nz.g, seed, key, zf, stream, s_w_c, s_dep_c))) return rc;
    PointSrc srcf{rays, zf, Kt, nullptr, nullptr};
    if ((rc = point_dispatch(params, fine ? fine : coarse, views, srcf, n_rays * Kt, rays_per_obj * Kt, rgbs, pws,
                             pws_bytes, s))) return rc;
    return composite_launch(rays, zf, rgbs, n_rays, Kt, params->white_bkgd, outputs->fine_weights, outputs->fine_rgb,
                            outputs->fine_depth, s_w_f, s_rgb, s_dep, stream);
}

extern "C" int32_t pnr_render(const pnr_params* params, const pnr_mlp* coarse, const pnr_mlp* fine,
                              const pnr_views* views, const float* rays, int64_t n_rays, int64_t rays_per_obj,
                              const pnr_noise* noise, uint64_t seed, int64_t ray_index_base,
                              const pnr_outputs* outputs, void* workspace, uint64_t workspace_bytes, void* stream) {
    if (!rays) return PNR_E_NULL;
    return render_impl(params, coarse, fine, views, rays, nullptr, 0, n_rays, rays_per_obj, noise, seed, ray_index_base,
                       outputs, workspace, workspace_bytes, stream);
}

extern "C" int32_t pnr_render_camera(const pnr_params* params, const pnr_mlp* coarse, const pnr_mlp* fine,
                                     const pnr_views* views, const float* c2w, int32_t W, int32_t H, float fx, float fy,
                                     float cx, float cy, float z_near, float z_far, int64_t pix0, int64_t n_rays,
                                     const pnr_noise* noise, uint64_t seed, int64_t ray_index_base,
                                     const pnr_outputs* outputs, void* workspace, uint64_t workspace_bytes, void* stream) {
    if (!c2w || !views) return PNR_E_NULL;
    if (W <= 0 || H <= 0 || n_rays < 0 || pix0 < 0 || pix0 + n_rays > (int64_t)W * H || (int64_t)W * H > 0x7fffffffLL)
        return PNR_E_SHAPE;
    if (views->n_objs != 1) return PNR_E_SHAPE;          // one camera looks at one object
    const RayCam cam = make_ray_cam(c2w, W, H, fx, fy, cx, cy, z_near, z_far);
    return render_impl(params, coarse, fine, views, nullptr, &cam, pix0, n_rays, n_rays, noise, seed, ray_index_base,
                       outputs, workspace, workspace_bytes, stream);
}

// ---- hipEvent helpers for ctypes callers (bench.py times kernels on the stream they run on)
extern "C" int32_t pnr_event_create(void** ev) {
    if (!ev) return PNR_E_NULL;
    hipEvent_t e;
    PNR_HIP_CHECK(hipEventCreate(&e));
    *ev = (void*)e;
    return PNR_OK;
}
extern "C" int32_t pnr_event_record(void* ev, void* stream) {
    if (!ev) return PNR_E_NULL;
    PNR_HIP_CHECK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return PNR_OK;
}
extern "C" int32_t pnr_event_elapsed_ms(void* start, void* stop, float* ms) {
    if (!start || !stop || !ms) return PNR_E_NULL;
    PNR_HIP_CHECK(hipEventSynchronize((hipEvent_t)stop));
    PNR_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return PNR_OK;
}
extern "C" int32_t pnr_event_destroy(void* ev) {
    if (!ev) return PNR_E_NULL;
    PNR_HIP_CHECK(hipEventDestroy((hipEvent_t)ev));
    return PNR_OK;
}
