/*
 * pnr.h — C ABI of libpnr_hip.so, the MI355X (gfx950) native pixelNeRF render hot path.
 *
 * The reference (Zxhh123/pixel-nerf-multiscale) is pure Python and has NO native interface; this header
 * is the boundary a maintainer would bind with ctypes.  Each entry point names
 * the reference code it replaces (paths relative to the reference's src/).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors' data_ptr()), fp32,
 *    contiguous, row-major, unless stated otherwise; the library never allocates or frees device
 *    memory and keeps no mutable global state.
 *  - `stream` is a hipStream_t (0 = default stream).  Every call is asynchronous on it and
 *    never synchronises the host.
 *  - return: 0 ok; <0 PNR_E_* (bad argument / unsupported configuration); >0 a hipError_t.
 */
#ifndef PNR_H
#define PNR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNR_VERSION 102          /* 0.1.2: pnr_views.uv_scale_{x,y} (opt-in upstream texel mapping); 101: output strides, per-object ray index stride */
#define PNR_MAX_LEVELS 5         /* encoder levels of a multi-scale latent (encoder.py:62-73) */
#define PNR_MAX_BLOCKS 8         /* ResnetFC blocks (resnetfc.py:147) */

enum {
    PNR_OK = 0,
    PNR_E_NULL = -1,             /* required pointer is NULL */
    PNR_E_SHAPE = -2,            /* inconsistent / out-of-range sizes */
    PNR_E_UNSUPPORTED = -3,      /* configuration the kernels do not implement */
    PNR_E_WORKSPACE = -4,        /* workspace too small */
    PNR_E_ALIGN = -5,            /* pointer not 16-byte aligned where required */
    PNR_E_PACKED = -6            /* packed weights / latents missing or of the wrong kind */
};

enum { PNR_F32 = 0, PNR_BF16 = 1, PNR_F16 = 2,          /* arithmetic type of the fc layers */
       PNR_BF16X3 = 3 };                                 /* training entry points only: bf16 MFMA with every operand split
                                                          * hi + lo, 3 products per term — fp32-class results */
enum { PNR_COMBINE_AVERAGE = 0, PNR_COMBINE_MAX = 1 }; /* util.combine_interleaved (util.py:466-476) */

/* ResnetFC parameters (resnetfc.py:128-158), PyTorch nn.Linear layout: weight (out,in), y = x W^T + b.
 * State-dict keys: lin_in, lin_z.{b}, blocks.{b}.fc_0 / fc_1, lin_out.  use_spade / softplus (beta>0)
 * are not supported (no shipped config uses them). */
typedef struct pnr_mlp {
    int32_t d_in;                /* 42 = 39 pos-enc + 3 viewdirs, or 78 with use_code_viewdirs */
    int32_t d_latent;            /* 256 single-scale, 512 multi-scale; sum of level channels */
    int32_t d_hidden;            /* 512 in every shipped config; MFMA path requires 512 */
    int32_t d_out;               /* 4 */
    int32_t n_blocks;            /* 5 */
    int32_t combine_layer;       /* 3: views are reduced before this block; lin_z exists for b < min(combine_layer, n_blocks) */
    int32_t combine_type;        /* PNR_COMBINE_* */
    int32_t packed_objs;         /* projected streams only: number of objects `packed` holds a stream for (0 = 1) */
    const float* lin_in_w;  const float* lin_in_b;
    const float* lin_z_w[PNR_MAX_BLOCKS];  const float* lin_z_b[PNR_MAX_BLOCKS];
    const float* fc0_w[PNR_MAX_BLOCKS];    const float* fc0_b[PNR_MAX_BLOCKS];
    const float* fc1_w[PNR_MAX_BLOCKS];    const float* fc1_b[PNR_MAX_BLOCKS];
    const float* lin_out_w; const float* lin_out_b;
    /* produced once by pnr_pack_mlp(); required when precision != PNR_F32 */
    const void* packed;
    uint64_t packed_bytes;
    int32_t packed_dtype;        /* PNR_BF16 / PNR_F16 */
    int32_t packed_texels;       /* 0: plain stream (pnr_pack_mlp).  T > 0: stream from pnr_pack_mlp_projected; T = Hl*Wl of
                                  * the last latent level (lin_z pre-multiplied with that level's maps) */
} pnr_mlp;

/* What PixelNeRFNet.encode() leaves on the module (models.py.backup2:108-150) + the encoder's latent
 * map(s) (encoder.py:106-136).  View index = obj * n_views + v (repeat_interleave order, util.py:58-65). */
typedef struct pnr_views {
    int32_t n_objs;              /* SB */
    int32_t n_views;             /* NS per object */
    const float* w2c;            /* (SB*NS, 3, 4) [R^T | -R^T t] */
    const float* focal;          /* (n_focal, 2): fx, fy with fy ALREADY negated (backup2:139) */
    const float* c;              /* (n_c, 2) principal point */
    int32_t n_focal;             /* 1 (broadcast) or SB*NS */
    int32_t n_c;                 /* 1 (broadcast) or SB*NS */
    int32_t n_levels;            /* 1 = single-scale */
    int32_t reserved0;
    const float* latent[PNR_MAX_LEVELS];   /* (SB*NS, C_i, H_i, W_i) NCHW fp32, reference layout */
    int32_t lat_c[PNR_MAX_LEVELS];
    int32_t lat_h[PNR_MAX_LEVELS];
    int32_t lat_w[PNR_MAX_LEVELS];
    /* required when precision != PNR_F32: per level (SB*NS, H_i, W_i, C_i) channels-last, 16-bit (packed_dtype),
     * 16-byte aligned.  Either carved from the blob pnr_pack_latents() writes (level i at the offset it reports), or
     * the encoder's own output when it already runs channels-last in half precision (zero copy).
     * With these present the fp32 NCHW pointers above may be NULL. */
    const void* latent_packed[PNR_MAX_LEVELS];
    int32_t packed_dtype;
    int32_t reserved1;
    /* Texel coordinate of an image point on level i = uv * uv_scale_{x,y}[i].  0 (the default of a zeroed struct) = 1.0 =
     * the reference fork's mapping: encoder.py:152-164 normalises uv by the LATENT size and ignores image_size, so the
     * texel coordinate equals the image-pixel coordinate — the parity target.  Opt-in: upstream pixelNeRF's
     * mapping (uv * latent_scaling / image_size, align_corners) = W_i / W_image, H_i / H_image per level — what a
     * checkpoint trained with upstream semantics expects (SpatialEncoder.uv_scale = "image"; parity unpinned: the
     * reference holds no fixture for it).  The gradient of the lookup w.r.t. uv carries the same factor. */
    float uv_scale_x[PNR_MAX_LEVELS];
    float uv_scale_y[PNR_MAX_LEVELS];
} pnr_views;

/* NeRFRenderer attributes (render/nerf.py:62-96) + the PixelNeRFNet switches the kernels need. */
typedef struct pnr_params {
    int32_t n_coarse;            /* Kc */
    int32_t n_fine;              /* Kf (0 = coarse pass only) */
    int32_t n_fine_depth;        /* Kfd <= Kf */
    int32_t white_bkgd;
    int32_t lindisp;
    int32_t use_code_viewdirs;   /* positional-encode [xyz, viewdirs] together (backup2:207-209) */
    int32_t num_freqs;           /* 6 (code.py:11) */
    int32_t precision;           /* PNR_F32 | PNR_BF16 | PNR_F16 */
    float depth_std;
    float freq_factor;           /* 1.5 (conf/default.conf:18) */
    int32_t train_tape_fp32;     /* training with precision = PNR_BF16: 0 = 16-bit tape (block inputs and fc_0 outputs kept as
                                  * bf16 — the values the bf16-product GEMMs stage anyway: gradients are bit-identical to the
                                  * fp32 tape's), 1 = fp32 tape */
    int32_t park_fp32;           /* fused kernel, several source views: 0 = the per-view residual streams wait for the view
                                  * reduction (util.combine_interleaved, util.py:466-476) in the kernel's 16-bit format — half
                                  * the bytes, the rounding every layer input takes anyway; 1 = as fp32, the reference's
                                  * reduction of fp32 activations (symmetric in the view order), ~3 % slower on 3-view shapes */
    int32_t reserved[4];
} pnr_params;

/* Explicit random draws, reference order (render/nerf.py:111,135,141,158).  A NULL member (or a NULL
 * struct) selects the in-kernel counter-based generator keyed by (seed, global ray index). */
typedef struct pnr_noise {
    const float* noise_c;        /* (N, Kc)      U[0,1) */
    const float* u;              /* (N, Kf-Kfd)  U[0,1) */
    const float* r;              /* (N, Kf-Kfd)  U[0,1) */
    const float* g;              /* (N, Kfd)     N(0,1) */
    /* Key of the in-kernel generator for a call that holds a RANGE of every object's rays (one rank's shard of an
     * (SB, B, 8) batch cut along B, as nn.DataParallel(dim=1) cuts it, render/nerf.py:367-371): ray i of object o counts as
     * global ray  ray_index_base + o * ray_index_obj_stride + i.  0 = objects follow each other (stride = rays_per_obj),
     * which is what an unsharded call means; a shard passes base = first ray of its range, stride = B. */
    int64_t ray_index_obj_stride;
} pnr_noise;

/* Outputs of NeRFRenderer.forward (render/nerf.py:278-303); any member may be NULL. */
typedef struct pnr_outputs {
    float* coarse_rgb;           /* (N,3) */
    float* coarse_depth;         /* (N)   */
    float* coarse_weights;       /* (N,Kc) */
    float* fine_rgb;             /* (N,3) */
    float* fine_depth;           /* (N)   */
    float* fine_weights;         /* (N,Kc+Kf) */
    float* z_coarse;             /* (N,Kc)    sampled depths (debug / tests) */
    float* z_fine;               /* (N,Kc+Kf) sorted */
    /* optional hipEvent_t handles recorded on `stream` immediately before / after the COARSE pass's point-network
     * launch (the dominant kernel), so a caller can time that kernel inside a whole-path call; NULL = off */
    void* ev_point_begin;
    void* ev_point_end;
    /* Row strides in floats of the rgb / depth / weights outputs; 0 = dense (3, 1, Kc, Kc+Kf).  With strides the members
     * above may point INTO one packed per-ray record — e.g. rgb at +0, depth at +3 of a (N, 4) buffer that is this rank's
     * slice of an all_gather buffer — so a sharded frame is written where the collective reads it (no copies). */
    int32_t rgb_stride;
    int32_t depth_stride;
    int32_t coarse_weights_stride;
    int32_t fine_weights_stride;
} pnr_outputs;

int32_t pnr_version(void);
const char* pnr_error_string(int32_t code);

/* ---- one-time packing ------------------------------------------------------------------------ */
/* Repack an MLP's fp32 weights into the fragment stream the MFMA kernel consumes in order
 * (bf16 or fp16; biases folded in).  `out` must hold pnr_packed_mlp_bytes() bytes, 16-B aligned. */
uint64_t pnr_packed_mlp_bytes(const pnr_mlp* mlp);
int32_t pnr_pack_mlp(const pnr_mlp* mlp, int32_t dtype, void* out, uint64_t out_bytes, void* stream);
/* Projected stream: for 1..16 objects (1..8 source views each) whose LAST latent level has 256 channels on
 * 4 <= Hl*Wl <= 256 texels (single-scale SRN / NMR maps; the coarsest level of the multi-scale encoder — the levels
 * before it, whole 256-channel groups, are still gathered) bilinear lookup and lin_z are both linear, so
 * lin_z_b(index(uv)) = (W_z,b . Lat) . w(uv) with w the point's 4 tap weights spread over the Hl*Wl texels.  The
 * stream then carries W_z,b . Lat (512 x Hl*Wl; the block's bias added to every column, the tap weights summing to 1) in
 * place of W_z,b (512 x d_latent) and the kernel needs no latent
 * gather; with several views the per-view part of the stream is laid out once per view, each copy with its own
 * view's product; with several objects the blob holds one such stream per object and the kernels assign their workgroups
 * per object.  Re-pack whenever the weights OR the latent maps change; pnr_packed_mlp_projected_bytes() returns 0 when
 * the shapes do not qualify.  Set pnr_mlp.packed_texels = Hl*Wl and pnr_mlp.packed_objs = views->n_objs on the struct that
 * carries this stream; it is refused (PNR_E_PACKED) for any other object count or map size. */
uint64_t pnr_packed_mlp_projected_bytes(const pnr_mlp* mlp, const pnr_views* views);
int32_t pnr_pack_mlp_projected(const pnr_mlp* mlp, const pnr_views* views, int32_t dtype, void* out,
                               uint64_t out_bytes, void* stream);
/* Channels-last low-precision copy of the latent maps for the MFMA kernel's gather. */
uint64_t pnr_packed_latent_bytes(const pnr_views* views);
/* level_offsets (host array of PNR_MAX_LEVELS, may be NULL) receives the byte offset of every level inside `out` */
int32_t pnr_pack_latents(const pnr_views* views, int32_t dtype, void* out, uint64_t out_bytes,
                         uint64_t* level_offsets, void* stream);

/* ---- stage entry points (also what the tests call) ------------------------------------------- */
/* NeRFRenderer.sample_coarse (render/nerf.py:98-118).  rays (N,8) -> z (N,Kc). */
int32_t pnr_sample_coarse(const float* rays, int64_t n_rays, int32_t n_coarse, int32_t lindisp,
                          const float* noise_c, uint64_t seed, int64_t ray_index_base,
                          float* z_out, void* stream);

/* NeRFRenderer.composite, compositing half (render/nerf.py:178-182,223-249).
 * rgbsigma (N,K,4) model output -> weights (N,K) [nullable], rgb (N,3), depth (N). */
int32_t pnr_composite(const float* rays, const float* z, const float* rgbsigma, int64_t n_rays, int32_t K,
                      int32_t white_bkgd, float* weights_out, float* rgb_out, float* depth_out, void* stream);

/* sample_fine + sample_fine_depth + cat + sort (render/nerf.py:120-161,285-295).
 * -> z_out (N, Kc + n_fine) ascending. */
int32_t pnr_sample_fine(const float* rays, const float* z_coarse, const float* weights, const float* depth,
                        int64_t n_rays, int32_t n_coarse, int32_t n_fine, int32_t n_fine_depth,
                        float depth_std, int32_t lindisp, const float* u, const float* r, const float* g,
                        uint64_t seed, int64_t ray_index_base, float* z_out, void* stream);

/* PixelNeRFNet.forward (models.py.backup2:155-282): world points -> (r,g,b,sigma).
 * Two ways to name the points:
 *   rays != NULL: point (ray i, sample k) = o_i + z[i,k] d_i, viewdir d_i  (render/nerf.py:185,204);
 *                 n_points = n_rays*K, out (n_rays, K, 4)
 *   rays == NULL: explicit xyz / viewdirs (SB, P, 3); n_points = SB*P, out (SB, P, 4)
 * points_per_obj = points per object (n_points / views->n_objs). */
int32_t pnr_point_mlp(const pnr_params* params, const pnr_mlp* mlp, const pnr_views* views,
                      const float* rays, const float* z, int32_t K,
                      const float* xyz, const float* viewdirs,
                      int64_t n_points, int64_t points_per_obj,
                      float* out, void* workspace, uint64_t workspace_bytes, void* stream);

/* ---- the whole path: NeRFRenderer.forward with a PixelNeRFNet model (render/nerf.py:251-303) --- */
/* rays (N,8), N = SB*B, rays_per_obj = B.  fine may be NULL (mlp_fine=None -> coarse MLP, backup2:258).
 * PNR_BF16 / PNR_F16: each pass is ONE kernel launch — coarse positions (sample_coarse) are generated in the kernel, every
 * workgroup composites the rays it evaluated; the resampling (sample_fine .. sort) runs at the head of the fine launch for
 * batches of up to 2048 rays and as one launch between the passes above that; the results are bit-identical to calling the
 * stage entry points above in sequence.  PNR_F32: the stages in sequence. */
uint64_t pnr_workspace_bytes(const pnr_params* params, const pnr_mlp* mlp, const pnr_views* views,
                             int64_t n_rays);
int32_t pnr_render(const pnr_params* params, const pnr_mlp* coarse, const pnr_mlp* fine,
                   const pnr_views* views, const float* rays, int64_t n_rays, int64_t rays_per_obj,
                   const pnr_noise* noise, uint64_t seed, int64_t ray_index_base,
                   const pnr_outputs* outputs, void* workspace, uint64_t workspace_bytes, void* stream);

/* ---- training: the same path under autograd, train/train.py:324-346,382-410 ---------- */
/* Gradient buffers, shaped like the pnr_mlp weights; every non-NULL member is ACCUMULATED into (+=, fp32
 * atomics), so the caller zeroes them (or passes .grad tensors).  NULL members are skipped. */
typedef struct pnr_mlp_grads {
    float* lin_in_w;  float* lin_in_b;
    float* lin_z_w[PNR_MAX_BLOCKS];  float* lin_z_b[PNR_MAX_BLOCKS];
    float* fc0_w[PNR_MAX_BLOCKS];    float* fc0_b[PNR_MAX_BLOCKS];
    float* fc1_w[PNR_MAX_BLOCKS];    float* fc1_b[PNR_MAX_BLOCKS];
    float* lin_out_w; float* lin_out_b;
} pnr_mlp_grads;

/* Saved activations of one pnr_point_mlp_train_fwd call (what autograd would keep for ResnetFC.forward,
 * resnetfc.py:173-236) and the scratch its backward needs.  fp32 arithmetic, fp32 latent maps required. */
uint64_t pnr_train_tape_bytes(const pnr_mlp* mlp, const pnr_views* views, int64_t n_points);   /* fp32 tape: the upper bound */
/* the tape of a call with these params (precision = PNR_BF16 keeps a 16-bit tape: ~0.6 x the bytes) */
uint64_t pnr_train_tape_bytes_for(const pnr_params* params, const pnr_mlp* mlp, const pnr_views* views, int64_t n_points);
uint64_t pnr_train_bwd_workspace_bytes(const pnr_mlp* mlp, const pnr_views* views, int64_t n_points);

/* PixelNeRFNet.forward as pnr_point_mlp (same point naming), keeping the tape. */
int32_t pnr_point_mlp_train_fwd(const pnr_params* params, const pnr_mlp* mlp, const pnr_views* views,
                                const float* rays, const float* z, int32_t K,
                                const float* xyz, const float* viewdirs,
                                int64_t n_points, int64_t points_per_obj,
                                float* out, void* tape, uint64_t tape_bytes, void* stream);

/* Backward of the call above: d_out (n_points,4) w.r.t. the activated outputs `out`.
 *   grads      MLP weight/bias gradients (+=)
 *   d_latent   per level, same shape as views->latent[level] (+=), or NULL entries / NULL array (encoder frozen,
 *              PixelNeRFNet.stop_encoder_grad, models.py.backup2:228-229)
 *   d_xyz      (n_points,3) explicit mode, or d_z (n_rays,K) rays mode: gradient w.r.t. the sample
 *              positions (needed because nerf.py:287-289 does not detach the depth-guided samples); NULL = skip */
int32_t pnr_point_mlp_bwd(const pnr_params* params, const pnr_mlp* mlp, const pnr_views* views,
                          const float* rays, const float* z, int32_t K,
                          const float* xyz, const float* viewdirs,
                          int64_t n_points, int64_t points_per_obj,
                          const float* out, const float* d_out, void* tape, uint64_t tape_bytes,
                          const pnr_mlp_grads* grads, float* const* d_latent, float* d_xyz, float* d_z,
                          void* workspace, uint64_t workspace_bytes, void* stream);

/* Backward of pnr_composite (render/nerf.py:178-182,223-249).  d_weights / d_rgb / d_depth may be NULL (zero);
 * writes d_rgbsigma (N,K,4) and, when non-NULL, d_z (N,K) (deltas and depth depend on z). */
int32_t pnr_composite_bwd(const float* rays, const float* z, const float* rgbsigma, int64_t n_rays, int32_t K,
                          int32_t white_bkgd, const float* d_weights, const float* d_rgb, const float* d_depth,
                          float* d_rgbsigma, float* d_z, void* stream);

/* Backward of pnr_sample_fine w.r.t. the coarse depth (only the n_fine_depth samples are differentiable):
 * g / seed / ray_index_base as given to the forward call; z_sorted is its output. */
int32_t pnr_sample_fine_bwd(const float* rays, const float* depth, int64_t n_rays, int32_t n_coarse,
                            int32_t n_fine, int32_t n_fine_depth, float depth_std, const float* g,
                            uint64_t seed, int64_t ray_index_base, const float* z_sorted,
                            const float* d_z_sorted, float* d_depth, void* stream);

/* ResnetFC.forward (model/resnetfc.py:173-236) on rows the caller assembled: zx (n_outer, n_inner_views, n_inner_points,
 * d_latent + d_in) fp32 with the latent part FIRST; the n_inner_views row blocks are reduced (mean / max, util.py:466-476)
 * in front of block combine_layer, i.e. combine_inner_dims = (n_inner_views, n_inner_points).  out (n_outer,
 * n_inner_points, d_out), no activation.  fp32 arithmetic (the 1e-4 parity path); any d_hidden / d_out. */
uint64_t pnr_resnetfc_workspace_bytes(const pnr_mlp* mlp, int32_t n_inner_views);
int32_t pnr_resnetfc_forward(const pnr_mlp* mlp, const float* zx, int64_t n_outer, int32_t n_inner_views,
                             int64_t n_inner_points, float* out, void* workspace, uint64_t workspace_bytes,
                             void* stream);

/* SpatialEncoder.index (model/encoder.py:138-205): uv (uv_views, n_points, 2) image points -> out (n_objs*n_views, L,
 * n_points) fp32; bilinear, border padding, align_corners, every level normalised by ITS latent size and concatenated
 * along the channels.  uv_views = 1 broadcasts one set of points to every view (encoder.py:148-149), else n_objs*n_views.
 * Reads views->latent[] (fp32 NCHW) and lat_c/h/w only. */
int32_t pnr_index_latent(const pnr_views* views, const float* uv, int64_t n_points, int32_t uv_views, float* out,
                         void* stream);

/* pnr_render for the rays of ONE camera, generated inside the render launch (util.gen_rays, util/util.py:118-148,243-281, then
 * NeRFRenderer.forward): ray i is pixel pix0 + i (row-major) of the W x H pinhole image of camera-to-world matrix c2w.
 * What the reference's eval drivers do per frame (eval/eval.py:250-293: gen_rays on the host, H2D, split, render_par per
 * chunk) as one call with no ray tensor.  views->n_objs must be 1.  Results are bit-identical to pnr_gen_rays + pnr_render.
 * ray_index_base counts rays for the in-kernel noise exactly as in pnr_render (pass pix0 when sharding one frame). */
int32_t pnr_render_camera(const pnr_params* params, const pnr_mlp* coarse, const pnr_mlp* fine,
                          const pnr_views* views, const float* c2w /* host, 16 floats */, int32_t W, int32_t H,
                          float fx, float fy, float cx, float cy, float z_near, float z_far, int64_t pix0,
                          int64_t n_rays, const pnr_noise* noise, uint64_t seed, int64_t ray_index_base,
                          const pnr_outputs* outputs, void* workspace, uint64_t workspace_bytes, void* stream);

/* util.gen_rays for one camera (util/util.py:118-148,243-281): pixels [pix0, pix0+n) of a W x H pinhole image. */
int32_t pnr_gen_rays(const float* c2w /* host, 16 floats */, int32_t W, int32_t H, float fx, float fy,
                     float cx, float cy, float z_near, float z_far, int64_t pix0, int64_t n,
                     float* rays_out, void* stream);

/* Timing hook for bench.py: microseconds between the first and last point-MLP launch of the most recent
 * pnr_render on this thread is NOT kept (no global state); instead the caller brackets calls with
 * hipEvents on `stream`.  These two helpers expose hipEvent timing on an arbitrary hipStream_t to
 * ctypes callers (torch.cuda.Event only sees torch's current stream). */
int32_t pnr_event_create(void** ev);
int32_t pnr_event_record(void* ev, void* stream);
int32_t pnr_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on `stop` */
int32_t pnr_event_destroy(void* ev);

/* Diagnostics of the tile GEMMs behind the fp32 path and the training path (host functions, no device work): size of their
 * 1-D launch grid, and the tile a workgroup id maps to — (m tile, n tile, reduction split) in out3, return 1, or 0 for a
 * padding workgroup of the rounded-up grid.  The order is XCD-aware: workgroup ids go round the 8 XCDs, and the tiles that
 * share an operand take consecutive slots of ONE XCD (tests/test_host_cpu.py checks the bijection and that property). */
int64_t pnr_debug_gemm_grid(int32_t M, int32_t N, int32_t Rn, int32_t rows_per_split, int32_t split);
int32_t pnr_debug_gemm_tile(int32_t block, int32_t M, int32_t N, int32_t Rn, int32_t rows_per_split, int32_t split,
                            int32_t* out3);

#ifdef __cplusplus
}
#endif
#endif /* PNR_H */
