// Fused per-point network for gfx950 (CDNA4): PixelNeRFNet.forward (models.py.backup2:155-282) + ResnetFC
// (resnetfc.py:173-236) in ONE persistent kernel, bf16 or fp16 MFMA with fp32 accumulation.
//
// Shape of the computation (d_hidden = 512):
//   * a workgroup = 4 waves (one per SIMD, up to 512 registers each) owns a tile of 128 points; a wave owns 32 of
//     them = two 16-column groups of v_mfma_f32_16x16x32 (lane l: column c = l & 15, k-quarter g = l >> 4).
//   * activations are kept TRANSPOSED, X^T = [512 features x 32 points] fp32, as 32 x 2 accumulators of 16x16
//     (all 256 AGPRs) for the whole network: Y^T = W . X^T uses the PyTorch (out,in) weight as the A operand and the
//     previous layer's accumulators, converted in registers to 16 bit, as the B operand — the sum runs over the
//     accumulator's ROW index, so no lane movement and no LDS round trip for activations.
//   * the weights of the whole MLP are pre-packed (pnr_pack_mlp) into a linear stream of 1-KiB MFMA A-fragments
//     (16 rows x 32 k) in exactly the order the kernel consumes them; every fragment feeds TWO MFMAs (the two column
//     groups).  All 4 waves consume the same stream, staged through a 4-slot x 16-KiB LDS ring filled by LDS-DMA
//     (global_load_lds_dwordx4) three stages ahead, one barrier per stage of 32 MFMAs.  Every workgroup streams the same
//     bytes in the same order => L2-resident across the XCD.
//   * every MFMA is issued from the hand-scheduled asm blocks of resblock_asm.inc (tools/gen_resblock_asm.py); this
//     file holds the packers, the per-tile prologue (geometry, positional features, latent gather / tap weights into
//     the wave's LDS B-operand image) and the epilogue.
//   * multi-view: the first `combine_layer` blocks run once per source view on the same 128 points; the per-view
//     residual streams are parked in a caller-provided workspace and reduced (mean/max) in registers.
//   * render launch (RayJob, pnr_render / pnr_render_camera): workgroups own whole rays; rays come from the ray tensor or are
//     formed from a camera and the pixel index, coarse sample positions are generated in the tile prologue, and the rays a
//     tile finished are composited (composite_ray, pnr_common.h) by their workgroup at the start of the next tile — one
//     launch per pass, bit-identical to the stage kernels (NeRFRenderer.forward, render/nerf.py:98-118,176-249,251-303).
//   * nothing per-lane stays live across the asm blocks: lane index, divisors and launch-uniform switches are re-derived
//     through opaque copies at their uses (a value hipcc keeps across a block comes back through scratch, behind a
//     vmcnt(0) that also drains the run-ahead LDS-DMA).
#include <type_traits>

#include "pnr_common.h"
#include "resblock_asm.inc"

namespace pnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

static constexpr int HID = 512;          // d_hidden the kernel is specialised for
static constexpr int NT = HID / 32;      // 16 pinned register tiles of 16 accumulator registers
static constexpr int STAGE_BYTES = 16384;
static constexpr int RING_SLOTS = 4;     // power of two; stage i+3 is loaded while stage i is consumed and stage i+1 read ahead
static constexpr int TILE_PTS = 128;
static constexpr int ZBUF_BYTES = 16384; // per wave: 8 k-steps x 2 column groups x 64 lanes x 16 B (256 latent channels)
static constexpr int LDS_RING = 0;
static constexpr int PTS_BYTES = 1280;   // per wave and buffer: the tile's 32 rays (2 x 512 B) + 32 z (128 B, written twice), or 32 x (xyz, dirs)
static constexpr int LDS_Z = RING_SLOTS * STAGE_BYTES;
static constexpr int LDS_PTS = LDS_Z + 4 * ZBUF_BYTES;
static constexpr int LDS_UV = LDS_PTS + 4 * 2 * PTS_BYTES;      // per wave: 64 lanes x (u, v) of the lane's two points
static constexpr int LDS_BTAB = LDS_UV + 4 * 1024;
// behind the bias table (when it fits, RayJob.cmp_lds): the compositing ring — (rgb, sigma) and z of the workgroup's last
// CMP_RING points, written by the tile epilogue, read by the wave that composites a finished ray.  A ray of K <= CMP_MAX_K
// samples that ends in tile t starts after the end of tile t - 2, and is composited before tile t + 1's epilogue writes
// again: two tiles of slots are enough.
static constexpr int CMP_RING = 2 * TILE_PTS, CMP_MAX_K = TILE_PTS;
static constexpr int CMP_BYTES = CMP_RING * (16 + 4);
static constexpr int LDS_LIMIT = 160 * 1024;

// ---------------------------------------------------------------------------- 16-bit helpers
template <int DT> struct Num;
template <> struct Num<PNR_BF16> {
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        f32x2 f = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));      // v_cvt_pk_bf16_f32
    }
    static __device__ __forceinline__ float lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
    static __device__ __forceinline__ float hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
    static __device__ __forceinline__ uint16_t one() { return 0x3F80; }
    static __device__ __forceinline__ uint16_t cvt(float a) { return (uint16_t)(pack(a, 0.f) & 0xffff); }
    static __device__ __forceinline__ float back(uint16_t v) { return lo(v); }
};
template <> struct Num<PNR_F16> {
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        f32x2 f = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, f16x2));       // v_cvt_pk_f16_f32
    }
    static __device__ __forceinline__ float lo(uint32_t p) { return (float)__builtin_bit_cast(f16x2, p)[0]; }
    static __device__ __forceinline__ float hi(uint32_t p) { return (float)__builtin_bit_cast(f16x2, p)[1]; }
    static __device__ __forceinline__ uint16_t one() { return 0x3C00; }
    static __device__ __forceinline__ uint16_t cvt(float a) { return (uint16_t)(pack(a, 0.f) & 0xffff); }
    static __device__ __forceinline__ float back(uint16_t v) { return lo(v); }
};

// ---------------------------------------------------------------------------- stream layout (shared by pack + kernel)
// Stage = 16 fragments of 1 KiB; a k-step (32 k) of an x-stage = 2 stages (row groups 0-15, 16-31).
// Per (tile, view):  LIN_IN (S_in k-steps), then per phase-1 block [LIN_Z: SZ k-steps] [bias k-step] [64 chunk stages];
// phase 2: per block [bias k-step][64 chunk stages]; then [last fc_1 bias k-step][LIN_OUT 1 stage].
// The bias k-step at the head of block b carries lin_z[b].bias (b < nb1) + fc_1[b-1].bias (b > 0): both are added to
// the residual stream between the previous block's fc_1 and this block's relu, so they travel together.
// Projected mode (proj_T > 0; small last latent level of T texels): bilinear interpolation and lin_z are both linear,
// so lin_z_b(z(p)) = (W_z,b . Lat) . w(p) with w(p) the T-vector of the point's 4 tap weights.  The stream then
// carries the (512 x T) products M_b = W_z,b . Lat in place of W_z,b: ceil(T/32) k-steps instead of L/32, and the
// per-point latent gather disappears (the B operand is the tap-weight image).
struct Layout {
    int d_in, D, S_in, L, SZ, n_blocks, nb1, nb2, P1, P2, btab_floats, proj_T, ZK, PV, Gg;   // PV: per-view copies of the P1 part (projected), else 1; Gg: 256-channel groups still gathered
    int fold;      // projected: the bias of a block with lin_z rides on the columns of W_z . Lat (no bias stages in P1)
    uint64_t btab_bytes, stream_bytes, total_bytes, proj_bytes;
};
static constexpr int CHUNK_STAGES = 64;

__host__ __device__ inline bool make_layout(const pnr_mlp& m, Layout& y, int proj_T = 0, int proj_views = 1, int proj_gathered = 0) {
    if (m.d_hidden != HID || m.d_out != 4 || m.d_latent <= 0 || (m.d_latent % 256) != 0 || m.d_latent > 1024) return false;
    if (m.n_blocks < 1 || m.n_blocks > PNR_MAX_BLOCKS || m.d_in < 1 || m.d_in > 78) return false;
    // LIN_IN k layout (chosen for the kernel, see the prologue): lane group g (k-quarter) of k-step s, element j holds the
    // group-local slot 8 s + j of
    //   [ sine n = 3 D g + jj, jj = 0..3D-1 | other_0, other_1 | lo(other_0), lo(other_1) | 0.. ]
    // sine n = sin(f_q v_i + phase pi/2) with n = 2 D q + D phase + i (code.py's own order), D = 3 (code over xyz, d_in 42) or
    // 6 (code over xyz + dirs, d_in 78), 6 frequencies only; others: raw input 2 g + o for g < 3 (x_rot, then the rotated
    // view dir), the folded lin_in bias (hi, lo as two 1.0 inputs) for g = 3; lo(v) = v - 16-bit(v) doubles the mantissa of
    // the raw coordinates.
    y.d_in = m.d_in;
    if (m.d_in != 42 && m.d_in != 78) return false;
    y.D = m.d_in == 78 ? 6 : 3;
    y.S_in = (3 * y.D + 4 + 7) / 8;                  // 2 or 3 k-steps of 32
    y.L = m.d_latent;
    // projected: the LAST latent level (256 channels, T <= 256 texels) is folded into lin_z; the Gg 256-channel groups of
    // the levels before it are still gathered.  lin_z k-steps (32 wide): 8 per gathered group, then ceil(T/32) texel steps.
    if (proj_T < 0 || proj_T > 256 || proj_gathered < 0) return false;
    if (proj_T > 0 && m.d_latent != 256 * (proj_gathered + 1)) return false;
    y.proj_T = proj_T;
    y.Gg = proj_T > 0 ? proj_gathered : 0;
    y.ZK = proj_T > 0 ? ((proj_T + 31) / 32) * 32 : 0;  // texel extent (padded) of the projected part
    y.SZ = proj_T > 0 ? 8 * y.Gg + y.ZK / 32 : m.d_latent / 32;
    y.n_blocks = m.n_blocks;
    y.nb1 = m.combine_layer < m.n_blocks ? m.combine_layer : m.n_blocks;
    if (y.nb1 < 0) y.nb1 = 0;
    y.nb2 = m.n_blocks - y.nb1;
    // projected: the four tap weights of a point sum to 1, so lin_z.bias[b] + fc_1.bias[b-1] is added to every texel column
    // of W_z,b . Lat and the bias k-step of the blocks with lin_z disappears (2 stages per block)
    y.fold = proj_T > 0 ? 1 : 0;
    y.P1 = 2 * y.S_in + y.nb1 * (2 * y.SZ + (y.fold ? 0 : 2) + CHUNK_STAGES);
    y.P2 = y.nb2 * (2 + CHUNK_STAGES) + 2 + 1;
    y.btab_floats = ((m.n_blocks * HID + 4 + 63) / 64) * 64;
    y.btab_bytes = (uint64_t)y.btab_floats * 4;
    // projected: the per-view part of the stream is materialised once per source view ([P1 view 0][P1 view 1]..[P2]) with
    // that view's M_b inside, so a tile consumes ONE linear stream of PV*P1 + P2 stages
    y.PV = proj_T > 0 ? proj_views : 1;
    if (y.PV < 1 || y.PV * y.P1 + y.P2 > 4095) return false;
    y.stream_bytes = (uint64_t)(y.PV * y.P1 + y.P2) * STAGE_BYTES;
    y.proj_bytes = proj_T > 0 ? (uint64_t)y.PV * (y.nb1 > 0 ? y.nb1 : 1) * HID * y.ZK * 4 : 0;   // fp32 M_{v,b} scratch behind the stream
    y.total_bytes = y.btab_bytes + y.stream_bytes + y.proj_bytes;
    return true;
}

// k order of a B operand built from accumulators (16x16x32): element j of lane group g in a 32-wide k-step is row
// (j < 4 ? 4 g + j : 16 + 4 g + j - 4) of that k-step (accumulator register j & 3 of row group 2 ks + (j >> 2))
__host__ __device__ inline int perm_k(int g, int j) { return (j < 4) ? 4 * g + j : 16 + 4 * g + (j - 4); }

// ---------------------------------------------------------------------------- pack kernels
// M_b[n][t] = sum_c lin_z[b].weight[n][c] * latent[c][t]  (fp32; last latent level), t padded to ZK with 0
__global__ void k_project_latent(pnr_mlp m, Layout y, const float* __restrict__ lat, int T, float* __restrict__ M) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n_out = (int64_t)y.PV * y.nb1 * HID * y.ZK;                 // M[view][block][n][t]
    if (i >= n_out) return;
    int t = (int)(i % y.ZK);
    int n = (int)((i / y.ZK) % HID);
    int b = (int)((i / ((int64_t)y.ZK * HID)) % y.nb1);
    int v = (int)(i / ((int64_t)y.ZK * HID * y.nb1));
    float acc = 0.f;
    if (t < T) {
        const int c0 = 256 * y.Gg, Cl = y.L - c0;            // channels of the projected (last) level
        const float* w = m.lin_z_w[b] + (size_t)n * y.L + c0;
        const float* lv = lat + (size_t)v * Cl * T;
        for (int c = 0; c < Cl; ++c) acc = fmaf(w[c], lv[(size_t)c * T + t], acc);
    }
    M[i] = acc;
}

template <int DT>
__global__ void k_pack_mlp(pnr_mlp m, Layout y, char* __restrict__ out, const float* __restrict__ M, uint64_t stream_off = 0) {
    // bias table (fc_0 biases: chunk-accumulator init; lin_out bias: epilogue)
    float* bt = (float*)out;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < y.btab_floats; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < m.n_blocks * HID) v = m.fc0_b[i / HID][i % HID];
        else if (i < m.n_blocks * HID + 4) v = m.lin_out_b[i - m.n_blocks * HID];
        bt[i] = v;
    }
    uint16_t* st = (uint16_t*)(out + y.btab_bytes + stream_off);       // projected: one stream per object
    const int64_t n_elems = (int64_t)(y.PV * y.P1 + y.P2) * 16 * 64 * 8;
    const int nbias1 = y.fold ? 0 : 2;               // bias stages at the head of a block with lin_z
    const int per1 = 2 * y.SZ + nbias1 + CHUNK_STAGES, per2 = 2 + CHUNK_STAGES;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_elems; e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), f = (int)((e >> 9) & 15);
        int stage = (int)(e >> 13);
        int pview = 0;                                   // which per-view copy (projected streams)
        if (stage < y.PV * y.P1) { pview = stage / y.P1; stage -= pview * y.P1; }
        else stage -= (y.PV - 1) * y.P1;                // phase 2 follows the last copy
        const int row = lane & 15, g = lane >> 4;
        float val = 0.f;
        // x-stage fragment: k-step ks, half hf -> output row n = 16 (16 hf + f) + row, natural k = 32 ks + 8 g + j
        // kind: 0 LIN_IN, 1 LIN_Z (block b), 2 bias (head of block b; b == n_blocks: the last fc_1 bias), 3 chunk stage q of block b, 4 LIN_OUT
        int kind, b = 0, xs = 0, q = 0;                  // xs: x-stage index (2 ks + hf) within its group
        if (stage < 2 * y.S_in) { kind = 0; xs = stage; }
        else if (stage < y.P1) {
            int s = stage - 2 * y.S_in;
            b = s / per1; s %= per1;
            if (s < 2 * y.SZ) { kind = 1; xs = s; }
            else if (s < 2 * y.SZ + nbias1) { kind = 2; xs = s - 2 * y.SZ; }
            else { kind = 3; q = s - 2 * y.SZ - nbias1; }
        } else {
            int s = stage - y.P1;
            if (s < y.nb2 * per2) {
                b = y.nb1 + s / per2; s %= per2;
                if (s < 2) { kind = 2; xs = s; } else { kind = 3; q = s - 2; }
            } else {
                s -= y.nb2 * per2;
                if (s < 2) { kind = 2; b = y.n_blocks; xs = s; } else kind = 4;
            }
        }
        const int n = 16 * (16 * (xs & 1) + f) + row, ks = xs >> 1;
        if (kind == 0) {                                 // LIN_IN: group-local slot layout of make_layout
            const int D = y.D, sl = 8 * ks + j;
            if (sl < 3 * D) {
                val = m.lin_in_w[(size_t)n * m.d_in + D + 3 * D * g + sl];               // code.py: [x (D), sines (12 D)]
            } else if (sl < 3 * D + 4) {
                const int o = (sl - 3 * D) & 1, is_lo = (sl - 3 * D) >> 1;
                if (g < 3) {
                    const int ri = 2 * g + o;            // raw input: x_rot 0..2, rotated view dir 3..5
                    const int col = (D == 6 || ri < 3) ? ri : 3 + 12 * D + (ri - 3);    // D = 3: the dirs follow the code
                    val = m.lin_in_w[(size_t)n * m.d_in + col];
                } else if (!is_lo) {                     // folded bias: two inputs of 1.0 carry (hi, lo) of lin_in.bias
                    const float bv = m.lin_in_b[n];
                    val = o == 0 ? bv : bv - Num<DT>::back(Num<DT>::cvt(bv));
                }
            }
        } else if (kind == 1) {                          // LIN_Z k-step ks: natural k
            const int k = 32 * ks + 8 * g + j;
            if (M && ks >= 8 * y.Gg) {
                const int t = k - 256 * y.Gg;            // texel column of the projected level
                val = M[(((size_t)pview * y.nb1 + b) * HID + n) * y.ZK + t];
                if (y.fold && t < y.proj_T) {            // the block's bias on every real texel column (tap weights sum to 1)
                    val += m.lin_z_b[b][n];
                    if (b > 0) val += m.fc1_b[b - 1][n];
                }
            } else {
                val = m.lin_z_w[b][(size_t)n * y.L + k];
            }
        } else if (kind == 2) {                          // bias k-step: k-slot 0 = hi, 1 = lo
            if (g == 0 && j < 2) {
                float bv = 0.f;
                if (b < y.nb1) bv += m.lin_z_b[b][n];
                if (b > 0) bv += m.fc1_b[b - 1][n];
                val = j == 0 ? bv : bv - Num<DT>::back(Num<DT>::cvt(bv));
            }
        } else if (kind == 3) {
            // 64 chunk stages in the kernel's software-pipelined order: F(0) | F(1) G(0) | ... | F(15) G(14) | G(15),
            // F(c) = the 2 fc_0 stages of chunk c (parts 0,1), G(c) = its 2 fc_1 stages (parts 2,3)
            int c, part;
            if (q < 2) { c = 0; part = q; }
            else if (q >= 62) { c = 15; part = 2 + (q - 62); }
            else {
                int jq = q - 2, cc = jq >> 2, pp = jq & 3;
                if (pp < 2) { c = cc + 1; part = pp; } else { c = cc; part = pp; }
            }
            if (part < 2) {                             // fc_0 chunk c: fragment f = (row group f & 1 of the chunk, k-step 8 part + f / 2)
                const int kk = 8 * part + (f >> 1);
                val = m.fc0_w[b][(size_t)(32 * c + 16 * (f & 1) + row) * HID + 32 * kk + perm_k(g, j)];
            } else {                                    // fc_1 chunk c: output row group 16 (part - 2) + f, k = the chunk's 32 rows of h
                val = m.fc1_w[b][(size_t)(16 * (16 * (part - 2) + f) + row) * HID + 32 * c + perm_k(g, j)];
            }
        } else {                                        // LIN_OUT: fragment f = k-step f, rows 0..3 valid
            if (row < 4) val = m.lin_out_w[(size_t)row * HID + 32 * f + perm_k(g, j)];
        }
#ifdef PNR_DIAG_WMASK      // experiment only (tools/dev): weights rounded to 10 - PNR_DIAG_WMASK mantissa bits of the 16-bit format
        {
            uint32_t hb = Num<DT>::cvt(val);
            hb = (hb + (1u << (PNR_DIAG_WMASK - 1))) & ~((1u << PNR_DIAG_WMASK) - 1u) & 0xffffu;
            st[e] = (uint16_t)hb;
            continue;
        }
#endif
        st[e] = Num<DT>::cvt(val);
    }
}

struct LatPack { uint64_t off[PNR_MAX_LEVELS]; uint64_t total; };
__host__ __device__ inline LatPack lat_layout(const pnr_views& v) {
    LatPack p; uint64_t o = 0;
    int nv = v.n_objs * v.n_views;
    for (int i = 0; i < PNR_MAX_LEVELS; ++i) {
        p.off[i] = o;
        if (i < v.n_levels) o += (((uint64_t)nv * v.lat_c[i] * v.lat_h[i] * v.lat_w[i] * 2) + 255) & ~(uint64_t)255;
    }
    p.total = o;
    return p;
}

template <int DT>
__global__ void k_pack_latents(pnr_views v, LatPack lp, char* __restrict__ out) {
    int nv = v.n_objs * v.n_views;
    for (int lvl = 0; lvl < v.n_levels; ++lvl) {
        int C = v.lat_c[lvl], HW = v.lat_h[lvl] * v.lat_w[lvl];
        int64_t n = (int64_t)nv * C * HW;
        uint16_t* o = (uint16_t*)(out + lp.off[lvl]);
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
            int c = (int)(e % C); int64_t t = e / C; int px = (int)(t % HW); int view = (int)(t / HW);
            o[e] = Num<DT>::cvt(v.latent[lvl][((size_t)view * C + c) * HW + px]);     // (view, px, c) <- (view, c, px)
        }
    }
}

// ---------------------------------------------------------------------------- the fused kernel
struct MfmaArgs {
    pnr_views vw;
    PointSrc src;
    int64_t n_points, pts_per_obj;
    const char* stream;
    const float* btab;
    const char* lat[PNR_MAX_LEVELS];   // packed latents per level
    float* out;
    float4* spill;                 // (grid, 4 waves, NS-1, 64 x 64) float4
    char* gcache;                  // (grid, 4 waves, n_cached, 16 KiB): a view's gathered lin_z images, kept for its later blocks
    int n_cached;
    int n_tiles, NS, combine_max;
    int park32;                    // multi-view: park the per-view streams as fp32 (pnr_params.park_fp32) instead of the kernel's 16-bit format
    int S_in, SZ, n_blocks, nb1, P1, P2, btab_floats, d_in, proj, Gg;
    int ldP1, ldNS;                // the loader's stream: (P1, NS), or (NS*P1, 1) when every view has its own copy (projected)
    int use_code_viewdirs, num_freqs;
    float freq_factor;
    int wgs_per_obj;               // > 0: workgroups are assigned per OBJECT (projected streams of several objects: a tile never
    int64_t obj_stream_stride;     //      mixes objects); object o streams from stream + o * obj_stream_stride
    int tiles_per_wg;              // plain launch: workgroup b owns tiles [b tiles_per_wg, (b+1) tiles_per_wg)
    RayJob job;                    // fused render launch: workgroup b owns rays [b rays_per_wg, ...) (see pnr_common.h)
};

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// LDS-DMA from C++ (global_load_lds_*): inline asm so that hipcc neither counts these loads in its own vmcnt bookkeeping
// nor drains them before barriers / ds_reads; they are retired by the counted s_waitcnt of the asm blocks
// (cdna_hip_programming.md §5.7).  M0 (the LDS destination base) is compiler-reserved, so a statement saves, sets and
// restores it.  A piece reads M0 when it ISSUES (measured: tools/dev/ubench/waw_ubench.hip part C — M0 rewritten in the
// next instruction, with or without older loads queued, and every byte still lands at the original destination), so the
// restore may follow the last piece directly; `s_mov m0` -> piece needs its one documented wait state.  (Round 2 blamed
// M0 for wrong results of a one-statement-per-piece variant; that was the statement-entry hazard of DESIGN.md 4.1.)
// These statements write no VGPR, so rule R1 of tools/gen_resblock_asm.py does not apply to them.  Offsets apply to the
// global AND the LDS address.
//
// One stage of the weight ring (only the first three stages are issued from here; the steady state lives in
// resblock_asm.inc): lane l moves 4 x 16 B from g_base + lane_off + 1024 q to LDS lds_dst + 1024 q + 16 l.
__device__ __forceinline__ void glds_stage(const char* g_base /* wave-uniform */, uint32_t lane_off, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(g_base), "s"(lds_dst) : "memory");
}

// 4 KiB of a wave's own workspace data back into LDS (the cached lin_z images): as glds_stage, but through the L2 — the lines
// were written by this wave a block ago and read a tile ago, so the CU's L1 may hold their previous contents.
__device__ __forceinline__ void glds_stage_l2(const char* g_base /* wave-uniform */, uint32_t lane_off, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 sc1\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 sc1\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048 sc1\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072 sc1\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(g_base), "s"(lds_dst) : "memory");
}

// Per-lane-source LDS-DMA (a gather): fetches the NEXT tile's rays / sample positions a whole tile ahead of their use,
// without registers and without a wait: they are older than the weight pieces issued after them, so the counted waits of
// the asm blocks retire them long before the next tile.  Rays mode: lane l moves 16 B of its ray to dst + 16 l and its z to
// dst + 1024 + 4 l (pz = &z[point] - 1024 bytes: the instruction offset also moves the source).
__device__ __forceinline__ void glds_gather_ray(const float* pr, const char* pz_m1024, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dword %2, off offset:1024\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(pr), "v"(pz_m1024), "s"(lds_dst) : "memory");
}
// the two halves of glds_gather_ray on their own (positions generated in the kernel / rays generated from a camera)
__device__ __forceinline__ void glds_gather_ray_only(const float* pr, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(pr), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds_gather_z_only(const char* pz_m1024, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, off offset:1024\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(pz_m1024), "s"(lds_dst) : "memory");
}
// Explicit points: lane l moves the three components of its 12-byte record to dst + 256 k + 4 l (p = &record[0]).
__device__ __forceinline__ void glds_gather_xyz(const char* p, uint32_t lds_dst) {
    uint32_t keep;
    const char* p1 = p + 4 - 256;
    const char* p2 = p + 8 - 512;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, off\n\tglobal_load_lds_dword %2, off offset:256\n\tglobal_load_lds_dword %3, off offset:512\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "v"(p1), "v"(p2), "s"(lds_dst) : "memory");
}

// Diagnostic build only (-DPNR_STAMPS): per-section shader-cycle sums, never part of the product library.
#ifdef PNR_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_ACC(idx, t0v) do { unsigned long long _t; STAMP(_t); st_acc[idx] += _t - (t0v); (t0v) = _t; } while (0)
#else
#define STAMP(var) do { } while (0)
#define STAMP_ACC(idx, t0v) do { } while (0)
#endif

template <int DT, bool MULTIVIEW>
__global__ void __launch_bounds__(256, 1) k_point_mfma(MfmaArgs a) {
#ifdef PNR_STAMPS
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = 0, st_tile = 0;
    const unsigned long long st_k0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef Num<DT> NM;
    // fp16: MODE.FP16_OVFL (bit 23 of HW_REG_MODE) for the whole kernel — a float -> fp16 conversion whose finite input
    // overflows gives +-65504 instead of +-inf (a true inf stays inf), so the activations saturate inside v_cvt_pk_f16_f32 and
    // the snapshot needs no v_pk_min per converted pair (tools/dev/ubench/ovfl_ubench.hip shows the behaviour on gfx950).
    // hwreg(HW_REG_MODE = 1, offset 23, width 1) = 1 | 23 << 6 | 0 << 11.  The mode register is per wave and dies with it.
    if (DT == PNR_F16) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Nothing per-lane stays live across the asm blocks (they clobber all but ~40 VGPRs, and a spilled value comes back
    // through a scratch load whose vmcnt(0) also drains the run-ahead LDS-DMA: ~2 us each under the kernel's load): every
    // use site re-derives its lane index here, opaque to the optimiser so that it is not hoisted to the kernel's entry.
    auto lane_id = []() __attribute__((always_inline)) -> int {
        int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(l));
        return l;
    };
    float* btab = (float*)(smem + LDS_BTAB);
    for (int i = tid; i < a.btab_floats; i += 256) btab[i] = a.btab[i];
    __syncthreads();

    // ---------------- weight-stream loader cursor: stage index in the packed stream, source-view pass it belongs to, and where
    // that pass wraps to 0 (a non-last view repeats phase 1; the last view runs on into phase 2).  Every wave issues 4 x 1 KiB
    // LDS-DMA per stage; the first three stages from here, the rest from the asm blocks (which carry the cursor on).
    int ld_idx = 0, ld_rep = 0, ld_slot = 0, st_slot = 0;
    int ld_wrap = (a.ldNS == 1) ? a.ldP1 + a.P2 : a.ldP1;
    const uint32_t ring_lds = lds_addr(smem + LDS_RING) + wv * 4096;
    // 64-bit divisions are ~100 instructions each on this machine: point counts below 2^31 take a 32-bit path.  The divisor
    // goes through an opaque copy so that its reciprocal is formed at the use, not kept from the kernel's entry in scratch.
    auto div_pts = [&](int64_t num, int64_t den) __attribute__((always_inline)) -> int64_t {
        if (a.n_points < 0x7fffffffLL) {
            uint32_t d32 = (uint32_t)den;
            asm volatile("" : "+s"(d32));
            return (int64_t)((uint32_t)num / d32);
        }
        asm volatile("" : "+s"(den));          // (the 64-bit expansion has a 32-bit fast path with a hoistable reciprocal too)
        return num / den;
    };
    // ---------------- this workgroup's points [p_begin, p_end): whole rays in a fused render launch (so that a ray is composited
    // by the workgroup that evaluated it), whole tiles otherwise.  The host keeps p_end - p_begin below 2^31.
    // Launch-uniform switches are read through an opaque copy where they are tested: hipcc otherwise evaluates them once at the
    // kernel's entry, keeps the 0/1 results in VECTOR registers (the scalar file is full) and, those being clobbered by the
    // asm blocks, reloads them from scratch in every tile — with a vmcnt(0) that also drains the run-ahead LDS-DMA.
    auto flag = [](int v) __attribute__((always_inline)) -> bool {
        v = __builtin_amdgcn_readfirstlane(v);
        asm volatile("" : "+s"(v));
        return v != 0;
    };
    auto flagp = [](const void* q) __attribute__((always_inline)) -> bool {
        asm volatile("" : "+s"(q));
        return q != nullptr;
    };
    int64_t p_begin, p_end, ray_begin = 0;
    // per-object assignment (several objects with projected streams): workgroup = (object, index within the object)
    const int wg_obj = a.wgs_per_obj > 0 ? (int)blockIdx.x / a.wgs_per_obj : 0;
    const int wg_idx = a.wgs_per_obj > 0 ? (int)blockIdx.x - wg_obj * a.wgs_per_obj : (int)blockIdx.x;
    const char* const stream = a.stream + (size_t)wg_obj * a.obj_stream_stride;
    if (a.job.on) {
        const int64_t rays_lim = a.wgs_per_obj > 0 ? (wg_obj + 1) * (a.pts_per_obj / a.job.K) : a.job.n_rays;
        ray_begin = (a.wgs_per_obj > 0 ? wg_obj * (a.pts_per_obj / a.job.K) : 0) + (int64_t)wg_idx * a.job.rays_per_wg;
        ray_begin = ray_begin < rays_lim ? ray_begin : rays_lim;
        int64_t ray_end = ray_begin + a.job.rays_per_wg;
        ray_end = ray_end < rays_lim ? ray_end : rays_lim;
        p_begin = ray_begin * a.job.K;
        p_end = ray_end * a.job.K;
    } else {
        const int64_t pts_lim = a.wgs_per_obj > 0 ? (wg_obj + 1) * a.pts_per_obj : a.n_points;
        p_begin = (a.wgs_per_obj > 0 ? wg_obj * a.pts_per_obj : 0) + (int64_t)wg_idx * a.tiles_per_wg * TILE_PTS;
        p_begin = p_begin < pts_lim ? p_begin : pts_lim;
        p_end = p_begin + (int64_t)a.tiles_per_wg * TILE_PTS;
        p_end = p_end < pts_lim ? p_end : pts_lim;
    }
    const int n_loc = (int)(p_end - p_begin);
    const int my_tiles = (n_loc + TILE_PTS - 1) / TILE_PTS;
    // ---------------- point inputs of a tile -> this wave's LDS buffer `buf` (0/1), by LDS-DMA with per-lane sources.
    // rays mode: [half][point][16 B] = (o, d.x | d.yz, near, far) at 0 / 512, z at 1024 + 4 point; explicit points: component k
    // of xyz at 256 k + 4 point, of dirs at 256 k + 128 + 4 point.  Point pl = lane & 31 of the wave; indices clamped to the last point.
    char* pts_wave = smem + LDS_PTS + wv * (2 * PTS_BYTES);
    auto prefetch_points = [&](int tile, int buf) __attribute__((always_inline)) {
        const int lane = lane_id();
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr(pts_wave) + buf * PTS_BYTES);
        int lp = tile * TILE_PTS + wv * 32 + (lane & 31);            // index within the workgroup's range
        lp = lp < n_loc ? lp : n_loc - 1;
        const int64_t gp = p_begin + lp;
        if (flag(a.job.on)) {
            int Kq = a.job.K;
            asm volatile("" : "+s"(Kq));
            const int64_t ray = ray_begin + (uint32_t)lp / (uint32_t)Kq;
            if (!flag(a.job.from_cam)) glds_gather_ray_only(a.src.rays + ray * 8 + (lane >> 5) * 4, dst);
            if (!flag(a.job.gen_z)) glds_gather_z_only((const char*)(a.src.z + gp) - 1024, dst);
        } else if (flagp(a.src.rays)) {
            const int64_t ray = div_pts(gp, a.src.K);
            glds_gather_ray(a.src.rays + ray * 8 + (lane >> 5) * 4, (const char*)(a.src.z + gp) - 1024, dst);
        } else {
            glds_gather_xyz((const char*)((lane < 32 ? a.src.xyz : a.src.dirs) + gp * 3), dst);     // 12-byte records
        }
    };
    if (flag(a.job.resample)) {
        // ---------------- fine pass: resample this workgroup's rays first (one wave per ray, the wave's 16-KiB B-image buffer as
        // scratch), the function k_sample_fine runs.  The positions go to z_fine, which the tile prefetch then gathers from:
        // the stores are drained and the workgroup synchronised before the first gather (nothing of z_fine is in this CU's L1).
        const FineArgs f = a.job.fine;
        const int Kt = f.Kc + f.n_imp + f.n_dep;
        float* cdf = (float*)(smem + LDS_Z + wv * ZBUF_BYTES);
        float* buf = cdf + f.Kc + 2;
        const int lane = lane_id();
        const int n_rays_loc = n_loc / Kt;
        auto wave_sync = []() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's LDS accesses are ordered; this orders the compiler
        };
        for (int lr = wv; lr < n_rays_loc; lr += 4) {
            const int64_t ray = ray_begin + lr;
            const float near = flag(a.job.from_cam) ? a.job.cam.zn : a.src.rays[ray * 8 + 6];
            const float far = flag(a.job.from_cam) ? a.job.cam.zf : a.src.rays[ray * 8 + 7];
            const float depth = f.n_dep > 0 ? a.job.depth_c[ray * a.job.dc_stride] : 0.f;
            sample_fine_ray<false>(f, a.job.zc + ray * f.Kc, a.job.wc ? a.job.wc + ray * a.job.wc_stride : nullptr, depth, near, far, ray,
                                   global_ray(a.job.key, ray), true, cdf, buf, a.job.z_fine + ray * Kt, lane, wave_sync);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (my_tiles > 0) prefetch_points(0, 0);
    {
        const uint32_t gl_off = (uint32_t)(wv * 4096 + lane_id() * 16);
        const char* dma_g = stream;                                 // global base of the stage being loaded (uniform)
        uint32_t dma_l = __builtin_amdgcn_readfirstlane(ring_lds);  // LDS base of its slot (+ this wave's quarter)
#pragma unroll
        for (int i = 0; i < RING_SLOTS - 1; ++i) {
            glds_stage(dma_g, gl_off, dma_l);
            ld_slot = (ld_slot + 1) & (RING_SLOTS - 1);
            ++ld_idx;
            if (ld_idx == ld_wrap) {
                ld_idx = 0;
                ld_rep = (ld_rep + 1 == a.ldNS) ? 0 : ld_rep + 1;
                ld_wrap = (ld_rep == a.ldNS - 1) ? a.ldP1 + a.P2 : a.ldP1;
            }
            dma_g = stream + (size_t)ld_idx * STAGE_BYTES;
            dma_l = __builtin_amdgcn_readfirstlane(ring_lds + ld_slot * STAGE_BYTES);
        }
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // stage 0 landed (stages 1, 2 may be in flight)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    f32x16 x[NT];          // X^T: the 256 accumulator registers (see the header of tools/gen_resblock_asm.py for the layout)
    const uint32_t one2 = (uint32_t)NM::one() | ((uint32_t)NM::one() << 16);
    char* zwave = smem + LDS_Z + wv * ZBUF_BYTES;
    float4* uvw = (float4*)(smem + LDS_UV + wv * 1024);

    // accumulator tiles pinned to a[16t : 16t+15] + the loader/consumer cursor: operand list shared by the hand-scheduled
    // blocks of resblock_asm.inc (tools/gen_resblock_asm.py documents the contract)
#define PNR_ASM_STATE_OPERANDS                                                                                          \
    "+{a[0:15]}"(x[0]), "+{a[16:31]}"(x[1]), "+{a[32:47]}"(x[2]), "+{a[48:63]}"(x[3]), "+{a[64:79]}"(x[4]),               \
    "+{a[80:95]}"(x[5]), "+{a[96:111]}"(x[6]), "+{a[112:127]}"(x[7]), "+{a[128:143]}"(x[8]), "+{a[144:159]}"(x[9]),      \
    "+{a[160:175]}"(x[10]), "+{a[176:191]}"(x[11]), "+{a[192:207]}"(x[12]), "+{a[208:223]}"(x[13]),                    \
    "+{a[224:239]}"(x[14]), "+{a[240:255]}"(x[15]), "+s"(st_slot), "+s"(ld_idx), "+s"(ld_slot), "+s"(ld_rep), "+s"(ld_wrap)
#define PNR_ASM_STATE_OPERANDS_OUT                                                                                      \
    "={a[0:15]}"(x[0]), "={a[16:31]}"(x[1]), "={a[32:47]}"(x[2]), "={a[48:63]}"(x[3]), "={a[64:79]}"(x[4]),               \
    "={a[80:95]}"(x[5]), "={a[96:111]}"(x[6]), "={a[112:127]}"(x[7]), "={a[128:143]}"(x[8]), "={a[144:159]}"(x[9]),      \
    "={a[160:175]}"(x[10]), "={a[176:191]}"(x[11]), "={a[192:207]}"(x[12]), "={a[208:223]}"(x[13]),                    \
    "={a[224:239]}"(x[14]), "={a[240:255]}"(x[15]), "+s"(st_slot), "+s"(ld_idx), "+s"(ld_slot), "+s"(ld_rep), "+s"(ld_wrap)
    const int asm_cfg = a.ldP1 | ((a.ldP1 + a.P2) << 12) | (a.ldNS << 24);
    // per-lane operands of the asm blocks, re-derived at every call (see lane_id): ring read base, DMA lane offset, B-image
    // address, B fragment of a bias k-step (k-slots 0,1 = 1.0 in the lanes of k-quarter 0 pick (hi, lo))
#define PNR_LANE_OPERANDS                                                                   \
    const int ln_ = lane_id();                                                              \
    const uint32_t ring_lane = lds_addr(smem + LDS_RING) + ln_ * 16;                        \
    const uint32_t gl_off = (uint32_t)(wv * 4096 + ln_ * 16);                               \
    const uint32_t zaddr = lds_addr(zwave) + ln_ * 16;                                      \
    const uint32_t bias_dword = (ln_ >> 4) == 0 ? one2 : 0u
    // n k-steps whose B pair is image [k-step][column group][lane] in this wave's LDS buffer
    auto x_stages = [&](int n_ksteps) __attribute__((always_inline)) {
        const int cfg2 = n_ksteps;
        PNR_LANE_OPERANDS;
        if (DT == PNR_BF16)
            asm volatile(PNR_XSTAGES_ASM_BF16 : PNR_ASM_STATE_OPERANDS
                         : "s"(asm_cfg), "s"(stream), "s"(ring_lds), "v"(ring_lane), "v"(gl_off), "v"(zaddr), "v"(bias_dword), "s"(cfg2)
                         : PNR_RESBLOCK_CLOBBERS);
        else
            asm volatile(PNR_XSTAGES_ASM_F16 : PNR_ASM_STATE_OPERANDS
                         : "s"(asm_cfg), "s"(stream), "s"(ring_lds), "v"(ring_lane), "v"(gl_off), "v"(zaddr), "v"(bias_dword), "s"(cfg2)
                         : PNR_RESBLOCK_CLOBBERS);
    };

    const int n_gather = a.proj ? a.Gg : a.SZ / 8;        // 256-channel groups gathered per block (the LDS image holds one)
    const int p_steps = a.proj ? a.SZ - 8 * a.Gg : 0;     // texel k-steps of the projected last level
    // the last lin_z call of a block (a gathered group, or the projected part) runs as the prefix of the resblock asm
    const int n_groups = a.proj ? n_gather : n_gather - 1;  // gathered groups that go through separate x_stages calls
    const bool one_part = n_groups == 0;                    // a block has ONE lin_z part: its B image is written once per view
    // ---------------- compositing of this workgroup's finished rays (fused render launch): ray lr of the workgroup by wave
    // lr & 3, from what the waves stored to `out` (and z_out) — through the L2: see ld_f in pnr_common.h
    const float4* cmp_c = (const float4*)(smem + LDS_BTAB + a.btab_floats * 4);      // the compositing ring (cmp_lds)
    const float* cmp_z = (const float*)(cmp_c + CMP_RING);
    auto composite_rays = [&](int lr0, int lr1) __attribute__((always_inline)) {
        const int lane = lane_id();
        const int K = a.job.K;
        for (int lr = lr0 + ((wv - lr0) & 3); lr < lr1; lr += 4) {
            const int64_t ray = ray_begin + lr;
            const float far = flag(a.job.from_cam) ? a.job.cam.zf : a.src.rays[ray * 8 + 7];
            float* wr = a.job.w_out ? a.job.w_out + ray * a.job.w_stride : nullptr;
            float4 r;
            if (flag(a.job.cmp_lds)) {
                // on-chip route: sample k of the workgroup's ray lr is its point lr K + k, ring slot (lr K + k) mod CMP_RING
                const int p0 = lr * K;
                r = composite_ray_t([=](int k) { return cmp_z[(p0 + k) & (CMP_RING - 1)]; },
                                    [=](int k) { return cmp_c[(p0 + k) & (CMP_RING - 1)]; }, K, far, a.job.white_bkgd, wr, lane);
            } else {
                const float* zr = (flag(a.job.gen_z) ? a.job.z_out : a.src.z) + ray * K;
                r = composite_ray<true>(zr, (const float4*)a.out + ray * K, K, far, a.job.white_bkgd, wr, lane);
            }
            if (lane == 0) {
                float* po = a.job.rgb_out + ray * a.job.rgb_stride;
                po[0] = r.x; po[1] = r.y; po[2] = r.z;
                a.job.depth_out[ray * a.job.depth_stride] = r.w;
            }
        }
    };
    int rays_done = 0;     // rays of the workgroup composited so far
    int pbuf = 0;
    for (int tile = 0; tile < my_tiles; ++tile, pbuf ^= 1) {
        STAMP(st_t);
#ifdef PNR_STAMPS
        st_tile = st_t;
#endif
        // this tile's point inputs landed a tile ago (or before the ring prologue's wait); fetch the next tile's now
        if (tile + 1 < my_tiles) prefetch_points(tile + 1, pbuf ^ 1);
        const char* pts = pts_wave + pbuf * PTS_BYTES;
        // lane (g, c) serves points (cg, c), cg = 0, 1: tile-local index wv*32 + 16 cg + c.  Source-view index of such a point
        // for view pass v (one object: v itself, wave-uniform):
        auto view_of = [&](int cc, int cg, int vv) __attribute__((always_inline)) -> int {
            if (!flag(a.vw.n_objs - 1)) return vv;          // tested where it is used (see flag)
            int li = tile * TILE_PTS + wv * 32 + 16 * cg + cc;
            li = li < n_loc ? li : n_loc - 1;
            const int64_t gi = p_begin + li;
            const int ob = (int)div_pts(gi, a.pts_per_obj);
            return ob * a.NS + vv;
        };
        int v = 0;      // current source-view pass
        // ---- latent gather: group grp (256 channels) -> this wave's LDS B image [k-step][column group][lane][8]:
        //      lane (g, c) interpolates channels 32 ks + 8 g .. + 7 of its two points
        // With several lin_z parts per block the one LDS image is rebuilt for every part of every block: the gather runs for
        // the view's first block only and leaves a copy in the workspace (`keep`), which the later blocks fetch back by
        // LDS-DMA (restore) instead of interpolating again.
        char* cache_wave = a.gcache + (size_t)(blockIdx.x * 4 + wv) * a.n_cached * ZBUF_BYTES;
        auto restore = [&](int grp) __attribute__((always_inline)) {
            const uint32_t lane16 = lane_id() * 16;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_addr(zwave));
#pragma unroll
            for (int i = 0; i < ZBUF_BYTES / 4096; ++i)
                glds_stage_l2(cache_wave + (size_t)grp * ZBUF_BYTES + i * 4096, lane16, dst + i * 4096);
        };
        auto gather = [&](int grp, bool keep) __attribute__((always_inline)) {
            const int lane = lane_id(), c = lane & 15, g = lane >> 4;
            const float4 uv4 = uvw[lane];
            const float pu[2] = {uv4.x, uv4.z}, pv[2] = {uv4.y, uv4.w};
            int ch0 = 0;
            for (int lvl = 0; lvl < a.vw.n_levels; ++lvl) {
                const int C = a.vw.lat_c[lvl], W = a.vw.lat_w[lvl], H = a.vw.lat_h[lvl];
                const int lo = ch0 > grp * 256 ? ch0 : grp * 256;
                const int hi = (ch0 + C) < (grp + 1) * 256 ? (ch0 + C) : (grp + 1) * 256;
                if (lo < hi) {
#pragma unroll
                    for (int cg = 0; cg < 2; ++cg) {
                        const Taps tp = bilinear_taps(pu[cg] * uv_sx(a.vw, lvl), pv[cg] * uv_sy(a.vw, lvl), W, H);
                        const char* lb = a.lat[lvl] + (size_t)view_of(c, cg, v) * H * W * C * 2;
                        // 4 k-steps (16 tap loads) in flight per iteration: the loop is latency-bound on L2 otherwise
                        for (int cbase = lo; cbase < hi; cbase += 128) {       // (cbase: wave-uniform; the lane's channels start at + 8 g)
                            const int chb = cbase + 8 * g;
                            uint4 q[4][4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                // level widths are multiples of 32: k-step u of this pass exists for the whole wave or not at all — a
                                // 64-channel level (multi-scale) issues 8 of the 16 loads instead of re-reading k-step 0 twice more
                                if (cbase + 32 * u < hi) {
#pragma unroll
                                    for (int i = 0; i < 4; ++i)
                                        q[u][i] = *(const uint4*)(lb + ((size_t)tp.off[i] * C + (chb + 32 * u - ch0)) * 2);
                                }
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int ch = chb + 32 * u;
                                if (cbase + 32 * u < hi) {
                                    float acc8[8];
#pragma unroll
                                    for (int j = 0; j < 8; ++j) acc8[j] = 0.f;
#pragma unroll
                                    for (int i = 0; i < 4; ++i) {
                                        const float w = tp.w[i];
                                        acc8[0] += NM::lo(q[u][i].x) * w; acc8[1] += NM::hi(q[u][i].x) * w;
                                        acc8[2] += NM::lo(q[u][i].y) * w; acc8[3] += NM::hi(q[u][i].y) * w;
                                        acc8[4] += NM::lo(q[u][i].z) * w; acc8[5] += NM::hi(q[u][i].z) * w;
                                        acc8[6] += NM::lo(q[u][i].w) * w; acc8[7] += NM::hi(q[u][i].w) * w;
                                    }
                                    uint4 o;
                                    o.x = NM::pack(acc8[0], acc8[1]); o.y = NM::pack(acc8[2], acc8[3]);
                                    o.z = NM::pack(acc8[4], acc8[5]); o.w = NM::pack(acc8[6], acc8[7]);
                                    const int ks = (ch - grp * 256) >> 5;
                                    *(uint4*)(zwave + (ks * 2 + cg) * 1024 + lane * 16) = o;
                                    if (keep) *(uint4*)(cache_wave + (size_t)grp * ZBUF_BYTES + (ks * 2 + cg) * 1024 + lane * 16) = o;
                                }
                            }
                        }
                    }
                }
                ch0 += C;
            }
        };

        // ---- projected mode: the B operand of lin_z is the point's tap-weight vector over the T texels,
        //      image [k-step][column group][lane group][col][8]: texel t sits at k-step t/32, lane group (t/8)&3, element t&7.
        //      Lane (g, c) writes tap g of its two points.
        auto tap_image = [&]() __attribute__((always_inline)) {
            const int lane = lane_id(), c = lane & 15, g = lane >> 4;
            const float4 uv4 = uvw[lane];
            const float pu[2] = {uv4.x, uv4.z}, pv[2] = {uv4.y, uv4.w};
            const int ll = a.vw.n_levels - 1;
            const uint4 z4 = {0u, 0u, 0u, 0u};
            for (int s = 0; s < 2 * p_steps; ++s) *(uint4*)(zwave + s * 1024 + lane * 16) = z4;
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                const Taps tp = bilinear_taps(pu[cg] * uv_sx(a.vw, ll), pv[cg] * uv_sy(a.vw, ll), a.vw.lat_w[ll], a.vw.lat_h[ll]);
                // one tap per instruction: horizontally adjacent texels are the two halves of ONE dword, and two lanes
                // storing 16 bits each into the same dword in the same instruction lose one of the stores
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = tp.off[i];
                    const float w = tp.w[i];
                    if (g == i && w != 0.f)
                        *(uint16_t*)(zwave + ((t >> 5) * 2 + cg) * 1024 + ((((t >> 3) & 3) * 16 + c) * 16) + (t & 7) * 2) = NM::cvt(w);
                }
            }
        };
        // x_stages with the tiles as OUTPUTS only: LIN_IN starts the residual stream (init mode: its first k-step writes x)
        auto lin_in_stages = [&]() __attribute__((always_inline)) {
            const int cfg2 = a.S_in | (1 << 9);
            PNR_LANE_OPERANDS;
            if (DT == PNR_BF16)
                asm volatile(PNR_XSTAGES_ASM_BF16 : PNR_ASM_STATE_OPERANDS_OUT
                             : "s"(asm_cfg), "s"(stream), "s"(ring_lds), "v"(ring_lane), "v"(gl_off), "v"(zaddr), "v"(bias_dword), "s"(cfg2)
                             : PNR_RESBLOCK_CLOBBERS);
            else
                asm volatile(PNR_XSTAGES_ASM_F16 : PNR_ASM_STATE_OPERANDS_OUT
                             : "s"(asm_cfg), "s"(stream), "s"(ring_lds), "v"(ring_lane), "v"(gl_off), "v"(zaddr), "v"(bias_dword), "s"(cfg2)
                             : PNR_RESBLOCK_CLOBBERS);
        };
        // ---- resblock b: [last lin_z part] + bias k-step + x += fc_1(relu(fc_0(relu(x))))  (resnetfc.py:53-62, 203-234)
        // n_pre consecutive blocks from b on with the lin_z prefix, then n_plain without, in ONE statement (its block loop)
        // prefetch: the statement also fetches the NEXT block's first gathered image back from the workspace into the wave's
        // B-image buffer (16 LDS-DMA pieces in its two head stages: the buffer is free from there on), so that block's restore()
        // disappears — see gen() in tools/gen_resblock_asm.py
        auto resblocks = [&](int b, int n_pre, int n_plain, bool prefetch) __attribute__((always_inline)) {
            if (n_pre + n_plain == 0) return;
            PNR_LANE_OPERANDS;
            const uint32_t bias_addr = lds_addr(btab) + b * (HID * 4) + (ln_ >> 4) * 16;
            // k-steps of the prefix (the last lin_z part) that run as x-stages; with the bias folded into the projected columns
            // the last of them takes the place of the bias k-step (bit 24)
            const int cfg2z = (a.proj ? p_steps - 1 : 8) | (n_pre << 16) | (n_plain << 20) | (a.proj ? 1 << 24 : 0) | (prefetch ? 1 << 25 : 0);
            const char* img_g = cache_wave;                        // image of group 0 (wave-uniform)
            const uint32_t lane16 = (uint32_t)ln_ * 16;
            const uint32_t img_l = __builtin_amdgcn_readfirstlane(lds_addr(zwave));
            if (DT == PNR_BF16)
                asm volatile(PNR_RESBLOCK_ASM_BF16 : PNR_ASM_STATE_OPERANDS
                             : "s"(asm_cfg), "s"(stream), "s"(ring_lds), "v"(ring_lane), "v"(gl_off), "v"(bias_addr), "v"(bias_dword),
                               "v"(zaddr), "s"(cfg2z), "s"(img_g), "v"(lane16), "s"(img_l)
                             : PNR_RESBLOCK_CLOBBERS);
            else
                asm volatile(PNR_RESBLOCK_ASM_F16 : PNR_ASM_STATE_OPERANDS
                             : "s"(asm_cfg), "s"(stream), "s"(ring_lds), "v"(ring_lane), "v"(gl_off), "v"(bias_addr), "v"(bias_dword),
                               "v"(zaddr), "s"(cfg2z), "s"(img_g), "v"(lane16), "s"(img_l)
                             : PNR_RESBLOCK_CLOBBERS);
        };

        // ---- one source view: features, LIN_IN, the blocks before the view reduction
        auto view_pass = [&]() __attribute__((always_inline)) {
            {
                // ---- per-view geometry (models.py.backup2:166-221) and positional features -> wave-private LDS image
                //      [k-step][column group][lane][8] (16-bit) in the slot layout of make_layout: every lane builds the 8 S_in
                //      entries of its k-quarter for both of its points (uniform code; the quarter picks frequency / phase /
                //      coordinate by arithmetic) and stores them with one ds_write_b128 per k-step.
                const int lane = lane_id(), c = lane & 15, g = lane >> 4;
                float pu[2], pv[2];
#pragma unroll
                for (int cg = 0; cg < 2; ++cg) {
                    // one object: the view index is wave-uniform, the camera comes through the scalar cache
                    const Cam cam = a.vw.n_objs == 1 ? load_cam(a.vw, __builtin_amdgcn_readfirstlane(v)) : load_cam(a.vw, view_of(c, cg, v));
                    float p[3], d[3], xr[3], dr[3];
                    const int pl = 16 * cg + c;
                    // several views: the point and its direction do not depend on the view — view pass 0 leaves them in the
                    // wave's tile-input buffer (over the ray records it has consumed) and the later passes read them back
                    // instead of repeating ray / sample arithmetic (Philox, divisions)
                    if (MULTIVIEW && v > 0 && (flag(a.job.from_cam) || flagp(a.src.rays))) {
                        const float4 cp = *(const float4*)(pts + pl * 16), cd = *(const float4*)(pts + 512 + pl * 16);
                        p[0] = cp.x; p[1] = cp.y; p[2] = cp.z;
                        d[0] = cd.x; d[1] = cd.y; d[2] = cd.z;
                    } else if ((flag(a.job.from_cam) || flagp(a.src.rays))) {
                        float o3[3], near, far, zz;
                        int li = tile * TILE_PTS + wv * 32 + pl;
                        const bool in_range = li < n_loc;
                        li = in_range ? li : n_loc - 1;
                        // divisors opaque to the optimiser: their reciprocals would otherwise be formed once at the kernel's
                        // entry and kept across the MFMA blocks in scratch (see lane_id above)
                        int Kq = a.job.K;
                        asm volatile("" : "+s"(Kq));
                        const int lr = flag(a.job.on) ? (int)((uint32_t)li / (uint32_t)Kq) : 0;     // ray within the workgroup
                        if (flag(a.job.from_cam)) {
                            RayCam cam_q = a.job.cam;
                            asm volatile("" : "+s"(cam_q.W), "+s"(cam_q.fx), "+s"(cam_q.fy));
                            pinhole_ray(cam_q, a.job.pix0 + (int)ray_begin + lr, d);
                            o3[0] = cam_q.o[0]; o3[1] = cam_q.o[1]; o3[2] = cam_q.o[2];
                            near = cam_q.zn; far = cam_q.zf;
                        } else {
                            const float4 r0 = *(const float4*)(pts + pl * 16), r1 = *(const float4*)(pts + 512 + pl * 16);
                            o3[0] = r0.x; o3[1] = r0.y; o3[2] = r0.z;
                            d[0] = r0.w; d[1] = r1.x; d[2] = r1.y; near = r1.z; far = r1.w;
                        }
                        if (flag(a.job.gen_z)) {
                            // sample_coarse (nerf.py:98-118), the arithmetic of k_sample_coarse; view pass 0 leaves the positions
                            // in z_out for the compositing, the fine resampling and the caller
                            const int k = li - lr * Kq;
                            const int64_t gp = p_begin + li;
                            const float u = flagp(a.job.noise_c) ? a.job.noise_c[gp]
                                                          : rng_uniform(a.job.seed, global_ray(a.job.key, ray_begin + lr), DRAW_COARSE, k);
                            const float t = fmaf(u, 1.0f / (float)Kq, linspace_k(k, Kq));
                            zz = z_from_t(t, near, far, a.job.lindisp);
                            if (v == 0 && g == 0 && in_range && (!flag(a.job.cmp_lds) || flag(a.job.z_needed))) a.job.z_out[gp] = zz;
                            // on-chip compositing: the tile epilogue picks the position up from the tile-input buffer's z slot
                            // (unused when the positions are generated here; with PointSrc.z the prefetch has put it there)
                            if (v == 0 && g == 0 && flag(a.job.cmp_lds)) *(float*)(const_cast<char*>(pts) + 1024 + pl * 4) = zz;
                        } else {
                            zz = *(const float*)(pts + 1024 + pl * 4);
                        }
                        p[0] = fmaf(zz, d[0], o3[0]); p[1] = fmaf(zz, d[1], o3[1]); p[2] = fmaf(zz, d[2], o3[2]);
                        if (MULTIVIEW && g == 0) {        // every lane of the wave has read its ray record above (program order)
                            *(float4*)(pts + pl * 16) = make_float4(p[0], p[1], p[2], zz);
                            *(float4*)(pts + 512 + pl * 16) = make_float4(d[0], d[1], d[2], 0.f);
                        }
                    } else {
                        const float* q3 = (const float*)(pts + pl * 4);
                        p[0] = q3[0]; p[1] = q3[64]; p[2] = q3[128];
                        d[0] = q3[32]; d[1] = q3[96]; d[2] = q3[160];
                    }
                    rot3(cam.R, p, xr);
                    rot3(cam.R, d, dr);
                    project(cam, xr, pu[cg], pv[cg]);
                    auto image = [&](auto dc) {
                        constexpr int D = decltype(dc)::value;
                        constexpr int SL = 8 * ((3 * D + 4 + 7) / 8);
                        float val[SL];
#pragma unroll
                        for (int jj = 0; jj < 3 * D; ++jj) {
                            // sine n = 3 D g + jj in code.py's order n = 2 D q + D phase + i.  3 D g is a multiple of D, so the
                            // input i = jj mod D is the same in every lane group (no per-lane select); with m = 3 g + jj / D:
                            // phase = m & 1, frequency q = m >> 1.
                            const int i = jj % D, m = 3 * g + jj / D;
                            const float vi = i < 3 ? xr[i] : dr[i - 3];
                            const float fq = __builtin_amdgcn_ldexpf(a.freq_factor, m >> 1);
                            val[jj] = __sinf(fmaf(vi, fq, (m & 1) ? 1.57079637f : 0.0f));
                        }
                        // others: raw input 2 g + o (x_rot 0..2, rotated view dir 3..5) for g < 3; the folded lin_in bias for g = 3
                        const float o0 = g == 0 ? xr[0] : g == 1 ? xr[2] : g == 2 ? dr[1] : 1.0f;
                        const float o1 = g == 0 ? xr[1] : g == 1 ? dr[0] : g == 2 ? dr[2] : 1.0f;
                        val[3 * D + 0] = o0; val[3 * D + 1] = o1;
                        val[3 * D + 2] = g == 3 ? 0.f : o0 - NM::back(NM::cvt(o0));
                        val[3 * D + 3] = g == 3 ? 0.f : o1 - NM::back(NM::cvt(o1));
#pragma unroll
                        for (int i = 3 * D + 4; i < SL; ++i) val[i] = 0.f;
#pragma unroll
                        for (int s = 0; s < SL / 8; ++s) {
                            uint4 o;
                            o.x = NM::pack(val[8 * s + 0], val[8 * s + 1]); o.y = NM::pack(val[8 * s + 2], val[8 * s + 3]);
                            o.z = NM::pack(val[8 * s + 4], val[8 * s + 5]); o.w = NM::pack(val[8 * s + 6], val[8 * s + 7]);
                            *(uint4*)(zwave + (s * 2 + cg) * 1024 + lane * 16) = o;
                        }
                    };
                    if (a.use_code_viewdirs) image(std::integral_constant<int, 6>{});
                    else image(std::integral_constant<int, 3>{});
                }
                uvw[lane] = make_float4(pu[0], pv[0], pu[1], pv[1]);      // for the gather / tap image of this view's blocks
                STAMP_ACC(1, st_t);
                lin_in_stages();
                STAMP_ACC(8, st_t);
                if (flag(a.job.on) && v == 0 && tile > 0) {
                    // the rays the previous tile finished: every wave has been through the vmcnt(0) and the barriers of the
                    // LIN_IN statement since it stored that tile's outputs, so they are in the L2 for any wave to read
                    int upto = tile * TILE_PTS;
                    int Kc_q = a.job.K;
                    asm volatile("" : "+s"(Kc_q));             // divisor formed here (see lane_id)
                    upto = __builtin_amdgcn_readfirstlane((int)((uint32_t)(upto < n_loc ? upto : n_loc) / (uint32_t)Kc_q));
                    composite_rays(rays_done, upto);
                    rays_done = upto;
                    STAMP_ACC(9, st_t);
                }
                if (a.proj && n_gather == 0) tap_image();          // fully projected: the image serves all blocks of this view
                else if (!a.proj && n_gather == 1) gather(0, false);
                STAMP_ACC(2, st_t);
            }
            // One lin_z part per block (one_part): the B image written above serves every block of the view, so they run in ONE
            // statement (single view: the blocks behind the reduction follow in it too).  The same loop and the same
            // statement serve both cases — a second call site would make hipcc merge the tiles from two paths (through scratch).
            const int b_step = one_part ? a.nb1 : 1;
            const int plain_here = (one_part && !MULTIVIEW) ? a.n_blocks - a.nb1 : 0;
            // one gathered group in front of the block's last lin_z part (multi-scale): block b's statement prefetches block
            // b + 1's image of that group while its chunks run
#ifdef PNR_NO_IMAGE_PREFETCH      // A/B builds only (tools/dev, with the generator's `noprefetch`)
            const bool pf = false;
#else
            const bool pf = n_groups == 1 && !one_part;
#endif
            for (int b = 0; b < a.nb1; b += b_step) {
                // ---- x += lin_z[b](z): all but the block's last part as separate x-stage calls
                for (int grp = 0; grp < n_groups; ++grp) {
                    if (b == 0) gather(grp, true);
                    else if (!pf) restore(grp);
                    STAMP_ACC(10, st_t);
                    x_stages(8);
                    STAMP_ACC(11, st_t);
                }
                if (a.proj && n_gather > 0) tap_image();           // partial projection: the buffer was just used by the gather
                else if (!a.proj && n_gather > 1) {
                    if (b == 0) gather(n_gather - 1, true);
                    else restore(n_gather - 1);
                }
                STAMP_ACC(3, st_t);
                resblocks(b, b_step, plain_here, pf && b + 1 < a.nb1);
                STAMP_ACC(6, st_t);
            }
        };
        // Straight-line tile lifetime for the register allocator: every view pass DEFINES the tiles (LIN_IN writes them), a
        // park only reads them (dead afterwards), the reduce updates the last view's in place — no tile value is carried
        // around a loop or merged from two paths, so hipcc has no reason to move tiles through VGPRs or scratch.
#define PNR_X_TILES                                                                                                     \
    "+{a[0:15]}"(x[0]), "+{a[16:31]}"(x[1]), "+{a[32:47]}"(x[2]), "+{a[48:63]}"(x[3]), "+{a[64:79]}"(x[4]),               \
    "+{a[80:95]}"(x[5]), "+{a[96:111]}"(x[6]), "+{a[112:127]}"(x[7]), "+{a[128:143]}"(x[8]), "+{a[144:159]}"(x[9]),      \
    "+{a[160:175]}"(x[10]), "+{a[176:191]}"(x[11]), "+{a[192:207]}"(x[12]), "+{a[208:223]}"(x[13]),                    \
    "+{a[224:239]}"(x[14]), "+{a[240:255]}"(x[15])
#define PNR_X_TILES_IN                                                                                                  \
    "{a[0:15]}"(x[0]), "{a[16:31]}"(x[1]), "{a[32:47]}"(x[2]), "{a[48:63]}"(x[3]), "{a[64:79]}"(x[4]),                    \
    "{a[80:95]}"(x[5]), "{a[96:111]}"(x[6]), "{a[112:127]}"(x[7]), "{a[128:143]}"(x[8]), "{a[144:159]}"(x[9]),           \
    "{a[160:175]}"(x[10]), "{a[176:191]}"(x[11]), "{a[192:207]}"(x[12]), "{a[208:223]}"(x[13]),                         \
    "{a[224:239]}"(x[14]), "{a[240:255]}"(x[15])
        if (MULTIVIEW) {
            // multi-view (util.combine_interleaved, util.py:466-476): the per-view residual streams of views 0 .. NS-2 are parked in
            // the workspace and reduced (mean / max) into the last view's, both as asm blocks on the pinned registers (NS >= 2 here)
            const float4* slot0 = a.spill + ((size_t)(blockIdx.x * 4 + wv) * (a.NS - 1)) * 4096;
            for (v = 0; v < a.NS - 1; ++v) {
                view_pass();
                const float4* slot = slot0 + (size_t)v * 4096;
                const uint32_t lane16 = lane_id() * 16;
                if (DT == PNR_BF16) asm volatile(PNR_VIEWSPILL_ASM_BF16 : : PNR_X_TILES_IN, "s"(slot), "v"(lane16), "s"(a.park32) : PNR_RESBLOCK_CLOBBERS);
                else asm volatile(PNR_VIEWSPILL_ASM_F16 : : PNR_X_TILES_IN, "s"(slot), "v"(lane16), "s"(a.park32) : PNR_RESBLOCK_CLOBBERS);
                STAMP_ACC(4, st_t);
            }
            view_pass();
            const int nm1 = a.NS - 1;
            int ns_q = a.NS;
            asm volatile("" : "+s"(ns_q));                  // formed here, not kept from the kernel's entry in scratch
            const float inv = 1.0f / (float)ns_q;
            const uint32_t lane16 = lane_id() * 16;
            if (DT == PNR_BF16)
                asm volatile(PNR_VIEWREDUCE_ASM_BF16 : PNR_X_TILES : "s"(slot0), "s"(nm1), "s"(a.combine_max), "v"(lane16), "v"(inv), "s"(a.park32)
                             : PNR_RESBLOCK_CLOBBERS);
            else
                asm volatile(PNR_VIEWREDUCE_ASM_F16 : PNR_X_TILES : "s"(slot0), "s"(nm1), "s"(a.combine_max), "v"(lane16), "v"(inv), "s"(a.park32)
                             : PNR_RESBLOCK_CLOBBERS);
            STAMP_ACC(5, st_t);
        } else {
            view_pass();
        }
#undef PNR_X_TILES
#undef PNR_X_TILES_IN
        // ---- the blocks after the view reduction
        // (a loop, not an `if`: see view_pass; with no lin_z block at all the view pass has run none of them)
        for (int i = (MULTIVIEW || !one_part || a.nb1 == 0) ? 0 : 1; i < 1; ++i) {
            resblocks(a.nb1, 0, a.n_blocks - a.nb1, false);
            STAMP_ACC(6, st_t);
        }

        // ---- last fc_1 bias, lin_out(relu(x)), sigmoid / relu (resnetfc.py:235, models.py.backup2:274-281)
        PNR_LANE_OPERANDS;
        if (DT == PNR_BF16)
            asm volatile(PNR_LINOUT_ASM_BF16 : PNR_ASM_STATE_OPERANDS
                         : "s"(asm_cfg), "s"(stream), "s"(ring_lds), "v"(ring_lane), "v"(gl_off), "v"(zaddr), "v"(bias_dword) : PNR_RESBLOCK_CLOBBERS);
        else
            asm volatile(PNR_LINOUT_ASM_F16 : PNR_ASM_STATE_OPERANDS
                         : "s"(asm_cfg), "s"(stream), "s"(ring_lds), "v"(ring_lane), "v"(gl_off), "v"(zaddr), "v"(bias_dword) : PNR_RESBLOCK_CLOBBERS);
        {
            // the block left rows 0..3 of the output in the wave's LDS buffer: [column group][lane (g = 0: lanes 0..15)] x float4.
            // Lanes 0..31 store the wave's 32 consecutive points.
            const int lane = lane_id();
            const int li = tile * TILE_PTS + wv * 32 + lane;
            const int64_t gi = p_begin + li;
            if (lane < 32 && li < n_loc) {
                const float4 o = *(const float4*)(zwave + (lane >> 4) * 1024 + (lane & 15) * 16);
                const float* bo = btab + a.n_blocks * HID;
                float4 res;
                res.x = 1.0f / (1.0f + __expf(-(o.x + bo[0])));
                res.y = 1.0f / (1.0f + __expf(-(o.y + bo[1])));
                res.z = 1.0f / (1.0f + __expf(-(o.z + bo[2])));
                res.w = fmaxf(o.w + bo[3], 0.f);
                if (flag(a.job.cmp_lds)) {
                    // fused render launch, K <= CMP_MAX_K: the point's output stays on the chip — (rgb, sigma) and z into the
                    // compositing ring (read after the next tile's LIN_IN statement, whose entry waits and stage barriers order
                    // these writes in front of the reads; the last tile's behind the loop's own wait + barrier)
                    const int slot = li & (CMP_RING - 1);
                    const_cast<float4*>(cmp_c)[slot] = res;
                    const_cast<float*>(cmp_z)[slot] = *(const float*)(pts + 1024 + lane * 4);
                } else {
                    ((float4*)a.out)[gi] = res;
                }
            }
        }
        STAMP_ACC(7, st_t);
#ifdef PNR_STAMPS
        st_acc[0] += st_t - st_tile;
#endif
    }
    if (a.job.on) {
        // the rays the last tile finished
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        composite_rays(rays_done, n_loc / a.job.K);
    }
#ifdef PNR_STAMPS
    if (lane_id() == 0) {
        for (int i = 0; i < 12; ++i) atomicAdd(&g_stamps[i], st_acc[i]);
        atomicAdd(&g_stamps[12], __builtin_amdgcn_s_memtime() - st_k0);          // shader cycles / 100 MHz ticks of the whole
        atomicAdd(&g_stamps[13], __builtin_amdgcn_s_memrealtime() - st_r0);      // kernel: their ratio is the in-kernel clock
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the run-ahead LDS-DMA before the LDS is released
}

// ---------------------------------------------------------------------------- host side
static int num_cus() {
    // immutable per-device property, cached (hipGetDeviceProperties is slow); not library state in the sense of pnr.h
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}
static constexpr int MAX_GRID = 512;

// 256-channel lin_z groups whose gathered image is kept in the workspace between a view's blocks
static int cached_groups(const pnr_mlp* mlp, const pnr_views* vw) {
    if (mlp->packed_texels) return (mlp->d_latent - vw->lat_c[vw->n_levels - 1]) / 256;
    const int groups = mlp->d_latent / 256;
    return groups > 1 ? groups : 0;
}
static uint64_t spill_bytes(const pnr_views* vw) {
    return vw->n_views > 1 ? (uint64_t)MAX_GRID * 4 * (vw->n_views - 1) * 4096 * sizeof(float4) : 0;
}
uint64_t point_mfma_workspace_bytes(const pnr_mlp* mlp, const pnr_views* vw) {
    return 256 + spill_bytes(vw) + (uint64_t)MAX_GRID * 4 * cached_groups(mlp, vw) * ZBUF_BYTES;
}

int32_t point_mfma(const pnr_params* prm, const pnr_mlp* mlp, const pnr_views* vw, PointSrc src, int64_t n_points,
                   int64_t pts_per_obj, float* out, void* workspace, uint64_t ws_bytes, hipStream_t s, const RayJob* job) {
    Layout y;
    const int proj = mlp->packed_texels;
    const int last = vw->n_levels - 1;
    const int gathered = proj ? (mlp->d_latent - vw->lat_c[last]) / 256 : 0;
    if (proj && (vw->lat_c[last] != 256 || (mlp->d_latent - 256) % 256 != 0)) return PNR_E_PACKED;
    if (!make_layout(*mlp, y, proj, vw->n_views, gathered)) return proj ? PNR_E_PACKED : PNR_E_UNSUPPORTED;
    // projected: one stream per object (each holds W_z . Lat of that object's views): the blob must have been packed for
    // exactly the objects being rendered
    const int packed_objs = mlp->packed_objs > 0 ? mlp->packed_objs : 1;
    if (!mlp->packed || mlp->packed_dtype != prm->precision ||
        mlp->packed_bytes < y.btab_bytes + (uint64_t)(proj ? packed_objs : 1) * y.stream_bytes) return PNR_E_PACKED;
    if (((uintptr_t)mlp->packed & 15) != 0) return PNR_E_ALIGN;
    if (proj) {      // the stream was packed for these views' latent maps: they must be the maps being rendered
        if (vw->n_objs != packed_objs || vw->lat_h[last] * vw->lat_w[last] != proj) return PNR_E_PACKED;
    }
    if (!proj || gathered > 0) {
        if (vw->packed_dtype != prm->precision) return PNR_E_PACKED;
        for (int i = 0; i < (proj ? last : vw->n_levels); ++i) {
            if (!vw->latent_packed[i]) return PNR_E_PACKED;
            if (((uintptr_t)vw->latent_packed[i] & 15) != 0) return PNR_E_ALIGN;
            if (vw->lat_c[i] % 32 != 0) return PNR_E_UNSUPPORTED;   // a k-step of the gather is 32 channels of one level
        }
    }
    if (ws_bytes < point_mfma_workspace_bytes(mlp, vw)) return PNR_E_WORKSPACE;
    if (vw->n_views > 1 && y.nb1 == 0) return PNR_E_UNSUPPORTED;     // reduction before the first block
    if (prm->use_code_viewdirs ? (mlp->d_in != 6 + 12 * prm->num_freqs) : (mlp->d_in != 6 + 6 * prm->num_freqs)) return PNR_E_SHAPE;
    if (prm->num_freqs != 6) return PNR_E_UNSUPPORTED;     // the LIN_IN slot layout is built for 6 frequencies (every shipped config)

    MfmaArgs a;
    a.vw = *vw; a.src = src; a.n_points = n_points; a.pts_per_obj = pts_per_obj;
    a.btab = (const float*)mlp->packed;
    a.stream = (const char*)mlp->packed + y.btab_bytes;
    for (int i = 0; i < PNR_MAX_LEVELS; ++i) a.lat[i] = (const char*)vw->latent_packed[i];
    a.out = out;
    a.spill = (float4*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    a.gcache = (char*)a.spill + spill_bytes(vw);
    a.n_cached = cached_groups(mlp, vw);
    a.n_tiles = (int)((n_points + TILE_PTS - 1) / TILE_PTS);
    a.NS = vw->n_views; a.combine_max = mlp->combine_type == PNR_COMBINE_MAX;
    a.park32 = prm->park_fp32 ? 1 : 0;
    a.S_in = y.S_in; a.SZ = y.SZ; a.n_blocks = y.n_blocks; a.nb1 = y.nb1; a.P1 = y.P1; a.P2 = y.P2;
    a.btab_floats = y.btab_floats; a.d_in = mlp->d_in; a.proj = proj; a.Gg = y.Gg;
    a.ldP1 = proj ? y.PV * y.P1 : y.P1; a.ldNS = proj ? 1 : vw->n_views;
    a.use_code_viewdirs = prm->use_code_viewdirs; a.num_freqs = prm->num_freqs; a.freq_factor = prm->freq_factor;
    int grid = num_cus();
    if (grid > MAX_GRID) grid = MAX_GRID;
    a.job = RayJob{};
    a.tiles_per_wg = 0;
    a.wgs_per_obj = 0;
    a.obj_stream_stride = 0;
    if (proj && vw->n_objs > 1) {
        // several objects with projected streams: workgroups per object, so that a workgroup streams ONE object's weights
        const int n_objs = vw->n_objs;
        int wpo = grid / n_objs;
        if (wpo < 1) wpo = 1;
        if ((int64_t)wpo * n_objs > MAX_GRID) return PNR_E_UNSUPPORTED;
        a.obj_stream_stride = (int64_t)y.stream_bytes;
        if (job && job->on) {
            a.job = *job;
            if (a.job.K < 1 || a.job.n_rays * a.job.K != n_points || !a.job.rgb_out || !a.job.depth_out) return PNR_E_SHAPE;
            if (a.job.gen_z ? !a.job.z_out : !src.z) return PNR_E_NULL;
            if (!a.job.from_cam && !src.rays) return PNR_E_NULL;
            const int64_t rays_per_obj = pts_per_obj / a.job.K;
            int64_t rpw = (rays_per_obj + wpo - 1) / wpo;
            const int64_t min_rpw = (TILE_PTS + a.job.K - 1) / a.job.K;
            if (rpw < min_rpw) rpw = min_rpw;
            if (rpw * a.job.K >= 0x7fffffffLL) return PNR_E_SHAPE;
            a.job.rays_per_wg = (int)rpw;
            wpo = (int)((rays_per_obj + rpw - 1) / rpw);
        } else {
            const int64_t tiles_obj = (pts_per_obj + TILE_PTS - 1) / TILE_PTS;
            int64_t tpw = (tiles_obj + wpo - 1) / wpo;
            if (tpw * TILE_PTS >= 0x7fffffffLL) return PNR_E_SHAPE;
            a.tiles_per_wg = (int)tpw;
            wpo = (int)((tiles_obj + tpw - 1) / tpw);
        }
        a.wgs_per_obj = wpo;
        grid = wpo * n_objs;
    } else if (job && job->on) {
        // whole rays per workgroup, at least about a tile's worth of points each
        a.job = *job;
        if (a.job.K < 1 || a.job.n_rays * a.job.K != n_points || !a.job.rgb_out || !a.job.depth_out) return PNR_E_SHAPE;
        if (a.job.gen_z ? !a.job.z_out : !src.z) return PNR_E_NULL;
        if (!a.job.from_cam && !src.rays) return PNR_E_NULL;
        int64_t rpw = (a.job.n_rays + grid - 1) / grid;
        const int64_t min_rpw = (TILE_PTS + a.job.K - 1) / a.job.K;
        if (rpw < min_rpw) rpw = min_rpw;
        if (rpw * a.job.K >= 0x7fffffffLL) return PNR_E_SHAPE;
        a.job.rays_per_wg = (int)rpw;
        grid = (int)((a.job.n_rays + rpw - 1) / rpw);
    } else {
        int64_t tpw = ((int64_t)a.n_tiles + grid - 1) / grid;
        if (tpw * TILE_PTS >= 0x7fffffffLL) return PNR_E_SHAPE;
        a.tiles_per_wg = (int)tpw;
        grid = (int)((a.n_tiles + tpw - 1) / tpw);
    }
    size_t lds = LDS_BTAB + (size_t)y.btab_floats * 4;
    // compositing from the LDS ring: fused render launches whose rays are at most a tile long, where the ring fits behind the
    // bias table; everything else composites from the (rgb, sigma) / z the launch leaves in global memory
    a.job.cmp_lds = (a.job.on && a.job.K <= CMP_MAX_K && lds + CMP_BYTES <= (size_t)LDS_LIMIT) ? 1 : 0;
#ifdef PNR_NO_LDS_COMPOSITE      // A/B builds only (tools/dev): every launch takes the memory route
    a.job.cmp_lds = 0;
#endif
    if (a.job.cmp_lds) lds += CMP_BYTES;
    const void* fn;
    const bool mv = a.NS > 1;
    if (prm->precision == PNR_BF16) fn = mv ? (const void*)k_point_mfma<PNR_BF16, true> : (const void*)k_point_mfma<PNR_BF16, false>;
    else fn = mv ? (const void*)k_point_mfma<PNR_F16, true> : (const void*)k_point_mfma<PNR_F16, false>;
    PNR_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* kargs[] = {(void*)&a};
    PNR_HIP_CHECK(hipLaunchKernel(fn, dim3(grid), dim3(256), kargs, lds, s));
    PNR_LAUNCH_CHECK();
    return PNR_OK;
}

}  // namespace pnr

using namespace pnr;

#ifdef PNR_STAMPS
extern "C" int32_t pnr_debug_stamps(unsigned long long* out16, int reset) {
    PNR_HIP_CHECK(hipDeviceSynchronize());
    PNR_HIP_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(pnr::g_stamps), 16 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[16] = {0};
        PNR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(pnr::g_stamps), z, sizeof(z)));
    }
    return 0;
}
#endif

extern "C" uint64_t pnr_packed_mlp_bytes(const pnr_mlp* mlp) {
    Layout y;
    if (!mlp || !make_layout(*mlp, y)) return 0;
    return y.total_bytes;
}

// texel count of the last level if (mlp, views) qualifies for the projected stream, else 0; *gathered = 256-channel
// groups of the levels before it
static int projectable(const pnr_mlp* mlp, const pnr_views* vw, int* gathered) {
    if (!mlp || !vw || vw->n_objs < 1 || vw->n_objs > 16 || vw->n_views < 1 || vw->n_views > 8 || vw->n_levels < 1 ||
        vw->n_levels > PNR_MAX_LEVELS)
        return 0;
    const int last = vw->n_levels - 1;
    if (!vw->latent[last] || vw->lat_c[last] != 256 || mlp->combine_layer < 1) return 0;
    int L = 0;
    for (int i = 0; i < vw->n_levels; ++i) L += vw->lat_c[i];
    if (L != mlp->d_latent || (L - 256) % 256 != 0) return 0;
    *gathered = (L - 256) / 256;
    int T = vw->lat_h[last] * vw->lat_w[last];
    // fully projected maps pay ceil(T/32) k-steps instead of 8 + the gather: worth it up to 256 texels
    return (T >= 4 && T <= 256) ? T : 0;
}

extern "C" uint64_t pnr_packed_mlp_projected_bytes(const pnr_mlp* mlp, const pnr_views* views) {
    Layout y;
    int gathered = 0;
    int T = projectable(mlp, views, &gathered);
    if (!T || !make_layout(*mlp, y, T, views->n_views, gathered)) return 0;
    return y.btab_bytes + (uint64_t)views->n_objs * (y.stream_bytes + y.proj_bytes);       // one stream (+ scratch) per object
}

extern "C" int32_t pnr_pack_mlp_projected(const pnr_mlp* mlp, const pnr_views* views, int32_t dtype, void* out,
                                          uint64_t out_bytes, void* stream) {
    if (!mlp || !views || !out) return PNR_E_NULL;
    int gathered = 0;
    int T = projectable(mlp, views, &gathered);
    Layout y;
    if (!T || !make_layout(*mlp, y, T, views->n_views, gathered)) return PNR_E_UNSUPPORTED;
    if (dtype != PNR_BF16 && dtype != PNR_F16) return PNR_E_UNSUPPORTED;
    const int n_objs = views->n_objs;
    if (out_bytes < y.btab_bytes + (uint64_t)n_objs * (y.stream_bytes + y.proj_bytes)) return PNR_E_WORKSPACE;
    if (((uintptr_t)out & 15) != 0) return PNR_E_ALIGN;
    if (!mlp->lin_in_w || !mlp->lin_in_b || !mlp->lin_out_w || !mlp->lin_out_b) return PNR_E_NULL;
    for (int b = 0; b < mlp->n_blocks; ++b) {
        if (!mlp->fc0_w[b] || !mlp->fc0_b[b] || !mlp->fc1_w[b] || !mlp->fc1_b[b]) return PNR_E_NULL;
        if (b < y.nb1 && (!mlp->lin_z_w[b] || !mlp->lin_z_b[b])) return PNR_E_NULL;
    }
    // blob: [bias table][stream of object 0][stream of object 1]..[fp32 M scratch of object 0]..
    const int last = views->n_levels - 1;
    const size_t lat_obj = (size_t)views->n_views * views->lat_c[last] * T;          // floats of one object's views in the last level
    int64_t n_out = (int64_t)y.PV * y.nb1 * HID * y.ZK;
    for (int o = 0; o < n_objs; ++o) {
        float* M = (float*)((char*)out + y.btab_bytes + (uint64_t)n_objs * y.stream_bytes + (uint64_t)o * y.proj_bytes);
        hipLaunchKernelGGL(k_project_latent, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *mlp, y,
                           views->latent[last] + (size_t)o * lat_obj, T, M);
        PNR_LAUNCH_CHECK();
        const uint64_t off = (uint64_t)o * y.stream_bytes;
        if (dtype == PNR_BF16) hipLaunchKernelGGL(k_pack_mlp<PNR_BF16>, dim3(2048), dim3(256), 0, (hipStream_t)stream, *mlp, y, (char*)out, (const float*)M, off);
        else hipLaunchKernelGGL(k_pack_mlp<PNR_F16>, dim3(2048), dim3(256), 0, (hipStream_t)stream, *mlp, y, (char*)out, (const float*)M, off);
        PNR_LAUNCH_CHECK();
    }
    return PNR_OK;
}

extern "C" int32_t pnr_pack_mlp(const pnr_mlp* mlp, int32_t dtype, void* out, uint64_t out_bytes, void* stream) {
    if (!mlp || !out) return PNR_E_NULL;
    Layout y;
    if (!make_layout(*mlp, y)) return PNR_E_UNSUPPORTED;
    if (dtype != PNR_BF16 && dtype != PNR_F16) return PNR_E_UNSUPPORTED;
    if (out_bytes < y.total_bytes) return PNR_E_WORKSPACE;
    if (((uintptr_t)out & 15) != 0) return PNR_E_ALIGN;
    if (!mlp->lin_in_w || !mlp->lin_in_b || !mlp->lin_out_w || !mlp->lin_out_b) return PNR_E_NULL;
    for (int b = 0; b < mlp->n_blocks; ++b) {
        if (!mlp->fc0_w[b] || !mlp->fc0_b[b] || !mlp->fc1_w[b] || !mlp->fc1_b[b]) return PNR_E_NULL;
        if (b < y.nb1 && (!mlp->lin_z_w[b] || !mlp->lin_z_b[b])) return PNR_E_NULL;
    }
    if (dtype == PNR_BF16) hipLaunchKernelGGL(k_pack_mlp<PNR_BF16>, dim3(2048), dim3(256), 0, (hipStream_t)stream, *mlp, y, (char*)out, (const float*)nullptr);
    else hipLaunchKernelGGL(k_pack_mlp<PNR_F16>, dim3(2048), dim3(256), 0, (hipStream_t)stream, *mlp, y, (char*)out, (const float*)nullptr);
    PNR_LAUNCH_CHECK();
    return PNR_OK;
}

extern "C" uint64_t pnr_packed_latent_bytes(const pnr_views* views) {
    if (!views || views->n_levels < 1 || views->n_levels > PNR_MAX_LEVELS) return 0;
    return lat_layout(*views).total;
}

extern "C" int32_t pnr_pack_latents(const pnr_views* views, int32_t dtype, void* out, uint64_t out_bytes,
                                    uint64_t* level_offsets, void* stream) {
    if (!views || !out) return PNR_E_NULL;
    if (views->n_levels < 1 || views->n_levels > PNR_MAX_LEVELS) return PNR_E_SHAPE;
    if (dtype != PNR_BF16 && dtype != PNR_F16) return PNR_E_UNSUPPORTED;
    for (int i = 0; i < views->n_levels; ++i) if (!views->latent[i]) return PNR_E_NULL;
    LatPack lp = lat_layout(*views);
    if (out_bytes < lp.total) return PNR_E_WORKSPACE;
    if (((uintptr_t)out & 15) != 0) return PNR_E_ALIGN;
    if (level_offsets)
        for (int i = 0; i < PNR_MAX_LEVELS; ++i) level_offsets[i] = lp.off[i];
    if (dtype == PNR_BF16) hipLaunchKernelGGL(k_pack_latents<PNR_BF16>, dim3(1024), dim3(256), 0, (hipStream_t)stream, *views, lp, (char*)out);
    else hipLaunchKernelGGL(k_pack_latents<PNR_F16>, dim3(1024), dim3(256), 0, (hipStream_t)stream, *views, lp, (char*)out);
    PNR_LAUNCH_CHECK();
    return PNR_OK;
}
