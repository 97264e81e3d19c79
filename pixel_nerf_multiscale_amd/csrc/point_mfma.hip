// Fused per-point network for gfx950 (CDNA4): PixelNeRFNet.forward (models.py.backup2:155-282) + ResnetFC
// (resnetfc.py:173-236) in ONE persistent kernel, bf16 or fp16 MFMA with fp32 accumulation.
//
// Shape of the computation (d_hidden = 512):
//   * a workgroup = 4 waves (one per SIMD, up to 512 registers each) owns a tile of 128 points;
//     a wave owns 32 of them: they are the 32 COLUMNS (lanes) of v_mfma_f32_32x32x16.
//   * activations are kept TRANSPOSED, X^T = [512 features x 32 points] fp32, as 16 accumulator tiles
//     (256 registers) for the whole network: Y^T = W . X^T uses the PyTorch (out,in) weight as the A operand
//     and the previous layer's accumulators, converted in registers to bf16, as the B operand — the sum runs
//     over the accumulator's ROW index, so no lane movement and no LDS round trip for activations.
//   * the weights of the whole MLP are pre-packed (pnr_pack_mlp) into a linear stream of 1-KiB MFMA A-fragments in
//     exactly the order the kernel consumes them; all 4 waves consume the same stream, so it is staged through a
//     4-slot x 16-KiB LDS ring filled by LDS-DMA (global_load_lds_dwordx4) two stages ahead, one barrier per stage
//     of 16 MFMAs.  Every workgroup streams the same bytes in the same order => L2-resident across the XCD.
//   * pixel-aligned features: each lane projects its point, gathers 4 bilinear taps from the channels-last 16-bit
//     latent copy and writes the interpolated channels straight into the wave's LDS B-fragment image.
//   * multi-view: the first `combine_layer` blocks run once per source view on the same 128 points; the per-view
//     residual streams are parked in a caller-provided workspace and reduced (mean/max) in registers.
#include <type_traits>

#include "pnr_common.h"
#include "resblock_asm.inc"

namespace pnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned frag_t __attribute__((ext_vector_type(4)));   // one 16-bit x8 MFMA operand fragment

static constexpr int HID = 512;          // d_hidden the kernel is specialised for
static constexpr int NT = HID / 32;      // 16 feature tiles of 32 rows
static constexpr int STAGE_BYTES = 16384;
static constexpr int RING_SLOTS = 4;     // power of two; stage i+3 is loaded while stage i is consumed and stage i+1 read ahead
static constexpr int TILE_PTS = 128;
static constexpr int ZBUF_BYTES = 16384; // per wave: 16 k-steps x 64 lanes x 16 B (256 latent channels)
static constexpr int LDS_RING = 0;
static constexpr int LDS_Z = RING_SLOTS * STAGE_BYTES;
static constexpr int LDS_BTAB = LDS_Z + 4 * ZBUF_BYTES;

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
    static __device__ __forceinline__ uint32_t sat(uint32_t p) { return p; }   // bf16 has fp32's range
    static __device__ __forceinline__ uint16_t cvt(float a) { return (uint16_t)(pack(a, 0.f) & 0xffff); }
    static __device__ __forceinline__ float back(uint16_t v) { return lo(v); }
    // relu(pack(a0, a1)) straight from two accumulator (AGPR) registers, as ONE asm statement: hipcc cannot pull the
    // reads ahead of the conversion (which made it spill whole tiles to scratch around the snapshot)
    static __device__ __forceinline__ uint32_t snap2(float a0, float a1) {
        uint32_t out, tmp;
        asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3\n\tv_cvt_pk_bf16_f32 %0, %0, %1\n\tv_pk_max_i16 %0, %0, 0"
                     : "=&v"(out), "=&v"(tmp) : "a"(a0), "a"(a1));
        return out;
    }
    static __device__ __forceinline__ f32x16 mfma(frag_t a, frag_t b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Num<PNR_F16> {
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        f32x2 f = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, f16x2));       // v_cvt_pk_f16_f32
    }
    static __device__ __forceinline__ float lo(uint32_t p) { return (float)__builtin_bit_cast(f16x2, p)[0]; }
    static __device__ __forceinline__ float hi(uint32_t p) { return (float)__builtin_bit_cast(f16x2, p)[1]; }
    static __device__ __forceinline__ uint16_t one() { return 0x3C00; }
    static __device__ __forceinline__ uint32_t sat(uint32_t p) {          // see snap2
        s16x2 v = __builtin_bit_cast(s16x2, p);
        s16x2 m = {0x7BFF, 0x7BFF};
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(v, m));
    }
    static __device__ __forceinline__ uint16_t cvt(float a) { return (uint16_t)(pack(a, 0.f) & 0xffff); }
    static __device__ __forceinline__ float back(uint16_t v) { return lo(v); }
    static __device__ __forceinline__ uint32_t snap2(float a0, float a1) {
        uint32_t out, tmp;
        // after the relu every half is a non-negative fp16, whose integer order is its float order: min with 0x7BFF
        // (65504) saturates +inf (and NaN) instead of letting an overflowing activation poison the next layer
        asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3\n\tv_cvt_pk_f16_f32 %0, %0, %1\n\tv_pk_max_i16 %0, %0, 0\n\tv_pk_min_i16 %0, %0, %4"
                     : "=&v"(out), "=&v"(tmp) : "a"(a0), "a"(a1), "v"(0x7BFF7BFFu));
        return out;
    }
    static __device__ __forceinline__ f32x16 mfma(frag_t a, frag_t b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
// relu on two packed 16-bit floats: negative <=> sign bit <=> negative as int16 (v_pk_max_i16 with 0)
__device__ __forceinline__ uint32_t relu_pk(uint32_t p) {
    s16x2 v = __builtin_bit_cast(s16x2, p);
    s16x2 z = {0, 0};
    v = __builtin_elementwise_max(v, z);
    return __builtin_bit_cast(uint32_t, v);
}

// ---------------------------------------------------------------------------- stream layout (shared by pack + kernel)
// Stage = 16 fragments of 1 KiB.  Per (tile, view):  LIN_IN (S_in stages), then per phase-1 block
// [LIN_Z: SZ stages + 1 bias stage] [fc_1 bias stage] [16 chunks x (2 fc_0 + 2 fc_1 stages)];
// phase 2: per block [fc_1 bias stage][16 x 4]; LIN_OUT 2 stages.
// Projected mode (proj_T > 0; one source view, one small latent map of T texels): bilinear interpolation and lin_z are
// both linear, so lin_z_b(z(p)) = (W_z,b . Lat) . w(p) with w(p) the T-vector of the point's 4 tap weights.  The
// stream then carries the (512 x T) products M_b = W_z,b . Lat in place of W_z,b: ceil(T/16) k-steps instead of L/16,
// and the per-point latent gather disappears (the B operand is the tap-weight image).
struct Layout {
    int d_in, d_in_pad, D, S_in, L, SZ, n_blocks, nb1, nb2, P1, P2, btab_floats, proj_T, ZK, PV, Gg;   // PV: per-view copies of the P1 part (projected), else 1; Gg: 256-channel groups still gathered
    uint64_t btab_bytes, stream_bytes, total_bytes, proj_bytes;
};
static constexpr int BLOCK_STAGES = 1 + 16 * 4;

__host__ __device__ inline bool make_layout(const pnr_mlp& m, Layout& y, int proj_T = 0, int proj_views = 1, int proj_gathered = 0) {
    if (m.d_hidden != HID || m.d_out != 4 || m.d_latent <= 0 || (m.d_latent % 256) != 0 || m.d_latent > 1024) return false;
    if (m.n_blocks < 1 || m.n_blocks > PNR_MAX_BLOCKS || m.d_in < 1 || m.d_in > 78) return false;
    // LIN_IN k layout (chosen for the kernel, see the prologue): lane half h of k-step s, element j holds slot 8 s + j of
    //   [ sin(f_q v_i + h pi/2): q = 0..5, i = 0..D-1 | raw_t: t = 0..2 (h = 0: x_rot, h = 1: rotated view dir) | 1.0 | 0.. ]
    // with D = 3 (code over xyz, d_in 42) or 6 (code over xyz + dirs, d_in 78); 6 frequencies only.
    y.d_in = m.d_in;
    if (m.d_in != 42 && m.d_in != 78) return false;
    y.D = m.d_in == 78 ? 6 : 3;
    y.S_in = (6 * y.D + 4 + 7) / 8;                  // 3 or 5 k-steps
    y.d_in_pad = 16 * y.S_in;
    y.L = m.d_latent;
    // projected: the LAST latent level (256 channels, T <= 256 texels) is folded into lin_z; the Gg 256-channel groups of
    // the levels before it are still gathered.  lin_z k-steps: 16 per gathered group, then ceil(T/16) texel steps.
    if (proj_T < 0 || proj_T > 256 || proj_gathered < 0) return false;
    if (proj_T > 0 && m.d_latent != 256 * (proj_gathered + 1)) return false;
    y.proj_T = proj_T;
    y.Gg = proj_T > 0 ? proj_gathered : 0;
    y.ZK = proj_T > 0 ? ((proj_T + 15) / 16) * 16 : 0;  // texel extent (padded) of the projected part
    y.SZ = proj_T > 0 ? 16 * y.Gg + y.ZK / 16 : m.d_latent / 16;
    y.n_blocks = m.n_blocks;
    y.nb1 = m.combine_layer < m.n_blocks ? m.combine_layer : m.n_blocks;
    if (y.nb1 < 0) y.nb1 = 0;
    y.nb2 = m.n_blocks - y.nb1;
    y.P1 = y.S_in + y.nb1 * (y.SZ + 1 + BLOCK_STAGES);
    y.P2 = y.nb2 * BLOCK_STAGES + 2;
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

// accumulator-row permutation of the 32x32 MFMA: B/A element j of lane half h in k-step s of a 32-row tile
// is feature row 16 s + 8 (j>>2) + 4 h + (j&3)
__host__ __device__ inline int perm_k(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

// ---------------------------------------------------------------------------- pack kernels
// M_b[n][t] = sum_c lin_z[b].weight[n][c] * latent[c][t]  (fp32; view 0 of a single-level latent), t padded to ZK with 0
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
__global__ void k_pack_mlp(pnr_mlp m, Layout y, char* __restrict__ out, const float* __restrict__ M) {
    // bias table
    float* bt = (float*)out;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < y.btab_floats; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < m.n_blocks * HID) v = m.fc0_b[i / HID][i % HID];
        else if (i < m.n_blocks * HID + 4) v = m.lin_out_b[i - m.n_blocks * HID];
        bt[i] = v;
    }
    uint16_t* st = (uint16_t*)(out + y.btab_bytes);
    const int64_t n_elems = (int64_t)(y.PV * y.P1 + y.P2) * 16 * 64 * 8;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_elems; e += (int64_t)gridDim.x * blockDim.x) {
        int j = (int)(e & 7), lane = (int)((e >> 3) & 63), f = (int)((e >> 9) & 15);
        int stage = (int)(e >> 13);
        int pview = 0;                                   // which per-view copy (projected streams)
        if (stage < y.PV * y.P1) { pview = stage / y.P1; stage -= pview * y.P1; }
        else stage -= (y.PV - 1) * y.P1;                // phase 2 follows the last copy
        int r = lane & 31, h = lane >> 5;
        float val = 0.f;
        // ---- locate the stage
        int s = stage;
        bool done = false;
        if (s < y.S_in) {                               // LIN_IN: slot layout of make_layout, bias folded into slot 6D+3 (h=0: hi, h=1: lo)
            const int n = 32 * f + r, idx = 8 * s + j, D = y.D;
            if (idx < 6 * D) {
                const int q = idx / D, i = idx % D;
                val = m.lin_in_w[(size_t)n * m.d_in + D + (2 * q + h) * D + i];      // code.py: [x, sin f0, cos f0, sin f1, ...]
            } else if (idx < 6 * D + 3) {
                const int t = idx - 6 * D;
                const int col = h == 0 ? t : (D == 3 ? 3 + 12 * D + t : 3 + t);         // h=1: view dirs (raw or coded inputs 3..5)
                val = m.lin_in_w[(size_t)n * m.d_in + col];
            } else if (idx == 6 * D + 3) {
                val = h == 0 ? m.lin_in_b[n] : m.lin_in_b[n] - Num<DT>::back(Num<DT>::cvt(m.lin_in_b[n]));
            }
            done = true;
        } else s -= y.S_in;
        int b = 0;
        if (!done) {
            const int per1 = y.SZ + 1 + BLOCK_STAGES;
            int in_blk;
            if (stage < y.P1) { b = s / per1; in_blk = s % per1; }
            else {
                int s2 = stage - y.P1;
                if (s2 >= y.nb2 * BLOCK_STAGES) {       // LIN_OUT: 2 stages, rows 0..3 valid
                    int hf = s2 - y.nb2 * BLOCK_STAGES;
                    int t = 8 * hf + (f >> 1), sk = f & 1;
                    int k = 32 * t + perm_k(sk, h, j);
                    if (r < 4) val = m.lin_out_w[(size_t)r * HID + k];
                    in_blk = -1;
                } else { b = y.nb1 + s2 / BLOCK_STAGES; in_blk = y.SZ + 1 + s2 % BLOCK_STAGES; }
            }
            if (in_blk >= 0) {
                if (in_blk < y.SZ) {                    // LIN_Z k-step in_blk: natural k
                    int n = 32 * f + r, k = 16 * in_blk + 8 * h + j;
                    val = (M && in_blk >= 16 * y.Gg) ? M[(((size_t)pview * y.nb1 + b) * HID + n) * y.ZK + (k - 256 * y.Gg)]
                                                     : m.lin_z_w[b][(size_t)n * y.L + k];
                } else if (in_blk == y.SZ || in_blk == y.SZ + 1) {   // bias stages: k-slot 0 = hi, 1 = lo
                    const float* bp = (in_blk == y.SZ) ? m.lin_z_b[b] : m.fc1_b[b];
                    int n = 32 * f + r;
                    if (h == 0 && j == 0) val = bp[n];
                    else if (h == 0 && j == 1) val = bp[n] - Num<DT>::back(Num<DT>::cvt(bp[n]));
                } else {
                    // 64 chunk stages in the kernel's software-pipelined order: F(0) | F(1) G(0) | ... | F(15) G(14) | G(15),
                    // F(c) = the 2 fc_0 stages of chunk c (parts 0,1), G(c) = its 2 fc_1 stages (parts 2,3)
                    int q = in_blk - (y.SZ + 2);        // 0..63
                    int c, part;
                    if (q < 2) { c = 0; part = q; }
                    else if (q >= 62) { c = 15; part = 2 + (q - 62); }
                    else {
                        int jq = q - 2, cc = jq >> 2, pp = jq & 3;
                        if (pp < 2) { c = cc + 1; part = pp; } else { c = cc; part = pp; }
                    }
                    if (part < 2) {                     // fc_0 chunk c: rows 32c.., k tiles t = 8*part + f/2
                        int t = 8 * part + (f >> 1), sk = f & 1;
                        val = m.fc0_w[b][(size_t)(32 * c + r) * HID + 32 * t + perm_k(sk, h, j)];
                    } else {                            // fc_1 chunk c: output tiles tn = 8*(part-2) + f/2, k in chunk c
                        int tn = 8 * (part - 2) + (f >> 1), sk = f & 1;
                        val = m.fc1_w[b][(size_t)(32 * tn + r) * HID + 32 * c + perm_k(sk, h, j)];
                    }
                }
            }
        }
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
    int n_tiles, NS, combine_max;
    int S_in, SZ, n_blocks, nb1, P1, P2, btab_floats, d_in, proj, Gg;
    int ldP1, ldNS;                // the loader's stream: (P1, NS), or (NS*P1, 1) when every view has its own copy (projected)
    int use_code_viewdirs, num_freqs;
    float freq_factor;
};

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// One 1 KiB LDS-DMA piece (global_load_lds_dwordx4): lane l moves 16 B from g_base + lane_off + 1024 Q to LDS
// lds_dst + 1024 Q + 16 l (the instruction offset applies to both addresses).  Inline asm so that hipcc neither counts
// these loads in its own vmcnt bookkeeping nor drains them before barriers / ds_reads; they are retired by the counted
// s_waitcnt in begin_stage (cdna_hip_programming.md §5.7).  A piece costs the wave ~16 issue cycles plus ~4 per scalar
// instruction around it, so the per-piece addresses come from the immediate offset and the 4 pieces a wave owes per
// stage are spread between the stage's MFMAs.  (The hand-scheduled blocks of resblock_asm.inc save M0 once per block.)
template <int Q>
__device__ __forceinline__ void glds_piece(const char* g_base /* wave-uniform */, uint32_t lane_off, uint32_t lds_dst) {
    // M0 (the LDS destination base) is compiler-reserved: hipcc keeps its own value there across statements (it indexes
    // kernel-argument arrays with s_movrels in the gather) and writes it with the same `s_mov_b32 m0, sN` this code
    // uses, so every piece saves, sets and restores M0 inside its own statement (cdna_hip_programming.md §5.7).
    uint32_t keep;
    if (Q == 0) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(g_base), "s"(lds_dst) : "memory");
    if (Q == 1) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(g_base), "s"(lds_dst) : "memory");
    if (Q == 2) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(g_base), "s"(lds_dst) : "memory");
    if (Q == 3) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(lane_off), "s"(g_base), "s"(lds_dst) : "memory");
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
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef Num<DT> NM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    float* btab = (float*)(smem + LDS_BTAB);
    for (int i = tid; i < a.btab_floats; i += 256) btab[i] = a.btab[i];
    __syncthreads();

    // ---------------- weight-stream loader: every wave issues 4 x 1 KiB LDS-DMA per stage, PREFETCH+1 stages ahead
    // loader cursor: stage index in the packed stream, source-view pass it belongs to, and where that pass wraps to 0
    // (a non-last view repeats phase 1; the last view runs on into phase 2)
    int ld_idx = 0, ld_rep = 0, ld_slot = 0, st_slot = 0;
    int ld_wrap = (a.ldNS == 1) ? a.ldP1 + a.P2 : a.ldP1;
    const uint32_t gl_off = (uint32_t)(wv * 4096 + lane * 16);
    const uint32_t ring_lds = lds_addr(smem + LDS_RING) + wv * 4096;
    // piece Q (0..3) of the loader's current stage; the cursor advances after the 4th piece
    const char* dma_g = a.stream;                               // global base of the stage being loaded (uniform)
    uint32_t dma_l = __builtin_amdgcn_readfirstlane(ring_lds);  // LDS base of its slot (+ this wave's quarter)
    auto issue_piece = [&](auto qc) {
        constexpr int Q = decltype(qc)::value;
#ifdef PNR_STAMPS_FINE
        unsigned long long _ti; STAMP(_ti);
#endif
#ifndef PNR_X_NODMA          // timing experiment only: no weight DMA (results are garbage)
        glds_piece<Q>(dma_g, gl_off, dma_l);
#endif
#ifdef PNR_STAMPS_FINE
        { unsigned long long _t2; STAMP(_t2); st_acc[9] += _t2 - _ti; }
#endif
        if (Q == 3) {
            ld_slot = (ld_slot + 1) & (RING_SLOTS - 1);
            ++ld_idx;
            if (ld_idx == ld_wrap) {
                ld_idx = 0;
                ld_rep = (ld_rep + 1 == a.ldNS) ? 0 : ld_rep + 1;
                ld_wrap = (ld_rep == a.ldNS - 1) ? a.ldP1 + a.P2 : a.ldP1;
            }
            dma_g = a.stream + (size_t)ld_idx * STAGE_BYTES;
            dma_l = __builtin_amdgcn_readfirstlane(ring_lds + ld_slot * STAGE_BYTES);
        }
    };
    // Stage protocol.  On entry to stage i its fragments are PUBLISHED (all waves' DMA landed + a barrier passed)
    // and its first 8 fragments sit in registers A[0..7].  begin_stage(): wait until this wave's DMA of stage i+1
    // has landed (the 4 younger pieces = stage i+2 may stay in flight), barrier => stage i+1 is published and
    // may be read ahead into A during stage i.  During stage i every wave issues its 4 pieces of stage i+3, one after
    // MFMAs 1, 5, 9, 13 (PNR_DMA).  That slot held stage i-1, whose last readers passed the barrier of stage i.
    frag_t A[8];
    const char* cur;    // this lane's read base of the stage being consumed
    const char* nxt;    // ... of the next stage
    auto begin_stage = [&]() {
#ifdef PNR_STAMPS_FINE
        unsigned long long _tb; STAMP(_tb);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        { unsigned long long _t2; STAMP(_t2); st_acc[8] += _t2 - _tb; _tb = _t2; }
        __builtin_amdgcn_s_barrier();
        { unsigned long long _t2; STAMP(_t2); st_acc[10] += _t2 - _tb; }
#else
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#ifndef PNR_X_NOBARRIER      // timing experiment only
        __builtin_amdgcn_s_barrier();
#endif
#endif
        asm volatile("" ::: "memory");
        cur = smem + LDS_RING + st_slot * STAGE_BYTES + lane * 16;
        st_slot = (st_slot + 1) & (RING_SLOTS - 1);
        nxt = smem + LDS_RING + st_slot * STAGE_BYTES + lane * 16;
    };
    // fragment f of the current stage is in A[f & 7]; after using it, refill the register 8 fragments ahead
#define PNR_REFILL(f) A[(f) & 7] = *(const frag_t*)(((f) < 8 ? cur : nxt) + (((f) + 8) & 15) * 1024)
    // fragments f..f+3 have landed once at most 4 younger LDS reads are outstanding: one wait per 4 MFMAs instead of
    // the per-MFMA waits hipcc would insert (s_waitcnt simm16 0xC47F = lgkmcnt(4), vmcnt/expcnt untouched)
#ifdef PNR_X_NO_LDSWAIT   // A/B experiment: leave the LDS waits to hipcc (one per MFMA)
#define PNR_LDSWAIT(f) do { } while (0)
#else
#define PNR_LDSWAIT(f) do { if (((f) & 3) == 0) __builtin_amdgcn_s_waitcnt(0xC47F); } while (0)
#endif
#define PNR_DMA(f) do { if ((f) == 1) issue_piece(std::integral_constant<int, 0>{}); else if ((f) == 5) issue_piece(std::integral_constant<int, 1>{}); \
                        else if ((f) == 9) issue_piece(std::integral_constant<int, 2>{}); else if ((f) == 13) issue_piece(std::integral_constant<int, 3>{}); } while (0)

#pragma unroll
    for (int i = 0; i < RING_SLOTS - 1; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q == 0) issue_piece(std::integral_constant<int, 0>{});
            if (q == 1) issue_piece(std::integral_constant<int, 1>{});
            if (q == 2) issue_piece(std::integral_constant<int, 2>{});
            if (q == 3) issue_piece(std::integral_constant<int, 3>{});
        }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // stage 0 landed (stages 1, 2 may be in flight)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int f = 0; f < 8; ++f) A[f] = *(const frag_t*)(smem + LDS_RING + lane * 16 + f * 1024);

    f32x16 x[NT];          // X^T: 16 tiles of [32 features x 32 points], fp32 residual stream
    frag_t xb[NT][2];       // relu(X^T) as 16-bit B fragments (k-steps 0,1 of every tile)
    const uint32_t one2 = (uint32_t)NM::one() | ((uint32_t)NM::one() << 16);
    const frag_t b_bias = {h == 0 ? one2 : 0u, 0u, 0u, 0u};   // k-slots 0,1 = 1.0: picks (hi, lo) of a bias fragment
    char* zwave = smem + LDS_Z + wv * ZBUF_BYTES;

    auto stage_x = [&](frag_t B) {                 // x[tn] += A_tn . B for the 16 output tiles
        begin_stage();
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            PNR_LDSWAIT(f);
            x[f] = NM::mfma(A[f & 7], B, x[f]);
            PNR_REFILL(f); PNR_DMA(f);
        }
    };
    // n_lds k-steps whose B fragment is image [k-step][lane] in this wave's LDS buffer, then n_bias bias stages
    // accumulator tiles pinned to a[16t : 16t+15] + the loader/consumer cursor: operand list shared by the two
    // hand-scheduled blocks of resblock_asm.inc (tools/gen_resblock_asm.py documents the contract)
#define PNR_ASM_STATE_OPERANDS                                                                                          \
    "+{a[0:15]}"(x[0]), "+{a[16:31]}"(x[1]), "+{a[32:47]}"(x[2]), "+{a[48:63]}"(x[3]), "+{a[64:79]}"(x[4]),               \
    "+{a[80:95]}"(x[5]), "+{a[96:111]}"(x[6]), "+{a[112:127]}"(x[7]), "+{a[128:143]}"(x[8]), "+{a[144:159]}"(x[9]),      \
    "+{a[160:175]}"(x[10]), "+{a[176:191]}"(x[11]), "+{a[192:207]}"(x[12]), "+{a[208:223]}"(x[13]),                    \
    "+{a[224:239]}"(x[14]), "+{a[240:255]}"(x[15]), "+s"(st_), "+s"(li_), "+s"(ls_), "+s"(lr_), "+s"(lw_)
    const int asm_cfg = a.ldP1 | ((a.ldP1 + a.P2) << 12) | (a.ldNS << 24);
    const uint32_t ring_lane = lds_addr(smem + LDS_RING) + lane * 16;
    auto asm_resync = [&](int st_, int li_, int ls_, int lr_, int lw_) {     // state back from an asm block
        st_slot = st_; ld_idx = li_; ld_slot = ls_; ld_rep = lr_; ld_wrap = lw_;
        dma_g = a.stream + (size_t)ld_idx * STAGE_BYTES;
        dma_l = __builtin_amdgcn_readfirstlane(ring_lds + ld_slot * STAGE_BYTES);
#pragma unroll
        for (int f = 0; f < 8; ++f)                  // the next stage is published: its first 8 fragments into A
            A[f] = *(const frag_t*)(smem + LDS_RING + st_slot * STAGE_BYTES + lane * 16 + f * 1024);
    };
    // n_lds k-steps whose B fragment is image [k-step][lane] in this wave's LDS buffer, then n_bias (0/1) bias stages
    auto x_stages = [&](int n_lds, int n_bias, bool use_asm = true) {
#ifndef PNR_NO_ASM_RESBLOCK
        if (use_asm) {
            int st_ = st_slot, li_ = ld_idx, ls_ = ld_slot, lr_ = ld_rep, lw_ = ld_wrap;
            const int cfg2 = n_lds | (n_bias << 8);
            const uint32_t zaddr = lds_addr(zwave) + lane * 16;
            if (DT == PNR_BF16)
                asm volatile(PNR_XSTAGES_ASM_BF16 : PNR_ASM_STATE_OPERANDS
                             : "s"(asm_cfg), "s"(a.stream), "s"(ring_lds), "v"(ring_lane), "v"(gl_off), "v"(zaddr), "v"(b_bias.x), "s"(cfg2)
                             : PNR_RESBLOCK_CLOBBERS);
            else
                asm volatile(PNR_XSTAGES_ASM_F16 : PNR_ASM_STATE_OPERANDS
                             : "s"(asm_cfg), "s"(a.stream), "s"(ring_lds), "v"(ring_lane), "v"(gl_off), "v"(zaddr), "v"(b_bias.x), "s"(cfg2)
                             : PNR_RESBLOCK_CLOBBERS);
            asm_resync(st_, li_, ls_, lr_, lw_);
            return;
        }
#endif
        frag_t Bz = *(const frag_t*)(zwave + lane * 16);
#pragma unroll 1
        for (int it = 0; it < n_lds + n_bias; ++it) {
            const frag_t Bn = *(const frag_t*)(zwave + ((it + 1) & 15) * 1024 + lane * 16);
            stage_x(it < n_lds ? Bz : b_bias);
            Bz = Bn;
        }
    };
    // relu(X^T) -> 16-bit B fragments (see Num::snap2)
    auto snapshot = [&]() {
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMA's accumulator writes have retired
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                frag_t v;
                v.x = NM::snap2(x[t][8 * s + 0], x[t][8 * s + 1]);
                v.y = NM::snap2(x[t][8 * s + 2], x[t][8 * s + 3]);
                v.z = NM::snap2(x[t][8 * s + 4], x[t][8 * s + 5]);
                v.w = NM::snap2(x[t][8 * s + 6], x[t][8 * s + 7]);
                xb[t][s] = v;
            }
        }
    };
    // hipcc gives every MFMA of a kernel that needs AGPRs the AGPR form, and all 256 AGPRs hold X^T: for this chain it
    // parks one X^T tile in VGPRs (v_accvgpr_read/write, hidden under the MFMAs).  A VGPR-form inline-asm chain was
    // tried and is wrong by construction: the register allocator may put v_mov copies of the accumulator between two
    // asm statements, and nothing pads the MFMA->VALU hazard for it.
    auto chunk_from_xb = [&](f32x16 acc) -> f32x16 {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            begin_stage();
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                PNR_LDSWAIT(f);
                acc = NM::mfma(A[f & 7], xb[8 * half + (f >> 1)][f & 1], acc);
                PNR_REFILL(f); PNR_DMA(f);
            }
        }
        return acc;
    };
    auto load_hbias = [&](int b, int c) -> f32x16 {   // fc_0.bias rows of chunk c in accumulator row order
        f32x16 hacc;
        const float* bp = btab + b * HID + 32 * c + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = *(const float4*)(bp + 8 * q);
            hacc[4 * q + 0] = v.x; hacc[4 * q + 1] = v.y; hacc[4 * q + 2] = v.z; hacc[4 * q + 3] = v.w;
        }
        return hacc;
    };

    const bool small_idx = a.n_points < 0x7fffffffLL;
    const int n_gather = a.proj ? a.Gg : a.SZ / 16;       // 256-channel groups gathered per block (the LDS image holds one)
    const int p_steps = a.proj ? a.SZ - 16 * a.Gg : 0;    // texel k-steps of the projected last level
    // the last lin_z call of a block (a gathered group, or the projected part) runs as the prefix of the resblock asm
    const int n_groups = a.proj ? n_gather : n_gather - 1;  // gathered groups that go through separate x_stages calls
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        STAMP(st_t);
#ifdef PNR_STAMPS
        st_tile = st_t;
#endif
        const int64_t g = (int64_t)tile * TILE_PTS + wv * 32 + r;
        const bool live = g < a.n_points;
        const int64_t gc = live ? g : a.n_points - 1;
        // 64-bit divisions are ~100 instructions each on this machine: one object / point counts below 2^31 take 32-bit paths
        const int obj = a.vw.n_objs == 1 ? 0 : (small_idx ? (int)((uint32_t)gc / (uint32_t)a.pts_per_obj) : (int)(gc / a.pts_per_obj));
        float pu = 0.f, pv = 0.f;
        int v = 0, b = 0, view = 0;
        bool start = true;

        // ---- latent gather: group grp (256 channels) -> this wave's LDS B-fragment image [k-step][lane][8]
        auto gather = [&](int grp) {
            int ch0 = 0;
            for (int lvl = 0; lvl < a.vw.n_levels; ++lvl) {
                const int C = a.vw.lat_c[lvl], W = a.vw.lat_w[lvl], H = a.vw.lat_h[lvl];
                const int lo = ch0 > grp * 256 ? ch0 : grp * 256;
                const int hi = (ch0 + C) < (grp + 1) * 256 ? (ch0 + C) : (grp + 1) * 256;
                if (lo < hi) {
                    const Taps tp = bilinear_taps(pu, pv, W, H);
                    const char* lb = a.lat[lvl] + (size_t)view * H * W * C * 2;
                    // 4 k-steps (16 tap loads) in flight per iteration: the loop is latency-bound on L2 otherwise
                    for (int chb = lo + 8 * h; chb < hi; chb += 64) {
                        uint4 q[4][4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int ch = (chb + 16 * u < hi) ? chb + 16 * u : chb;       // clamp: level widths are multiples of 16
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                q[u][i] = *(const uint4*)(lb + ((size_t)tp.off[i] * C + (ch - ch0)) * 2);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int ch = chb + 16 * u;
                            if (ch < hi) {
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
                                const int ks = (ch - grp * 256) >> 4;
                                *(uint4*)(zwave + ks * 1024 + lane * 16) = o;
                            }
                        }
                    }
                }
                ch0 += C;
            }
        };

        // ---- projected mode: the B operand of lin_z is the point's tap-weight vector over the T texels,
        //      image [k-step][lane half][col][8]: texel t sits at k-step t/16, half (t/8)&1, element t&7
        auto tap_image = [&]() {
            const int ll = a.vw.n_levels - 1;
            const Taps tp = bilinear_taps(pu, pv, a.vw.lat_w[ll], a.vw.lat_h[ll]);
            const uint4 z4 = {0u, 0u, 0u, 0u};
            for (int s = 0; s < p_steps; ++s) *(uint4*)(zwave + s * 1024 + lane * 16) = z4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int t = tp.off[2 * h + i];
                const float w = tp.w[2 * h + i];
                if (w != 0.f)
                    *(uint16_t*)(zwave + (t >> 4) * 1024 + ((((t >> 3) & 1) * 32 + r) * 16) + (t & 7) * 2) = NM::cvt(w);
            }
        };

        while (true) {
            if (start) {
                start = false;
                // ---- per-view geometry (models.py.backup2:166-221)
                view = obj * a.NS + v;
                const Cam cam = load_cam(a.vw, view);
                float p[3], d[3], xr[3], dr[3];
                if (small_idx && a.src.rays) {               // fetch_point with a 32-bit ray index
                    const uint32_t ray = (uint32_t)gc / (uint32_t)a.src.K;
                    const float* rp = a.src.rays + (size_t)ray * 8;
                    const float zz = a.src.z[gc];
                    d[0] = rp[3]; d[1] = rp[4]; d[2] = rp[5];
                    p[0] = rp[0] + zz * d[0]; p[1] = rp[1] + zz * d[1]; p[2] = rp[2] + zz * d[2];
                } else {
                    fetch_point(a.src, gc, p, d);
                }
                rot3(cam.R, p, xr);
                rot3(cam.R, d, dr);
                project(cam, xr, pu, pv);
#ifdef PNR_STAMPS
                { float sink = pu + pv + xr[0] + dr[0]; asm volatile("" :: "v"(sink)); }
                STAMP_ACC(8, st_t);
#endif
                // ---- positional features -> wave-private LDS image [k-step][lane][8] (16-bit) in the slot layout of
                //      make_layout: every lane builds its own 8 entries per k-step (uniform code: lane half h takes the
                //      phase-h sines, h=0 the raw x_rot, h=1 the rotated view dir) and stores them with one ds_write_b128.
                {
                    const float ph = h ? 1.57079637f : 0.0f;
                    auto image = [&](auto dc) {
                        constexpr int D = decltype(dc)::value;
                        constexpr int SL = 8 * ((6 * D + 4 + 7) / 8);
                        const float vv[6] = {xr[0], xr[1], xr[2], dr[0], dr[1], dr[2]};
                        float val[SL];
#pragma unroll
                        for (int q = 0; q < 6; ++q) {
                            const float fq = a.freq_factor * (float)(1 << q);
#pragma unroll
                            for (int i = 0; i < D; ++i) val[q * D + i] = __sinf(fmaf(vv[i], fq, ph));
                        }
#pragma unroll
                        for (int t = 0; t < 3; ++t) val[6 * D + t] = h ? dr[t] : xr[t];
                        val[6 * D + 3] = 1.0f;                       // folded lin_in bias (h=0: hi part, h=1: lo part)
#pragma unroll
                        for (int i = 6 * D + 4; i < SL; ++i) val[i] = 0.f;
#pragma unroll
                        for (int s = 0; s < SL / 8; ++s) {
                            uint4 o;
                            o.x = NM::pack(val[8 * s + 0], val[8 * s + 1]); o.y = NM::pack(val[8 * s + 2], val[8 * s + 3]);
                            o.z = NM::pack(val[8 * s + 4], val[8 * s + 5]); o.w = NM::pack(val[8 * s + 6], val[8 * s + 7]);
                            *(uint4*)(zwave + s * 1024 + lane * 16) = o;
                        }
                    };
                    if (a.use_code_viewdirs) image(std::integral_constant<int, 6>{});
                    else image(std::integral_constant<int, 3>{});
                }
                STAMP_ACC(1, st_t);
                // ---- LIN_IN: x = W_in . features
                {
                    f32x16 zero;
#pragma unroll
                    for (int i = 0; i < 16; ++i) zero[i] = 0.f;
#pragma unroll
                    for (int t = 0; t < NT; ++t) x[t] = zero;
                }
#ifdef PNR_X_NO_ASM_LININ
                x_stages(a.S_in, 0, false);
#else
                x_stages(a.S_in, 0);
#endif
                if (a.proj && n_gather == 0) tap_image();          // fully projected: the image serves all blocks of this view
                else if (!a.proj && n_gather == 1) gather(0);
                STAMP_ACC(2, st_t);
            }
            // ---- x += lin_z[b](z)  (blocks before the view reduction only)
            if (b < a.nb1) {
                for (int grp = 0; grp < n_groups; ++grp) {         // all but the block's last lin_z part
                    gather(grp);
#ifdef PNR_X_NO_ASM_LINZ
                    x_stages(16, 0, false);
#else
                    x_stages(16, 0);
#endif
                }
                if (a.proj && n_gather > 0) tap_image();           // partial projection: the buffer was just used by the gather
                else if (!a.proj && n_gather > 1) gather(n_gather - 1);
            }
            STAMP_ACC(3, st_t);
            // ---- resblock: x += fc_1(relu(fc_0(relu(x)))) + biases  (resnetfc.py:53-62)
#ifndef PNR_NO_ASM_RESBLOCK
            if constexpr (true) {
                // hand-scheduled block (tools/gen_resblock_asm.py -> resblock_asm.inc): snapshot, fc_1-bias stage, 16 chunks
                int st_ = st_slot, li_ = ld_idx, ls_ = ld_slot, lr_ = ld_rep, lw_ = ld_wrap;
                const uint32_t bias_addr = lds_addr(btab) + b * (HID * 4) + h * 16;
                const uint32_t zaddr = lds_addr(zwave) + lane * 16;
                const int cfg2z = (b < a.nb1) ? ((a.proj ? p_steps : 16) | (1 << 8)) : 0;       // lin_z prefix: last part + bias stage
                if (DT == PNR_BF16)
                    asm volatile(PNR_RESBLOCK_ASM_BF16 : PNR_ASM_STATE_OPERANDS
                                 : "s"(asm_cfg), "s"(a.stream), "s"(ring_lds), "v"(ring_lane), "v"(gl_off), "v"(bias_addr), "v"(b_bias.x),
                                   "v"(zaddr), "s"(cfg2z)
                                 : PNR_RESBLOCK_CLOBBERS);
                else
                    asm volatile(PNR_RESBLOCK_ASM_F16 : PNR_ASM_STATE_OPERANDS
                                 : "s"(asm_cfg), "s"(a.stream), "s"(ring_lds), "v"(ring_lane), "v"(gl_off), "v"(bias_addr), "v"(b_bias.x),
                                   "v"(zaddr), "s"(cfg2z)
                                 : PNR_RESBLOCK_CLOBBERS);
                asm_resync(st_, li_, ls_, lr_, lw_);
            } else
#endif
            {
            snapshot();
            STAMP_ACC(4, st_t);
            x_stages(0, 1);                       // + fc_1.bias
            STAMP_ACC(5, st_t);
            {
                // same stage order as the asm block: F(0) | F(c+1) G(c) ... | G(15)
                f32x16 hacc = chunk_from_xb(load_hbias(b, 0));
#pragma unroll 1
                for (int c = 0; c < 16; ++c) {
                    f32x16 hnext = hacc;
                    if (c < 15) hnext = chunk_from_xb(load_hbias(b, c + 1));
                    frag_t hb[2];
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        hb[s].x = NM::sat(relu_pk(NM::pack(hacc[8 * s + 0], hacc[8 * s + 1])));
                        hb[s].y = NM::sat(relu_pk(NM::pack(hacc[8 * s + 2], hacc[8 * s + 3])));
                        hb[s].z = NM::sat(relu_pk(NM::pack(hacc[8 * s + 4], hacc[8 * s + 5])));
                        hb[s].w = NM::sat(relu_pk(NM::pack(hacc[8 * s + 6], hacc[8 * s + 7])));
                    }
                    hacc = hnext;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        begin_stage();
#pragma unroll
                        for (int f = 0; f < 16; ++f) {
                            const int tn = 8 * half + (f >> 1);
                            PNR_LDSWAIT(f);
                            x[tn] = NM::mfma(A[f & 7], hb[f & 1], x[tn]);
                            PNR_REFILL(f); PNR_DMA(f);
                        }
                    }
                }
            }
            }
            STAMP_ACC(6, st_t);
            ++b;
            // ---- multi-view reduction after the last per-view block (util.combine_interleaved, util.py:466-476)
            if (MULTIVIEW && b == a.nb1 && a.NS > 1) {
                // Per-view residual streams are parked in the workspace and reduced by the last view.  Both steps are asm
                // blocks on the pinned accumulator tiles (resblock_asm.inc): element-wise C++ on x made hipcc stage whole
                // tiles through VGPR tuples, spill them, and (with pinned tiles) emit illegal copies.
                const float4* slot0 = a.spill + ((size_t)(blockIdx.x * 4 + wv) * (a.NS - 1)) * 4096;
                const uint32_t lane16 = lane * 16;
#define PNR_X_TILES                                                                                                     \
    "+{a[0:15]}"(x[0]), "+{a[16:31]}"(x[1]), "+{a[32:47]}"(x[2]), "+{a[48:63]}"(x[3]), "+{a[64:79]}"(x[4]),               \
    "+{a[80:95]}"(x[5]), "+{a[96:111]}"(x[6]), "+{a[112:127]}"(x[7]), "+{a[128:143]}"(x[8]), "+{a[144:159]}"(x[9]),      \
    "+{a[160:175]}"(x[10]), "+{a[176:191]}"(x[11]), "+{a[192:207]}"(x[12]), "+{a[208:223]}"(x[13]),                    \
    "+{a[224:239]}"(x[14]), "+{a[240:255]}"(x[15])
                if (v < a.NS - 1) {
                    const float4* slot = slot0 + (size_t)v * 4096;
                    asm volatile(PNR_VIEWSPILL_ASM : PNR_X_TILES : "s"(slot), "v"(lane16) : PNR_RESBLOCK_CLOBBERS);
                    ++v; b = 0; start = true;
                    continue;
                }
                const int nm1 = a.NS - 1;
                const float inv = 1.0f / (float)a.NS;
                asm volatile(PNR_VIEWREDUCE_ASM : PNR_X_TILES : "s"(slot0), "s"(nm1), "s"(a.combine_max), "v"(lane16), "v"(inv)
                             : PNR_RESBLOCK_CLOBBERS);
#undef PNR_X_TILES
            }
            if (b == a.n_blocks) break;
        }

        // ---- lin_out(relu(x)), sigmoid / relu (models.py.backup2:274-281)
        f32x16 o;
#ifndef PNR_NO_ASM_RESBLOCK
        {
            // hand-scheduled: each tile's relu/convert sits right before the two MFMAs that consume it (resblock_asm.inc)
            int st_ = st_slot, li_ = ld_idx, ls_ = ld_slot, lr_ = ld_rep, lw_ = ld_wrap;
            float o0, o1, o2, o3;
            if (DT == PNR_BF16)
                asm volatile(PNR_LINOUT_ASM_BF16 : PNR_ASM_STATE_OPERANDS, "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
                             : "s"(asm_cfg), "s"(a.stream), "s"(ring_lds), "v"(ring_lane), "v"(gl_off) : PNR_RESBLOCK_CLOBBERS);
            else
                asm volatile(PNR_LINOUT_ASM_F16 : PNR_ASM_STATE_OPERANDS, "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
                             : "s"(asm_cfg), "s"(a.stream), "s"(ring_lds), "v"(ring_lane), "v"(gl_off) : PNR_RESBLOCK_CLOBBERS);
            asm_resync(st_, li_, ls_, lr_, lw_);
            const float* bo = btab + a.n_blocks * HID;
            o[0] = o0 + bo[0]; o[1] = o1 + bo[1]; o[2] = o2 + bo[2]; o[3] = o3 + bo[3];
        }
#else
        snapshot();
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = 0.f;
        if (h == 0) {
            const float* bo = btab + a.n_blocks * HID;
            o[0] = bo[0]; o[1] = bo[1]; o[2] = bo[2]; o[3] = bo[3];
        }
        o = chunk_from_xb(o);
#endif
        if (h == 0 && live) {                     // rows 0..3 of the output tile sit in registers 0..3 of lanes 0..31
            float4 res;
            res.x = 1.0f / (1.0f + __expf(-o[0]));
            res.y = 1.0f / (1.0f + __expf(-o[1]));
            res.z = 1.0f / (1.0f + __expf(-o[2]));
            res.w = fmaxf(o[3], 0.f);
            ((float4*)a.out)[g] = res;
        }
        STAMP_ACC(7, st_t);
#ifdef PNR_STAMPS
        st_acc[0] += st_t - st_tile;
#endif
    }
#ifdef PNR_STAMPS
    if (lane == 0)
        for (int i = 0; i < 12; ++i) atomicAdd(&g_stamps[i], st_acc[i]);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the run-ahead LDS-DMA before the LDS is released
    asm volatile("" :: "v"(A[0].x), "v"(A[1].x), "v"(A[2].x), "v"(A[3].x), "v"(A[4].x), "v"(A[5].x), "v"(A[6].x), "v"(A[7].x));
}
#undef PNR_REFILL
#undef PNR_LDSWAIT
#undef PNR_DMA

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

uint64_t point_mfma_workspace_bytes(const pnr_mlp* mlp, const pnr_views* vw) {
    uint64_t b = 256;
    if (vw->n_views > 1) b += (uint64_t)MAX_GRID * 4 * (vw->n_views - 1) * 4096 * sizeof(float4);
    return b;
}

int32_t point_mfma(const pnr_params* prm, const pnr_mlp* mlp, const pnr_views* vw, PointSrc src, int64_t n_points,
                   int64_t pts_per_obj, float* out, void* workspace, uint64_t ws_bytes, hipStream_t s) {
    Layout y;
    const int proj = mlp->packed_texels;
    const int last = vw->n_levels - 1;
    const int gathered = proj ? (mlp->d_latent - vw->lat_c[last]) / 256 : 0;
    if (proj && (vw->lat_c[last] != 256 || (mlp->d_latent - 256) % 256 != 0)) return PNR_E_PACKED;
    if (!make_layout(*mlp, y, proj, vw->n_views, gathered)) return proj ? PNR_E_PACKED : PNR_E_UNSUPPORTED;
    if (!mlp->packed || mlp->packed_dtype != prm->precision || mlp->packed_bytes < y.total_bytes) return PNR_E_PACKED;
    if (((uintptr_t)mlp->packed & 15) != 0) return PNR_E_ALIGN;
    if (proj) {      // the stream was packed for ONE view's latent map: it must be the map being rendered
        if (vw->n_objs != 1 || vw->lat_h[last] * vw->lat_w[last] != proj) return PNR_E_PACKED;
    }
    if (!proj || gathered > 0) {
        if (vw->packed_dtype != prm->precision) return PNR_E_PACKED;
        for (int i = 0; i < (proj ? last : vw->n_levels); ++i) {
            if (!vw->latent_packed[i]) return PNR_E_PACKED;
            if (((uintptr_t)vw->latent_packed[i] & 15) != 0) return PNR_E_ALIGN;
            if (vw->lat_c[i] % 16 != 0) return PNR_E_UNSUPPORTED;
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
    a.n_tiles = (int)((n_points + TILE_PTS - 1) / TILE_PTS);
    a.NS = vw->n_views; a.combine_max = mlp->combine_type == PNR_COMBINE_MAX;
    a.S_in = y.S_in; a.SZ = y.SZ; a.n_blocks = y.n_blocks; a.nb1 = y.nb1; a.P1 = y.P1; a.P2 = y.P2;
    a.btab_floats = y.btab_floats; a.d_in = mlp->d_in; a.proj = proj; a.Gg = y.Gg;
    a.ldP1 = proj ? y.PV * y.P1 : y.P1; a.ldNS = proj ? 1 : vw->n_views;
    a.use_code_viewdirs = prm->use_code_viewdirs; a.num_freqs = prm->num_freqs; a.freq_factor = prm->freq_factor;
    int grid = num_cus();
    if (grid > MAX_GRID) grid = MAX_GRID;
    if (grid > a.n_tiles) grid = a.n_tiles;
    const size_t lds = LDS_BTAB + (size_t)y.btab_floats * 4;
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
    if (!mlp || !vw || vw->n_objs != 1 || vw->n_views < 1 || vw->n_views > 8 || vw->n_levels < 1 || vw->n_levels > PNR_MAX_LEVELS)
        return 0;
    const int last = vw->n_levels - 1;
    if (!vw->latent[last] || vw->lat_c[last] != 256 || mlp->combine_layer < 1) return 0;
    int L = 0;
    for (int i = 0; i < vw->n_levels; ++i) L += vw->lat_c[i];
    if (L != mlp->d_latent || (L - 256) % 256 != 0) return 0;
    *gathered = (L - 256) / 256;
    int T = vw->lat_h[last] * vw->lat_w[last];
    // fully projected maps pay T/16 k-steps instead of 16 + the gather: worth it up to 256 texels
    return (T >= 4 && T <= 256) ? T : 0;
}

extern "C" uint64_t pnr_packed_mlp_projected_bytes(const pnr_mlp* mlp, const pnr_views* views) {
    Layout y;
    int gathered = 0;
    int T = projectable(mlp, views, &gathered);
    if (!T || !make_layout(*mlp, y, T, views->n_views, gathered)) return 0;
    return y.total_bytes;
}

extern "C" int32_t pnr_pack_mlp_projected(const pnr_mlp* mlp, const pnr_views* views, int32_t dtype, void* out,
                                          uint64_t out_bytes, void* stream) {
    if (!mlp || !views || !out) return PNR_E_NULL;
    int gathered = 0;
    int T = projectable(mlp, views, &gathered);
    Layout y;
    if (!T || !make_layout(*mlp, y, T, views->n_views, gathered)) return PNR_E_UNSUPPORTED;
    if (dtype != PNR_BF16 && dtype != PNR_F16) return PNR_E_UNSUPPORTED;
    if (out_bytes < y.total_bytes) return PNR_E_WORKSPACE;
    if (((uintptr_t)out & 15) != 0) return PNR_E_ALIGN;
    if (!mlp->lin_in_w || !mlp->lin_in_b || !mlp->lin_out_w || !mlp->lin_out_b) return PNR_E_NULL;
    for (int b = 0; b < mlp->n_blocks; ++b) {
        if (!mlp->fc0_w[b] || !mlp->fc0_b[b] || !mlp->fc1_w[b] || !mlp->fc1_b[b]) return PNR_E_NULL;
        if (b < y.nb1 && (!mlp->lin_z_w[b] || !mlp->lin_z_b[b])) return PNR_E_NULL;
    }
    float* M = (float*)((char*)out + y.btab_bytes + y.stream_bytes);
    int64_t n_out = (int64_t)y.PV * y.nb1 * HID * y.ZK;
    hipLaunchKernelGGL(k_project_latent, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *mlp, y,
                       views->latent[views->n_levels - 1], T, M);
    PNR_LAUNCH_CHECK();
    if (dtype == PNR_BF16) hipLaunchKernelGGL(k_pack_mlp<PNR_BF16>, dim3(2048), dim3(256), 0, (hipStream_t)stream, *mlp, y, (char*)out, (const float*)M);
    else hipLaunchKernelGGL(k_pack_mlp<PNR_F16>, dim3(2048), dim3(256), 0, (hipStream_t)stream, *mlp, y, (char*)out, (const float*)M);
    PNR_LAUNCH_CHECK();
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
