#!/usr/bin/env python3
"""
Generates pixel_nerf_multiscale_amd/csrc/resblock_asm.inc: the resblock of the fused point kernel
(x += fc_1(relu(fc_0(relu(x)))) + biases, reference resnetfc.py:53-62) as ONE hand-scheduled gfx950 asm block per
MFMA dtype.  hipcc's version of the same loop runs ~3300 cycles per 64-MFMA chunk (AGPR-tile parking, per-MFMA
waits, ~50 scalar ops per chunk); this schedule runs ~2400 (tools/dev/ubench).

Contract with k_point_mfma (csrc/point_mfma.hip, PNR_ASM_RESBLOCK):
  * x tiles are pinned: tile t = a[16t : 16t+15] (operands %0..%15, "+{a[..]}").
  * entry: the next stage to consume (this block's fc_1-bias stage) is published, DMA of the two stages after it is
    in flight, nothing is assumed about the A-fragment registers (they are reloaded here).
  * in/out scalars: %16 st_slot, %17 ld_idx, %18 ld_slot, %19 ld_rep, %20 ld_wrap   (loader cursor, see issue_piece)
  * inputs: %21 cfg = P1 | (P1+P2)<<12 | NS<<24 (loader's view of the stream), %22 stream base (s64), %23 ring LDS base + wave*4096 (s),
            %24 ring LDS base + lane*16 (v), %25 DMA lane offset wave*4096 + lane*16 (v),
            %26 LDS address of fc_0.bias[block] + 16*(lane>>5) (v), %27 bias B-fragment dword 0 (v),
            %28 lin_z B image address (v), %29 lin_z cfg2 = n_lds | bias<<8, 0 = no lin_z prefix (s)
  * exit: all LDS reads drained, accumulators readable, 65 stages consumed, cursor advanced.
Register use inside (all declared as clobbers): v10-13 slot read bases, v14 bias address, v15 DMA lane offset,
v16-19 bias B fragment, v40-55 and v72-87 the two chunk accumulators (VGPR-form MFMA), v60-67 relu(h) fragments, v68-71 temps,
v96-127 A-fragment ring (8 x 4), v128-255 relu(x) fragments (16 tiles x 2 k-steps x 4), s20-s31, s33-s43 (s32 is the ABI stack pointer: left alone).
"""
import os
import sys


DIAG = os.environ.get("PNR_ASM_DIAG", "")      # timing experiments only: 'nobarrier', 'nodma', 'nowait' (results are garbage)


def A(i):
    return f"v[{96 + 4 * i}:{99 + 4 * i}]"


def XB(t, s):
    b = 128 + (t * 2 + s) * 4
    return f"v[{b}:{b + 3}]"


def gen(dt):
    mfma = {"bf16": "v_mfma_f32_32x32x16_bf16", "f16": "v_mfma_f32_32x32x16_f16"}[dt]
    cvt = {"bf16": "v_cvt_pk_bf16_f32", "f16": "v_cvt_pk_f16_f32"}[dt]
    L = []
    e = L.append

    def relu_pack(dst, a0, a1, tmp):
        e(f"v_accvgpr_read_b32 v{tmp}, a{a0}")
        e(f"v_accvgpr_read_b32 v{tmp + 1}, a{a1}")
        e(f"{cvt} v{dst}, v{tmp}, v{tmp + 1}")
        e(f"v_pk_max_i16 v{dst}, v{dst}, 0")
        if dt == "f16":
            e(f"v_pk_min_i16 v{dst}, v{dst}, s38")          # saturate +inf/NaN to 65504

    def loader_advance():
        e("s_add_u32 s21, s21, 1")
        e("s_add_u32 s24, s24, 0x4000")
        e("s_addc_u32 s25, s25, 0")
        e("s_cmp_lg_u32 s21, s29")
        e("s_cbranch_scc1 2f")
        e("s_mov_b32 s21, 0")
        e("s_mov_b64 s[24:25], s[36:37]")
        e("s_add_u32 s23, s23, 1")
        e("s_cmp_lg_u32 s23, s28")
        e("s_cselect_b32 s23, s23, 0")
        e("s_sub_u32 s35, s28, 1")
        e("s_cmp_eq_u32 s23, s35")
        e("s_cselect_b32 s29, s27, s26")
        e("2:")

    def dma(f, k):
        if f % 4 != 1:
            return
        q = f >> 2
        if q == 0:
            e(f"s_mov_b32 m0, s{40 + k}")
            e("s_nop 0")
        e(f"global_load_lds_dwordx4 v15, s[24:25]" + (f" offset:{q * 1024}" if q else ""))
        if q == 3:
            loader_advance()

    def begin_stage():
        e("s_waitcnt vmcnt(4)")
        e("s_barrier")

    def refill(f, k):
        base = 10 + k if f < 8 else 10 + ((k + 1) & 3)
        e(f"ds_read_b128 {A(f & 7)}, v{base} offset:{((f + 8) & 15) * 1024}")

    # ---------------------------------------------------------------- setup
    e("s_nop 15")
    e("s_nop 15")                                            # accumulator writes of the caller's last MFMAs retired
    e("s_mov_b32 s39, m0")                                   # hipcc may keep a value in M0 across the statement
    e("s_mov_b32 s20, %16")
    e("s_mov_b32 s21, %17")
    e("s_mov_b32 s22, %18")
    e("s_mov_b32 s23, %19")
    e("s_mov_b32 s29, %20")
    e("s_and_b32 s26, %21, 0xfff")
    e("s_bfe_u32 s27, %21, 0xc000c")                         # offset 12, width 12
    e("s_bfe_u32 s28, %21, 0x80018")                         # offset 24, width 8
    e("s_mov_b64 s[36:37], %22")
    e("s_lshl_b32 s35, s21, 14")
    e("s_add_u32 s24, s36, s35")
    e("s_addc_u32 s25, s37, 0")
    if dt == "f16":
        e("s_mov_b32 s38, 0x7bff7bff")
    # optional prefix (%29 != 0): the block's lin_z stages 'x += M . B' + lin_z.bias, B image at %28 — one asm entry/exit
    # and one pipeline refill less per block than a separate x-stages call
    e("s_cmp_eq_u32 %29, 0")
    e("s_cbranch_scc1 5f")
    xstages_core(e, mfma, "%28", "%29")
    e("5:")
    for k in range(4):                                       # stage k of a chunk uses slot (st_slot+1+k)&3; the bias stage = k 3
        e(f"s_add_u32 s35, s20, {1 + k}")
        e("s_and_b32 s35, s35, 3")
        e("s_lshl_b32 s35, s35, 14")
        e(f"v_add_u32 v{10 + k}, s35, %24")
        e(f"s_add_u32 s35, s22, {1 + k}")
        e("s_and_b32 s35, s35, 3")
        e("s_lshl_b32 s35, s35, 14")
        e(f"s_add_u32 s{40 + k}, s35, %23")
    e("v_mov_b32 v14, %26")
    e("v_mov_b32 v15, %25")
    e("v_mov_b32 v16, %27")
    e("v_mov_b32 v17, 0")
    e("v_mov_b32 v18, 0")
    e("v_mov_b32 v19, 0")
    # chunk 0's fc_0.bias rows into the chunk accumulator, then the first 8 fragments of the bias stage
    for q in range(4):
        e(f"ds_read_b128 v[{40 + 4 * q}:{43 + 4 * q}], v14 offset:{32 * q}")
    e("v_add_u32 v14, 0x80, v14")
    for i in range(8):
        e(f"ds_read_b128 {A(i)}, v13 offset:{i * 1024}")

    # ---------------------------------------------------------------- snapshot interleaved with the fc_1-bias stage
    begin_stage()
    for t in range(16):
        for s in range(2):
            for p in range(4):
                relu_pack(128 + (t * 2 + s) * 4 + p, 16 * t + 8 * s + 2 * p, 16 * t + 8 * s + 2 * p + 1, 68 + 2 * (p & 1))
        f = t
        if f % 4 == 0:
            e("s_waitcnt lgkmcnt(4)")
        e(f"{mfma} a[{16 * t}:{16 * t + 15}], {A(f & 7)}, v[16:19], a[{16 * t}:{16 * t + 15}]")
        refill(f, 3)
        dma(f, 3)

    # ---------------------------------------------------------------- 16 chunks, software-pipelined
    # Stage order of a block (the packer follows it, k_pack_mlp): bias | F(0) | F(1) G(0) | F(2) G(1) | ... | F(15) G(14) | G(15)
    # with F(c) = the 2 fc_0 stages of chunk c (chunk accumulator += W0[c] . relu(x)) and G(c) = its 2 fc_1 stages
    # (x += W1[:, c] . relu(h_c)).  Two chunk accumulators alternate (v40-55 / v72-87): while F(c+1) runs on one, the
    # other — finished 32 MFMAs earlier — is converted to relu(h_c) in the MFMA gaps and reloaded with the fc_0.bias
    # rows of chunk c+2, so no MFMA waits for a conversion and no s_nop pads the chain's tail.
    # Stage index i within the block (bias stage = 0) consumes ring slot (st_slot + i) & 3: refill/dma position k = (i-1) & 3.
    ACC = (40, 72)

    def fc0_stage(acc, i, half, hook=None, first_wait=4):
        k = (i - 1) & 3
        begin_stage()
        for f in range(16):
            if f % 4 == 0:
                e(f"s_waitcnt lgkmcnt({first_wait if f == 0 else 4})")
            e(f"{mfma} v[{acc}:{acc + 15}], {A(f & 7)}, {XB(8 * half + (f >> 1), f & 1)}, v[{acc}:{acc + 15}]")
            refill(f, k)
            dma(f, k)
            if hook:
                hook(f)

    def fc1_stage(i, half, first_wait=4):
        k = (i - 1) & 3
        begin_stage()
        for f in range(16):
            tn = 8 * half + (f >> 1)
            if f % 4 == 0:
                e(f"s_waitcnt lgkmcnt({first_wait if f == 0 else 4})")
            e(f"{mfma} a[{16 * tn}:{16 * tn + 15}], {A(f & 7)}, v[{60 + 4 * (f & 1)}:{63 + 4 * (f & 1)}], a[{16 * tn}:{16 * tn + 15}]")
            refill(f, k)
            dma(f, k)

    def convert_hook(acc, reload):
        """relu(h) of the finished accumulator -> v60-67, one register per MFMA gap (gaps 2..9); then (gap 15) the next
        fc_0.bias rows into it.  The 4 bias loads are younger than that stage's refills: the next stage's first wait
        allows 8 outstanding."""
        def hook(f):
            if 2 <= f <= 9:
                i = f - 2
                e(f"{cvt} v{60 + i}, v{acc + 2 * i}, v{acc + 2 * i + 1}")
                e(f"v_pk_max_i16 v{60 + i}, v{60 + i}, 0")
                if dt == "f16":
                    e(f"v_pk_min_i16 v{60 + i}, v{60 + i}, s38")
            if f == 15 and reload:
                for q in range(4):
                    e(f"ds_read_b128 v[{acc + 4 * q}:{acc + 4 * q + 3}], v14 offset:{32 * q}")
                e("v_add_u32 v14, 0x80, v14")
        return hook

    def body(c_parity, i0, reload=True):
        """F(c+1) into ACC[1-c_parity] with the conversion of ACC[c_parity] riding in its first stage, then G(c)."""
        fc0_stage(ACC[1 - c_parity], i0, 0, hook=convert_hook(ACC[c_parity], reload))
        fc0_stage(ACC[1 - c_parity], i0 + 1, 1, first_wait=8 if reload else 4)
        fc1_stage(i0 + 2, 0)
        fc1_stage(i0 + 3, 1)

    # chunk 1's fc_0.bias rows into the second accumulator (chunk 0's went into v40-55 in the setup)
    for q in range(4):
        e(f"ds_read_b128 v[{72 + 4 * q}:{75 + 4 * q}], v14 offset:{32 * q}")
    e("v_add_u32 v14, 0x80, v14")
    fc0_stage(ACC[0], 1, 0, first_wait=8)                     # F(0): stage indices 1, 2 (the 4 loads above are younger than its fragments)
    fc0_stage(ACC[0], 2, 1)
    e("s_mov_b32 s34, 7")
    e("1:")
    body(0, 3)                                               # c even:  F(c+1) on v72.., h_c from v40..
    body(1, 3)                                               # c odd :  F(c+1) on v40.., h_c from v72..  (stage positions repeat mod 4)
    e("s_sub_u32 s34, s34, 1")
    e("s_cmp_lg_u32 s34, 0")
    e("s_cbranch_scc1 1b")
    body(0, 3, reload=False)                                 # c = 14: F(15) on v72.., no chunk 16 to preload
    # G(15): convert v72.. (its chain's last MFMA is 32 MFMAs back), then the last two fc_1 stages
    begin_stage()
    for i in range(8):
        e(f"{cvt} v{60 + i}, v{72 + 2 * i}, v{73 + 2 * i}")
        e(f"v_pk_max_i16 v{60 + i}, v{60 + i}, 0")
        if dt == "f16":
            e(f"v_pk_min_i16 v{60 + i}, v{60 + i}, s38")
    k = (63 - 1) & 3
    for f in range(16):
        tn = f >> 1
        if f % 4 == 0:
            e("s_waitcnt lgkmcnt(4)")
        e(f"{mfma} a[{16 * tn}:{16 * tn + 15}], {A(f & 7)}, v[{60 + 4 * (f & 1)}:{63 + 4 * (f & 1)}], a[{16 * tn}:{16 * tn + 15}]")
        refill(f, k)
        dma(f, k)
    fc1_stage(64, 1)

    # ---------------------------------------------------------------- exit
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 15")
    e("s_nop 15")
    e("s_mov_b32 m0, s39")
    e("s_add_u32 s20, s20, 1")
    e("s_and_b32 %16, s20, 3")
    e("s_mov_b32 %17, s21")
    e("s_add_u32 s22, s22, 1")
    e("s_and_b32 %18, s22, 3")
    e("s_mov_b32 %19, s23")
    e("s_mov_b32 %20, s29")
    return L


def xstages_core(e, mfma, baddr, cfg2):
    """The stage loop of gen_xstages, entered with the cursor in s20-s29/s[24:25]/s[36:37]: n_lds = cfg2 & 0xff stages
    with B from the LDS image at `baddr` (+1024 per stage), then (cfg2 bit 8) one bias stage with B = (%27,0,0,0).
    Leaves s20 (st_slot) and s22 (ld_slot) advanced and masked; uses v10, v11, v14-v19, v44-v47, s34, s35, s40."""

    def loader_advance():
        e("s_add_u32 s21, s21, 1")
        e("s_add_u32 s24, s24, 0x4000")
        e("s_addc_u32 s25, s25, 0")
        e("s_cmp_lg_u32 s21, s29")
        e("s_cbranch_scc1 2f")
        e("s_mov_b32 s21, 0")
        e("s_mov_b64 s[24:25], s[36:37]")
        e("s_add_u32 s23, s23, 1")
        e("s_cmp_lg_u32 s23, s28")
        e("s_cselect_b32 s23, s23, 0")
        e("s_sub_u32 s35, s28, 1")
        e("s_cmp_eq_u32 s23, s35")
        e("s_cselect_b32 s29, s27, s26")
        e("2:")

    def stage_body(breg):
        # cur base v10, nxt base v11, M0 value s40, B fragment in v[breg:breg+3]
        e("s_waitcnt vmcnt(4)")
        e("s_barrier")
        for f in range(16):
            if f % 4 == 0:
                e("s_waitcnt lgkmcnt(4)")
            e(f"{mfma} a[{16 * f}:{16 * f + 15}], {A(f & 7)}, v[{breg}:{breg + 3}], a[{16 * f}:{16 * f + 15}]")
            base = 10 if f < 8 else 11
            e(f"ds_read_b128 {A(f & 7)}, v{base} offset:{((f + 8) & 15) * 1024}")
            if f % 4 == 1:
                q = f >> 2
                if q == 0:
                    e("s_mov_b32 m0, s40")
                    e("s_nop 0")
                e("global_load_lds_dwordx4 v15, s[24:25]" + (f" offset:{q * 1024}" if q else ""))
                if q == 3:
                    loader_advance()

    def advance_slots():
        e("s_add_u32 s20, s20, 1")
        e("s_and_b32 s20, s20, 3")
        e("s_add_u32 s22, s22, 1")
        e("s_and_b32 s22, s22, 3")
        e("v_mov_b32 v10, v11")                              # cur <- nxt
        e("s_add_u32 s35, s20, 1")
        e("s_and_b32 s35, s35, 3")
        e("s_lshl_b32 s35, s35, 14")
        e("v_add_u32 v11, s35, %24")                          # nxt
        e("s_lshl_b32 s35, s22, 14")
        e("s_add_u32 s40, s35, %23")                          # DMA destination of the stage being loaded

    e(f"s_and_b32 s34, {cfg2}, 0xff")                         # n_lds
    e("s_lshl_b32 s35, s20, 14")
    e("v_add_u32 v10, s35, %24")                              # cur
    e("s_add_u32 s35, s20, 1")
    e("s_and_b32 s35, s35, 3")
    e("s_lshl_b32 s35, s35, 14")
    e("v_add_u32 v11, s35, %24")                              # nxt
    e("s_lshl_b32 s35, s22, 14")
    e("s_add_u32 s40, s35, %23")
    e(f"v_mov_b32 v14, {baddr}")
    e("v_mov_b32 v15, %25")
    e("ds_read_b128 v[16:19], v14")                           # B of the first k-step
    e("v_add_u32 v14, 0x400, v14")
    for i in range(8):
        e(f"ds_read_b128 {A(i)}, v10 offset:{i * 1024}")
    e("s_cmp_eq_u32 s34, 0")
    e("s_cbranch_scc1 3f")
    e("1:")
    e("ds_read_b128 v[44:47], v14")                           # next k-step's B (the one past the last is never used)
    e("v_add_u32 v14, 0x400, v14")
    stage_body(16)
    advance_slots()
    e("s_waitcnt lgkmcnt(8)")                                 # next B landed (8 younger fragment reads may be in flight)
    e("s_nop 7")                                              # the stage's last MFMAs have read the old B
    e("v_mov_b32 v16, v44")
    e("v_mov_b32 v17, v45")
    e("v_mov_b32 v18, v46")
    e("v_mov_b32 v19, v47")
    e("s_sub_u32 s34, s34, 1")
    e("s_cmp_lg_u32 s34, 0")
    e("s_cbranch_scc1 1b")
    e("3:")
    e(f"s_bitcmp1_b32 {cfg2}, 8")                             # bias stage requested?
    e("s_cbranch_scc0 4f")
    e("v_mov_b32 v16, %27")
    e("v_mov_b32 v17, 0")
    e("v_mov_b32 v18, 0")
    e("v_mov_b32 v19, 0")
    e("s_nop 1")
    stage_body(16)
    advance_slots()
    e("4:")


def gen_xstages(dt):
    """n_lds stages 'x[tn] += A_tn . B' with B = the k-step image [k][lane] at LDS address %26 (+1024 per stage, 16 k-steps
    per 256-channel group -> the caller passes the address of the first k-step), then (cfg2 bit 8) one bias stage.
    Operands: %0-%15 x tiles, %16-%20 cursor in/out (as the resblock), %21 cfg, %22 stream, %23 ring+wave*4096 (s),
    %24 ring+lane*16 (v), %25 DMA lane offset (v), %26 B image address (v), %27 bias B dword 0 (v), %28 cfg2 = n_lds | bias<<8."""
    mfma = {"bf16": "v_mfma_f32_32x32x16_bf16", "f16": "v_mfma_f32_32x32x16_f16"}[dt]
    L = []
    e = L.append
    e("s_nop 15")
    e("s_nop 15")
    e("s_mov_b32 s39, m0")
    e("s_mov_b32 s20, %16")
    e("s_mov_b32 s21, %17")
    e("s_mov_b32 s22, %18")
    e("s_mov_b32 s23, %19")
    e("s_mov_b32 s29, %20")
    e("s_and_b32 s26, %21, 0xfff")
    e("s_bfe_u32 s27, %21, 0xc000c")
    e("s_bfe_u32 s28, %21, 0x80018")
    e("s_mov_b64 s[36:37], %22")
    e("s_lshl_b32 s35, s21, 14")
    e("s_add_u32 s24, s36, s35")
    e("s_addc_u32 s25, s37, 0")
    xstages_core(e, mfma, "%26", "%28")
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 15")
    e("s_nop 15")
    e("s_mov_b32 m0, s39")
    e("s_mov_b32 %16, s20")
    e("s_mov_b32 %17, s21")
    e("s_mov_b32 %18, s22")
    e("s_mov_b32 %19, s23")
    e("s_mov_b32 %20, s29")
    return L


def gen_linout(dt):
    """lin_out(relu(x)): the snapshot of tile t (relu + 16-bit pack of a[16t:16t+15]) is emitted right before the two MFMAs
    that consume it, so the conversion VALU work hides the chain's MFMAs; 2 stages, one accumulator (v40-55, rows 0..3 of
    the output tile are valid), returned in %21-%24 (the caller adds lin_out.bias and applies sigmoid / relu).
    Operands: %0-%15 x tiles, %16-%20 cursor in/out, %21-%24 out (=&v), %25 cfg, %26 stream, %27 ring+wave*4096 (s),
    %28 ring+lane*16 (v), %29 DMA lane offset (v)."""
    mfma = {"bf16": "v_mfma_f32_32x32x16_bf16", "f16": "v_mfma_f32_32x32x16_f16"}[dt]
    cvt = {"bf16": "v_cvt_pk_bf16_f32", "f16": "v_cvt_pk_f16_f32"}[dt]
    L = []
    e = L.append

    def loader_advance():
        e("s_add_u32 s21, s21, 1")
        e("s_add_u32 s24, s24, 0x4000")
        e("s_addc_u32 s25, s25, 0")
        e("s_cmp_lg_u32 s21, s29")
        e("s_cbranch_scc1 2f")
        e("s_mov_b32 s21, 0")
        e("s_mov_b64 s[24:25], s[36:37]")
        e("s_add_u32 s23, s23, 1")
        e("s_cmp_lg_u32 s23, s28")
        e("s_cselect_b32 s23, s23, 0")
        e("s_sub_u32 s35, s28, 1")
        e("s_cmp_eq_u32 s23, s35")
        e("s_cselect_b32 s29, s27, s26")
        e("2:")

    def relu_pack(dst, a0, a1, tmp):
        e(f"v_accvgpr_read_b32 v{tmp}, a{a0}")
        e(f"v_accvgpr_read_b32 v{tmp + 1}, a{a1}")
        e(f"{cvt} v{dst}, v{tmp}, v{tmp + 1}")
        e(f"v_pk_max_i16 v{dst}, v{dst}, 0")
        if dt == "f16":
            e(f"v_pk_min_i16 v{dst}, v{dst}, s38")

    e("s_nop 15")
    e("s_nop 15")
    e("s_mov_b32 s39, m0")
    e("s_mov_b32 s20, %16")
    e("s_mov_b32 s21, %17")
    e("s_mov_b32 s22, %18")
    e("s_mov_b32 s23, %19")
    e("s_mov_b32 s29, %20")
    e("s_and_b32 s26, %25, 0xfff")
    e("s_bfe_u32 s27, %25, 0xc000c")
    e("s_bfe_u32 s28, %25, 0x80018")
    e("s_mov_b64 s[36:37], %26")
    e("s_lshl_b32 s35, s21, 14")
    e("s_add_u32 s24, s36, s35")
    e("s_addc_u32 s25, s37, 0")
    if dt == "f16":
        e("s_mov_b32 s38, 0x7bff7bff")
    for k in range(3):                                       # read bases of stage 0, stage 1 and the stage after (read-ahead only)
        e(f"s_add_u32 s35, s20, {k}")
        e("s_and_b32 s35, s35, 3")
        e("s_lshl_b32 s35, s35, 14")
        e(f"v_add_u32 v{10 + k}, s35, %28")
    for k in range(2):                                       # DMA destinations of the two stages loaded meanwhile
        e(f"s_add_u32 s35, s22, {k}")
        e("s_and_b32 s35, s35, 3")
        e("s_lshl_b32 s35, s35, 14")
        e(f"s_add_u32 s{40 + k}, s35, %27")
    e("v_mov_b32 v15, %29")
    for i in range(16):
        e(f"v_mov_b32 v{40 + i}, 0")
    for i in range(8):
        e(f"ds_read_b128 {A(i)}, v10 offset:{i * 1024}")
    for half in range(2):
        e("s_waitcnt vmcnt(4)")
        e("s_barrier")
        for f in range(16):
            t = 8 * half + (f >> 1)
            if f % 2 == 0:
                for sidx in range(2):
                    for pidx in range(4):
                        relu_pack(128 + (t * 2 + sidx) * 4 + pidx, 16 * t + 8 * sidx + 2 * pidx, 16 * t + 8 * sidx + 2 * pidx + 1, 68 + 2 * (pidx & 1))
            if f % 4 == 0:
                e("s_waitcnt lgkmcnt(4)")
            e(f"{mfma} v[40:55], {A(f & 7)}, {XB(t, f & 1)}, v[40:55]")
            base = 10 + half if f < 8 else 11 + half
            e(f"ds_read_b128 {A(f & 7)}, v{base} offset:{((f + 8) & 15) * 1024}")
            if f % 4 == 1:
                q = f >> 2
                if q == 0:
                    e(f"s_mov_b32 m0, s{40 + half}")
                    e("s_nop 0")
                e("global_load_lds_dwordx4 v15, s[24:25]" + (f" offset:{q * 1024}" if q else ""))
                if q == 3:
                    loader_advance()
    e("s_waitcnt lgkmcnt(0)")
    e("s_nop 15")
    e("s_nop 15")
    e("s_nop 15")
    for i in range(4):
        e(f"v_mov_b32 %{21 + i}, v{40 + i}")
    e("s_mov_b32 m0, s39")
    e("s_add_u32 s20, s20, 2")
    e("s_and_b32 %16, s20, 3")
    e("s_mov_b32 %17, s21")
    e("s_add_u32 s22, s22, 2")
    e("s_and_b32 %18, s22, 3")
    e("s_mov_b32 %19, s23")
    e("s_mov_b32 %20, s29")
    return L


def gen_viewspill():
    """Park this view's residual stream: x (16 tiles) -> workspace slot, float4 index (t*4+q)*64 + lane.
    Operands: %0-%15 x tiles (pinned), %16 slot base (s64), %17 lane*16 (v)."""
    L = []
    e = L.append
    e("s_nop 15")
    e("s_nop 15")
    e("s_mov_b64 s[24:25], %16")
    for t in range(16):
        for q in range(4):
            e(f"global_store_dwordx4 %17, a[{16 * t + 4 * q}:{16 * t + 4 * q + 3}], s[24:25]" + (f" offset:{q * 1024}" if q else ""))
        e("s_add_u32 s24, s24, 0x1000")
        e("s_addc_u32 s25, s25, 0")
    # The stores source their data straight from the accumulator tiles: have them retired before anything may overwrite
    # a tile (measured: without this wait a build whose compiler-side code happened not to wait here read back wrong tiles).
    e("s_waitcnt vmcnt(0)")
    return L


def gen_viewreduce():
    """Last view: x = reduce(slot_0 .. slot_{NS-2}, x) (mean or max), one pass per parked view, nine tiles' loads
    in flight while a tile is combined.  Operands: %0-%15 x (pinned), %16 slot_0 base (s64), %17 NS-1 (s),
    %18 combine_max (s), %19 lane*16 (v), %20 1/NS (v)."""
    L = []
    e = L.append

    def loads(buf):
        for q in range(4):
            e(f"global_load_dwordx4 v[{buf + 4 * q}:{buf + 4 * q + 3}], %19, s[24:25]" + (f" offset:{q * 1024}" if q else ""))
        e("s_add_u32 s24, s24, 0x1000")
        e("s_addc_u32 s25, s25, 0")

    def combine_pass(op):
        # The parked streams sit in L2 / Infinity Cache: nine tiles (36 x 1 KiB loads per wave) are kept in flight ahead of
        # the tile being combined.
        NB, D = 10, 9
        buf = lambda t: 96 + 16 * (t % NB)
        for t in range(D):
            loads(buf(t))
        for t in range(16):
            if t + D < 16:
                loads(buf(t + D))
            e(f"s_waitcnt vmcnt({4 * min(D, 15 - t)})")
            b = buf(t)
            for i in range(16):
                tmp = 68 + (i & 3)
                e(f"v_accvgpr_read_b32 v{tmp}, a{16 * t + i}")
                e(f"{op} v{tmp}, v{tmp}, v{b + i}")
                e(f"v_accvgpr_write_b32 a{16 * t + i}, v{tmp}")

    e("s_nop 15")
    e("s_nop 15")
    e("s_mov_b64 s[26:27], %16")
    e("s_mov_b32 s34, %17")
    e("s_cmp_lg_u32 %18, 0")
    e("s_cbranch_scc1 5f")
    e("1:")                                                   # ---- sum over parked views, then scale
    e("s_mov_b64 s[24:25], s[26:27]")
    combine_pass("v_add_f32")
    e("s_add_u32 s26, s26, 0x10000")
    e("s_addc_u32 s27, s27, 0")
    e("s_sub_u32 s34, s34, 1")
    e("s_cmp_lg_u32 s34, 0")
    e("s_cbranch_scc1 1b")
    for t in range(16):
        for i in range(16):
            tmp = 68 + (i & 3)
            e(f"v_accvgpr_read_b32 v{tmp}, a{16 * t + i}")
            e(f"v_mul_f32 v{tmp}, v{tmp}, %20")
            e(f"v_accvgpr_write_b32 a{16 * t + i}, v{tmp}")
    e("s_branch 6f")
    e("5:")                                                   # ---- max over parked views
    e("s_mov_b64 s[24:25], s[26:27]")
    combine_pass("v_max_f32")
    e("s_add_u32 s26, s26, 0x10000")
    e("s_addc_u32 s27, s27, 0")
    e("s_sub_u32 s34, s34, 1")
    e("s_cmp_lg_u32 s34, 0")
    e("s_cbranch_scc1 5b")
    e("6:")
    e("s_nop 7")
    return L


def main():
    out = os.environ.get("PNR_ASM_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pixel_nerf_multiscale_amd", "csrc", "resblock_asm.inc")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_resblock_asm.py — do not edit.  See that file for the register contract.\n")
        for dt, name, fn in (("bf16", "PNR_RESBLOCK_ASM_BF16", gen), ("f16", "PNR_RESBLOCK_ASM_F16", gen),
                             ("bf16", "PNR_XSTAGES_ASM_BF16", gen_xstages), ("f16", "PNR_XSTAGES_ASM_F16", gen_xstages),
                             ("bf16", "PNR_LINOUT_ASM_BF16", gen_linout), ("f16", "PNR_LINOUT_ASM_F16", gen_linout),
                             (None, "PNR_VIEWSPILL_ASM", gen_viewspill), (None, "PNR_VIEWREDUCE_ASM", gen_viewreduce)):
            lines = fn(dt) if dt else fn()
            if "nobarrier" in DIAG:
                lines = [l for l in lines if l != "s_barrier"]
            if "nowait" in DIAG:
                lines = [("s_nop 0" if l == "s_waitcnt vmcnt(4)" else l) for l in lines]
            if "nodma" in DIAG:
                lines = [l for l in lines if not l.startswith("global_load_lds")]
                lines = [("s_nop 0" if l == "s_waitcnt vmcnt(4)" else l) for l in lines]
            f.write(f"#define {name} \\\n")
            for l in lines:
                f.write(f'    "{l}\\n\\t" \\\n')
            f.write('    ""\n\n')
        clob = ["memory", "scc", "vcc"] + [f"v{i}" for i in list(range(10, 20)) + list(range(40, 56)) + list(range(60, 88)) + list(range(96, 256))] \
            + [f"s{i}" for i in list(range(20, 32)) + list(range(33, 44))]
        f.write("#define PNR_RESBLOCK_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + "\n")
    # audit: every physical v/s register the text names must be declared clobbered (operands are %N references)
    import re
    body = open(out).read()
    body = body[:body.index("#define PNR_RESBLOCK_CLOBBERS")]
    used = {m.group(1) + m.group(2) for m in re.finditer(r"\b([vs])(\d+)\b", body)}
    for m in re.finditer(r"\b([vs])\[(\d+):(\d+)\]", body):
        used |= {m.group(1) + str(i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    stray = sorted(used - set(clob))
    assert not stray, f"registers used but not clobbered: {stray}"
    print("wrote", out)


if __name__ == "__main__":
    main()
