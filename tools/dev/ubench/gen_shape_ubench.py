#!/usr/bin/env python3
"""MFMA-shape study for the fused kernel's chunk loop (MI355X_MICROARCH.md 'DVFS give-back' item 7: at equal cycles
per FLOP the chip can hold a higher clock on v_mfma_f32_16x16x32_bf16 than on 32x32x16).  Emits the SAME chunk —
4 stages of 16 one-KiB A fragments read from the LDS ring by ds_read_b128 (8-deep register ring), 4 LDS-DMA pieces
per wave and stage, one barrier per stage, a VGPR accumulator chain for the fc_0 half and AGPR tiles for the fc_1 half —
once with one 32x32x16 MFMA per fragment and once with two 16x16x32 MFMAs per fragment (two 16-column groups).
Operands are random (the harness fills the stream, the LDS and the B registers), so the clocks are comparable."""
A = lambda i: f"v[{100 + 4 * i}:{103 + 4 * i}]"                  # 8 A-fragment registers v100..v131
XB = lambda i: f"v[{140 + 4 * (i % 24)}:{143 + 4 * (i % 24)}]"   # B fragments v140..v235 (random content)
HB = lambda i: f"v[{60 + 4 * (i & 1)}:{63 + 4 * (i & 1)}]"


def stage(kind, shape):
    s = ["s_waitcnt vmcnt(4)", "s_barrier"]
    pend = None                  # 16x16x32: the refill of fragment f is issued one MFMA late (after the first MFMA of f+1),
    for f in range(16):          # so no MFMA gap carries more than one of {ds_read, DMA piece}
        if f % 4 == 0:
            s.append("s_waitcnt lgkmcnt(4)" if not (shape == 16 and pend) else "s_waitcnt lgkmcnt(4)")
        refill = f"ds_read_b128 {A(f & 7)}, v2 offset:{((f + 8) & 15) * 1024}"
        dma = []
        if f % 4 == 1:
            q = f >> 2
            if q == 0:
                dma += ["s_mov_b32 m0, s20", "s_nop 0"]
            dma.append(f"global_load_lds_dwordx4 v3, s[22:23] offset:{q * 1024}")
            if q == 3:           # next stage of the stream: walk a 6 MiB region (the real stream's size), L2-resident like it
                dma += ["s_add_u32 s26, s26, 0x4000", "s_cmp_ge_u32 s26, s31", "s_cselect_b32 s26, 0, s26",
                        "s_add_u32 s22, s28, s26", "s_addc_u32 s23, s29, 0"]
        if shape == 32:
            if kind == 0:
                s.append(f"v_mfma_f32_32x32x16_bf16 v[40:55], {A(f & 7)}, {XB(f)}, v[40:55]")
            else:
                t = (f >> 1) + 8 * (kind - 2)
                s.append(f"v_mfma_f32_32x32x16_bf16 a[{16 * t}:{16 * t + 15}], {A(f & 7)}, {HB(f)}, a[{16 * t}:{16 * t + 15}]")
            s.append(refill)
            s += dma
        else:
            mm = []
            if kind == 0:        # fragment = (row group f&1, k-step f>>1); accumulators (rg, cg) = v[40 + 8 rg + 4 cg ..]
                rg = f & 1
                for cg in range(2):
                    acc = 40 + 8 * rg + 4 * cg
                    mm.append(f"v_mfma_f32_16x16x32_bf16 v[{acc}:{acc + 3}], {A(f & 7)}, {XB(2 * (f >> 1) + cg)}, v[{acc}:{acc + 3}]")
            else:                # fragment = row group 16 (kind-2) + f; accumulators a[8 rg + 4 cg ..]
                rg = 16 * (kind - 2) + f
                for cg in range(2):
                    acc = 8 * rg + 4 * cg
                    mm.append(f"v_mfma_f32_16x16x32_bf16 a[{acc}:{acc + 3}], {A(f & 7)}, {HB(cg)}, a[{acc}:{acc + 3}]")
            s.append(mm[0])
            if pend:
                s.append(pend)
            s.append(mm[1])
            s += dma
            pend = refill
    if pend:
        s.append(pend)
    return s


def chunk(shape):
    s = stage(0, shape) + stage(0, shape)
    s += ["s_nop 15", "s_nop 7"]
    cvt = "v_cvt_pk_bf16_f32"
    for i in range(8):
        s.append(f"{cvt} v{60 + i}, v{40 + 2 * i}, v{41 + 2 * i}")
        s.append(f"v_pk_max_i16 v{60 + i}, v{60 + i}, 0")
    for q in range(4):
        s.append(f"ds_read_b128 v[{40 + 4 * q}:{43 + 4 * q}], v4 offset:{32 * q}")
    s += stage(2, shape) + stage(3, shape)
    return s


for shape in (32, 16):
    print(f"#define CHUNK_ASM_{shape} \\")
    for l in chunk(shape):
        print(f'    "{l}\\n\\t" \\')
    print('    ""')
