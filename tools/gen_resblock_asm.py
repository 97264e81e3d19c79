#!/usr/bin/env python3
"""
Generates pixel_nerf_multiscale_amd/csrc/resblock_asm.inc: the hand-scheduled gfx950 asm blocks of the fused point
kernel (csrc/point_mfma.hip) — every MFMA of the network is issued from here:

  PNR_XSTAGES_ASM_{BF16,F16}    n k-steps 'x += W . B' with B = a k-step image in the wave's LDS buffer (LIN_IN, lin_z)
  PNR_RESBLOCK_ASM_{BF16,F16}   a RUN of blocks, each [lin_z k-steps] + bias k-step + x += fc_1(relu(fc_0(relu(x))))  (resnetfc.py:53-62)
  PNR_LINOUT_ASM_{BF16,F16}     last fc_1 bias + lin_out(relu(x))                                           (resnetfc.py:235)
  PNR_VIEWSPILL_ASM / PNR_VIEWREDUCE_ASM   park / reduce the per-view residual streams                       (util.py:466-476)

MFMA shape: v_mfma_f32_16x16x32_{bf16,f16}, TWO per 1-KiB weight fragment (the wave's 32 points = two 16-column
groups).  Same LDS bytes, same MFMA cycles per FLOP as one 32x32x16 per fragment, but the chip holds a ~15 % higher
clock on this shape under the kernel's load (tools/dev/ubench/shape_ubench.hip: 1.92-1.96 vs 1.66-1.69 GHz, 11 %
less wall time per chunk; MI355X_MICROARCH.md 'DVFS give-back' item 7).  The kernel is limited by board power, not by a
pipe: PNR_ASM_DIAG builds without the weight DMA run 12 % faster but only 4 % fewer cycles.

Data layout (shared with k_pack_mlp and the kernel prologue, point_mfma.hip):
  * lane l: c = l & 15 (column), g = l >> 4 (k-quarter).  The wave's points are (cg, c), cg = 0, 1.
  * X^T (512 features x 32 points, fp32) = 32 row groups x 2 column groups of 16x16 accumulators,
    X(rg, cg) = a[8 rg + 4 cg : +3]; lane (g, c) register r holds feature 16 rg + 4 g + r of point (cg, c).
    (The C++ side still sees 16 'tiles' of 16 registers, tile t = a[16t : 16t+15] = row groups 2t, 2t+1.)
  * a weight fragment = the A operand of one MFMA pair: 16 rows x 32 k, lane (row = l & 15, g) holds k = 8 g + j.
  * B operands built from accumulators (relu(x), relu(h)): element j of lane group g in a 32-wide k-step is row
    (j < 4 ? 4 g + j : 16 + 4 g + j - 4) of that k-step — the packer stores fc_0 / fc_1 / lin_out with the same permutation.
  * a stage = 16 fragments (16 KiB) of the packed stream; a k-step of an x-stage = 2 stages (row groups 0-15, 16-31).

Contract with k_point_mfma:
  * x tiles pinned: operands %0..%15 "+{a[16t:16t+15]}"; %16-%20 loader/consumer cursor in/out (st_slot, ld_idx,
    ld_slot, ld_rep, ld_wrap); then per block the inputs listed at each generator function.
  * entry: the next stage to consume is published, DMA of the two stages after it is in flight; exit: all LDS reads
    drained, accumulators readable, cursor advanced.
Statement rules (audited by main() on the generated text, see audit_statement):
  R1 entry   hipcc does NOT order its own pending loads against the registers a statement merely CLOBBERS: SIInsertWaitcnts
             skips the implicit defs of a memory-touching INLINEASM, and the hardware has no interlock — a scratch reload
             (or any VMEM / LDS / scalar load) still in flight lands AFTER the statement's own write of that register and
             wins.  (Round 2: `scratch_load_dwordx4 v[8:11]` pending on the path into the resblock statement zeroed its ring
             read bases v10 / v11, fp16 frame only; DESIGN.md 4.1, profiles/r03_entry_hazard_isa_excerpt.txt.)  So every
             statement that writes a clobbered VGPR opens with `s_waitcnt vmcnt(0)` + `s_waitcnt lgkmcnt(0)` BEFORE the first
             such write.  The LDS-DMA pieces of the previous statement are drained by it as well (0.1 %: they are two
             stages old).
  R2 exit    nothing that writes a register is pending at the end: the last LDS read is behind an lgkmcnt(0), the last
             register load behind a vmcnt(0).  LDS-DMA pieces (no register destination) may stay in flight.
  R3 M0      an LDS-DMA reads M0 when it ISSUES: rewriting M0 in the very next instruction, behind a busy VMEM queue or
             not, still lands every byte at the original destination (tools/dev/ubench/waw_ubench.hip part C,
             profiles/r03_waw_ubench.txt) — so M0 need not be held behind a piece.  The other direction is a documented
             hazard: `s_mov_b32 m0` -> LDS-DMA needs one wait state (s_nop 0).
  R4 MFMA    hipcc pads nothing across ';;#ASMEND': the last MFMA's result must be readable by whatever the compiler puts
             next (a v_accvgpr_read of a tile it parks, a VALU on an output).  An 8-pass MFMA's D wants 12 wait states
             before a non-MFMA reader, a 16-pass one 18+; every statement that issues an MFMA ends with at least
             MFMA_EXIT_STATES (24) states behind its last one (s_nop k = k + 1 states, any other instruction 1).
Registers used inside (all declared clobbered): v10-13 slot read bases, v14 address temp, v15 DMA lane offset, v16-23 B pair of the
x-stages, v40-55 / v72-87 the two chunk accumulators (VGPR-form MFMA), v44-51 next B pair (x-stages), v60-67 relu(h)
fragments, v68-71 temps, v96-127 A-fragment ring (8 x 4), v128-255 relu(x) fragments (16 k-steps x 2 column groups x 4),
s20-s31, s33-s43 (s32 is the ABI stack pointer: left alone).
"""
import os
import re

# timing experiments only (results are garbage): nobarrier, nodma, nowait, nosnap, dma1, dmaearly, dmaplain, dmaquarter, nodsread,
# nolgkm, noadvance; placement: align4; cache policy of the weight stream (results stay exact): nt0 .. nt3;
# fp16 saturation by v_pk_min instead of MODE.FP16_OVFL (round-3 form): satmin;
# hazard experiments of round 2 (kept for the record, all explained by rule R1): fullwait, ldswait, nosat, cvtnop, drainA/B/C, waitA4, sleepA, barA
DIAG = os.environ.get("PNR_ASM_DIAG", "")
# pieces q >= NT_FROM of every stage carry the non-temporal hint (nt0 = all .. nt3 = a quarter of the stream; default: none)
# cache policy of the park stores / reduce loads (experiments: parknt, reducent; results stay exact)
PARK_POLICY = " nt" if "parknt" in DIAG else ""
REDUCE_POLICY = " nt" if "reducent" in DIAG else ""
NT_FROM = next((int(m.group(1)) for m in [re.search(r"\bnt([0-3])\b", DIAG)] if m), 4)


def A(i):
    return f"v[{96 + 4 * i}:{99 + 4 * i}]"


def XB(ks, cg):
    b = 128 + (ks * 2 + cg) * 4
    return f"v[{b}:{b + 3}]"


def X(rg, cg):
    b = 8 * rg + 4 * cg
    return f"a[{b}:{b + 3}]"


def HB(cg):
    return f"v[{60 + 4 * cg}:{63 + 4 * cg}]"


def BX(cg):
    return f"v[{16 + 4 * cg}:{19 + 4 * cg}]"


class Emit:
    """Instruction list + a scoreboard of the outstanding LDS reads (they return in order), so every s_waitcnt
    lgkmcnt(N) is computed from the program order instead of counted by hand."""

    def __init__(self, dt):
        self.dt = dt
        self.mfma = {"bf16": "v_mfma_f32_16x16x32_bf16", "f16": "v_mfma_f32_16x16x32_f16"}[dt]
        self.cvt = {"bf16": "v_cvt_pk_bf16_f32", "f16": "v_cvt_pk_f16_f32"}[dt]
        self.L = []
        self.reads = []          # tags of outstanding LDS reads, oldest first
        self.loop_state = []

    def e(self, line):
        self.L.append(line)

    def ds_read(self, tag, text):
        if "nodsread" in DIAG and tag.startswith("A"):      # timing only: no weight-fragment reads
            return
        self.L.append(text)
        self.reads.append(tag)
        assert len(self.reads) <= 15, ("more than 15 LDS reads in flight", self.reads)

    def need(self, tags):
        idx = max([i for i, t in enumerate(self.reads) if t in tags], default=-1)
        if idx < 0:
            return
        if "nolgkm" not in DIAG:                            # timing only: no counted LDS waits
            self.e(f"s_waitcnt lgkmcnt({len(self.reads) - 1 - idx})")
        self.reads = self.reads[idx + 1:]

    def drain(self):
        self.e("s_waitcnt lgkmcnt(0)")
        self.reads = []

    def loop_begin(self, label):
        self.loop_state.append(list(self.reads))
        self.e(f"{label}:")

    def loop_end(self):
        st = self.loop_state.pop()
        assert st == self.reads, ("LDS scoreboard differs between loop entry and back edge", st, self.reads)

    def relu_pack(self, dst, a0, a1, tmp):
        if "nosnap" in DIAG:
            return
        self.e(f"v_accvgpr_read_b32 v{tmp}, a{a0}")
        self.e(f"v_accvgpr_read_b32 v{tmp + 1}, a{a1}")
        self.relu_pack_v(dst, tmp, tmp + 1)

    def relu_pack_v(self, dst, v0, v1):
        self.e(f"{self.cvt} v{dst}, v{v0}, v{v1}")
        if "cvtnop" in DIAG:
            self.e("s_nop 1")
        self.e(f"v_pk_max_i16 v{dst}, v{dst}, 0")
        # fp16: no saturating v_pk_min here — the kernel runs with MODE.FP16_OVFL set (k_point_mfma's entry), under which the
        # conversion itself clamps an overflowing finite value to +-65504 (measured on gfx950 for v_cvt_pk_f16_f32:
        # tools/dev/ubench/ovfl_ubench.hip, profiles/r04_fp16_ovfl.txt; -2.3 % kernel time on the headline frame)
        if self.dt == "f16" and "satmin" in DIAG:                # the round-3 form, for A/B runs
            self.e(f"v_pk_min_i16 v{dst}, v{dst}, s38")

    def snapshot_ks(self, ks):
        """relu(x) of row groups 2ks, 2ks+1 -> the two B fragments of k-step ks (16 accumulator reads, 8 packs)."""
        for cg in range(2):
            for half in range(2):            # half 0: row group 2ks (elements 0-3), half 1: row group 2ks+1 (elements 4-7)
                base = 8 * (2 * ks + half) + 4 * cg
                for p in range(2):
                    self.relu_pack(128 + (ks * 2 + cg) * 4 + 2 * half + p, base + 2 * p, base + 2 * p + 1, 68 + 2 * p)

    def loader_advance(self):
        e = self.e
        if "noadvance" in DIAG:                             # timing only: the loader re-reads one stage
            return
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

    def stage(self, mm, rd_cur, rd_nxt, m0_sreg, hooks=None, vm_alt=None):
        """One stage: 16 fragments, fragment f in A(f & 7).  mm(f) -> the two MFMA lines of fragment f.  rd_cur / rd_nxt:
        VGPRs holding this lane's read base of this stage's / the next stage's ring slot; m0_sreg: SGPR with the LDS-DMA
        destination of the stage being loaded meanwhile.  The refill of A(f & 7) (fragment f + 8) is issued one MFMA late
        (after the first MFMA of fragment f + 1) and the 4 DMA pieces after the second MFMA of fragments 1, 5, 9, 13, so
        no MFMA gap carries more than one of them.  hooks: {f: callable} extra work after fragment f's second MFMA.
        vm_alt = (flag test, N): where the run-time flag is set, extra LDS-DMA pieces (younger than the pieces this stage must
        see landed) are in flight, and the counted wait is vmcnt(N) instead of vmcnt(4) — loads retire in order."""
        e = self.e
        if vm_alt and "nodma" not in DIAG and "nowait" not in DIAG:
            test, n = vm_alt
            self.uid = getattr(self, "uid", 100) + 1
            e(test)
            e(f"s_cbranch_scc1 {self.uid}f")
            e("s_waitcnt vmcnt(4)")
            e(f"s_branch {self.uid + 1}f")
            e(f"{self.uid}:")
            e(f"s_waitcnt vmcnt({n})")
            self.uid += 1
            e(f"{self.uid}:")
        else:
            e("s_waitcnt vmcnt(4)")
        e("s_barrier")
        pend = None
        if "dmaearly" in DIAG:
            e(f"s_mov_b32 m0, {m0_sreg}")
            e("s_nop 0")
            for q in range(4):
                e("global_load_lds_dwordx4 v15, s[24:25]" + (f" offset:{q * 1024}" if q else ""))
            self.loader_advance()
        for f in range(16):
            if f % 4 == 0:
                self.need([f"A{(f + i) & 7}" for i in range(4)])
            m1, m2 = mm(f)
            e(m1)
            if pend:
                self.ds_read(*pend)
            e(m2)
            if "dmaearly" in DIAG:
                pass
            elif f % 4 == 1:
                q = f >> 2
                if q == 0:
                    e(f"s_mov_b32 m0, {m0_sreg}")
                    e("s_nop 0")
                if "dma1" not in DIAG or q == 0:
                    e("global_load_lds_dwordx4 v15, s[24:25]" + (f" offset:{q * 1024}" if q else "") + (" nt" if q >= NT_FROM else ""))
                if q == 3:
                    self.loader_advance()
            base = rd_cur if f < 8 else rd_nxt
            pend = (f"A{f & 7}", f"ds_read_b128 {A(f & 7)}, {base} offset:{((f + 8) & 15) * 1024}")
            if hooks and f in hooks:
                hooks[f]()
        self.ds_read(*pend)

    def read_first_frags(self, base):
        for i in range(8):
            self.ds_read(f"A{i}", f"ds_read_b128 {A(i)}, {base} offset:{i * 1024}")


def entry_guard(e):
    """Rule R1: before the statement writes any register it only declares clobbered.  lgkmcnt(0) also covers hipcc's scalar
    (kernel-argument) loads and LDS accesses: SMEM returns out of order and every later wait is a COUNTED lgkmcnt(N)."""
    e("s_waitcnt vmcnt(0)")
    e("s_waitcnt lgkmcnt(0)")


def setup_cursor(E, cfg, stream):
    e = E.e
    entry_guard(e)
    if "align4" in DIAG:                                     # placement experiment: shift the statement's text by 4 bytes
        e("s_nop 0")
    e("s_nop 15")
    e("s_nop 15")                                            # accumulator writes of the caller's last MFMAs retired
    e("s_mov_b32 s39, m0")                                   # hipcc may keep a value in M0 across the statement
    e("s_mov_b32 s20, %16")
    e("s_mov_b32 s21, %17")
    e("s_mov_b32 s22, %18")
    e("s_mov_b32 s23, %19")
    e("s_mov_b32 s29, %20")
    e(f"s_and_b32 s26, {cfg}, 0xfff")
    e(f"s_bfe_u32 s27, {cfg}, 0xc000c")                      # offset 12, width 12
    e(f"s_bfe_u32 s28, {cfg}, 0x80018")                      # offset 24, width 8
    e(f"s_mov_b64 s[36:37], {stream}")
    e("s_lshl_b32 s35, s21, 14")
    e("s_add_u32 s24, s36, s35")
    e("s_addc_u32 s25, s37, 0")
    if E.dt == "f16":
        e("s_mov_b32 s38, 0x7bff7bff")


def exit_cursor(E, advance):
    """advance = stages consumed by the fixed part since s20/s22 were last normalised."""
    e = E.e
    e("s_mov_b32 m0, s39")
    e(f"s_add_u32 s20, s20, {advance}")
    e("s_and_b32 %16, s20, 3")
    e("s_mov_b32 %17, s21")
    e(f"s_add_u32 s22, s22, {advance}")
    e("s_and_b32 %18, s22, 3")
    e("s_mov_b32 %19, s23")
    e("s_mov_b32 %20, s29")


def fixed_bases(E, ring_lane, ring_wave):
    """Stage i of the fixed part reads ring slot (s20 + i) & 3 through v[10 + (i & 3)]; while it runs, the stage three
    ahead is loaded into slot (s22 + i) & 3, DMA destination s[40 + (i & 3)]."""
    e = E.e
    for k in range(4):
        e(f"s_add_u32 s35, s20, {k}")
        e("s_and_b32 s35, s35, 3")
        e("s_lshl_b32 s35, s35, 14")
        e(f"v_add_u32 v{10 + k}, s35, {ring_lane}")
        e(f"s_add_u32 s35, s22, {k}")
        e("s_and_b32 s35, s35, 3")
        e("s_lshl_b32 s35, s35, 14")
        e(f"s_add_u32 s{40 + k}, s35, {ring_wave}")


def fixed_stage(E, i, mm, hooks=None, vm_alt=None):
    E.stage(mm, f"v{10 + (i & 3)}", f"v{10 + ((i + 1) & 3)}", f"s{40 + (i & 3)}", hooks, vm_alt)


def xstages_core(E, baddr, cfg2, ring_lane, ring_wave, dma_off, bias_dword, allow_init=False):
    """cfg2 & 0xff k-steps 'x += W . B' with B = the k-step image at LDS address `baddr` ([k-step][column group][lane] x 16 B,
    2 KiB per k-step), then (cfg2 bit 8) one bias k-step with B = (bias_dword, 0, 0, 0).  A k-step = 2 stages (row groups
    0-15, 16-31).  Enters with the cursor in s20-s29 / s[24:25] / s[36:37]; leaves s20 (st_slot) and s22 (ld_slot) advanced
    and masked, all LDS reads drained except the first 8 fragments of the next stage (in A, tags A0-A7), and the B pair of
    the k-step BEHIND the last one it ran in v16-23 (its read-ahead: the folded head k-step of a projected block uses it).

    Round 3: the k-steps run in PAIRS — four stages, i.e. once round the 4-slot ring, so the slot bases of fixed_bases()
    hold for the whole loop (round 2 recomputed them after every stage: 11 scalar / vector instructions), and the two B pairs
    alternate between v16-23 and v44-51 (round 2 copied the read-ahead pair down behind every k-step: s_nop 7 + 8 v_mov with
    the MFMA pipe drained).  An odd k-step, and LIN_IN's C = 0 k-step, run singly and re-base afterwards."""
    e = E.e

    def mm_half(half, init=False, b0=16):
        def mm(f):
            rg = 16 * half + f
            return [f"{E.mfma} {X(rg, cg)}, {A(f & 7)}, v[{b0 + 4 * cg}:{b0 + 4 * cg + 3}], " + ("0" if init else X(rg, cg))
                    for cg in range(2)]
        return mm

    def advance2():
        e("s_add_u32 s20, s20, 2")
        e("s_and_b32 s20, s20, 3")
        e("s_add_u32 s22, s22, 2")
        e("s_and_b32 s22, s22, 3")

    def single_kstep(init):
        """One k-step on B = v16-23 at ring offsets 0, 1; the read-ahead pair comes back into v16-23; bases re-derived."""
        E.ds_read("N0", "ds_read_b128 v[44:47], v14")        # next k-step's B pair (the one past the last is never used)
        E.ds_read("N1", "ds_read_b128 v[48:51], v14 offset:1024")
        e("v_add_u32 v14, 0x800, v14")
        fixed_stage(E, 0, mm_half(0, init))
        fixed_stage(E, 1, mm_half(1, init))
        E.need(["N0", "N1"])
        e("s_nop 7")                                         # the last MFMAs have read the old B
        for i in range(8):
            e(f"v_mov_b32 v{16 + i}, v{44 + i}")
        advance2()
        fixed_bases(E, ring_lane, ring_wave)

    def pair_of_ksteps():
        """Two k-steps = stages 0..3 of the ring: B(k) in v16-23, B(k+1) read ahead into v44-51 and used in place, B(k+2) read
        ahead into v16-23 behind the first fragment of stage 2 (the last MFMA reading the old pair issued a fragment earlier)."""
        E.ds_read("N0", "ds_read_b128 v[44:47], v14")
        E.ds_read("N1", "ds_read_b128 v[48:51], v14 offset:1024")
        fixed_stage(E, 0, mm_half(0))
        fixed_stage(E, 1, mm_half(1))
        E.need(["N0", "N1"])

        def ahead():
            E.ds_read("M0", "ds_read_b128 v[16:19], v14 offset:2048")
            E.ds_read("M1", "ds_read_b128 v[20:23], v14 offset:3072")
            e("v_add_u32 v14, 0x1000, v14")
        fixed_stage(E, 2, mm_half(0, b0=44), {0: ahead})
        fixed_stage(E, 3, mm_half(1, b0=44))
        E.need(["M0", "M1"])

    e(f"s_and_b32 s34, {cfg2}, 0xff")                        # k-steps with B from the LDS image
    fixed_bases(E, ring_lane, ring_wave)
    e(f"v_mov_b32 v14, {baddr}")
    e(f"v_mov_b32 v15, {dma_off}")
    E.ds_read("B0", "ds_read_b128 v[16:19], v14")            # B pair of the first k-step
    E.ds_read("B1", "ds_read_b128 v[20:23], v14 offset:1024")
    e("v_add_u32 v14, 0x800, v14")
    E.read_first_frags("v10")
    E.need(["B0", "B1"])
    if allow_init:
        # cfg2 bit 9: the first k-step WRITES x (C = 0) instead of accumulating — LIN_IN starts the residual stream, so the
        # caller neither zeroes 256 accumulators nor keeps them live into this statement
        e(f"s_bitcmp1_b32 {cfg2}, 9")
        e("s_cbranch_scc0 6f")
        saved = list(E.reads)
        single_kstep(True)
        assert saved == E.reads
        e("s_sub_u32 s34, s34, 1")
        e("6:")
    e("s_cmp_lt_u32 s34, 2")
    e("s_cbranch_scc1 3f")
    E.loop_begin("1")
    pair_of_ksteps()
    e("s_sub_u32 s34, s34, 2")
    e("s_cmp_ge_u32 s34, 2")
    e("s_cbranch_scc1 1b")
    E.loop_end()
    e("3:")
    e("s_cmp_eq_u32 s34, 0")
    e("s_cbranch_scc1 7f")
    saved = list(E.reads)
    single_kstep(False)                                      # the odd k-step
    assert saved == E.reads
    e("7:")
    e(f"s_bitcmp1_b32 {cfg2}, 8")                            # bias k-step requested?
    e("s_cbranch_scc0 4f")
    e(f"v_mov_b32 v16, {bias_dword}")
    for i in range(1, 8):
        e(f"v_mov_b32 v{16 + i}, 0")
    e("v_mov_b32 v20, v16")
    e("s_nop 1")
    saved = list(E.reads)
    fixed_stage(E, 0, mm_half(0))
    fixed_stage(E, 1, mm_half(1))
    assert saved == E.reads
    advance2()
    e("4:")


def gen_xstages(dt):
    """Operands: %0-%15 x tiles, %16-%20 cursor in/out, %21 cfg = P1 | (P1+P2)<<12 | NS<<24 (the loader's view of the stream),
    %22 stream base (s64), %23 ring LDS base + wave*4096 (s), %24 ring LDS base + lane*16 (v), %25 DMA lane offset
    wave*4096 + lane*16 (v), %26 B image address + lane*16 (v), %27 bias B dword 0 (v), %28 cfg2 = k-steps | bias<<8 | init<<9
    (init: the first k-step overwrites x; the caller then passes the tiles as outputs only)."""
    E = Emit(dt)
    setup_cursor(E, "%21", "%22")
    xstages_core(E, "%26", "%28", "%24", "%23", "%25", "%27", allow_init=True)
    E.drain()
    E.e("s_nop 15")
    E.e("s_nop 15")
    E.e("s_mov_b32 m0, s39")
    E.e("s_mov_b32 %16, s20")
    E.e("s_mov_b32 %17, s21")
    E.e("s_mov_b32 %18, s22")
    E.e("s_mov_b32 %19, s23")
    E.e("s_mov_b32 %20, s29")
    return E.L


def bias_kstep_with_snapshot(E, i0, extra0=None, extra1=None, vm_alt1=None):
    """The block's head k-step (the bias k-step, or with the fold the last lin_z k-step) as stages i0, i0+1 of the fixed part
    (B = v16-23), carrying the snapshot relu(x) -> XB just in time: a row group is final once ITS fragment of this k-step has
    run (fragment f of the first stage = row group f, of the second = row group 16 + f), so k-step ks of the snapshot (row
    groups 2 ks, 2 ks + 1) is converted two fragments later — k-steps 0-4 inside the first stage, 5-10 inside the second,
    11-15 by the caller in the first fragments of the next stage (tail_snapshot_hooks), none of which reads XB before
    fragment 11.  256 conversions in ONE stage made that stage VALU-bound (1024 issue cycles against 512 of MFMA); spread
    like this no stage carries more than 192."""
    def mm_half(half):
        def mm(f):
            rg = 16 * half + f
            return [f"{E.mfma} {X(rg, cg)}, {A(f & 7)}, {BX(cg)}, {X(rg, cg)}" for cg in range(2)]
        return mm
    snap = lambda ks: (lambda: E.snapshot_ks(ks))
    h0 = {3: snap(0), 5: snap(1), 7: snap(2), 9: snap(3), 11: snap(4)}
    h1 = {0: snap(5), 1: snap(6), 2: snap(7), 4: snap(8), 6: snap(9), 8: snap(10)}
    if extra0:
        h0.update(extra0)           # (the resblock's image prefetch rides in fragment 14 of both stages)
    if extra1:
        h1.update(extra1)
    fixed_stage(E, i0, mm_half(0), h0)
    fixed_stage(E, i0 + 1, mm_half(1), h1, vm_alt1)


def tail_snapshot_hooks(E):
    """k-steps 11-15 of the snapshot (row groups 22-31, final at the end of the head k-step) in fragments 0, 2, .., 8 of the
    stage that follows it."""
    return {2 * i: (lambda ks=11 + i: E.snapshot_ks(ks)) for i in range(5)}


def gen(dt):
    """One resblock.  Inputs after the common %0-%20: %21 cfg, %22 stream (s64), %23 ring + wave*4096 (s), %24 ring +
    lane*16 (v), %25 DMA lane offset (v), %26 LDS address of fc_0.bias[block] + 16*(lane>>4) (v), %27 bias B dword 0 (v),
    %28 lin_z B image address + lane*16 (v), %29 = lin_z k-steps of a prefixed block that run as x-stages | number of
    consecutive blocks with the lin_z prefix << 16 | number of blocks without it after them << 20 | bias folded into the last
    lin_z k-step << 24 | image prefetch << 25 (%26 names the FIRST block's bias rows; the two stages at the head of the fixed part
    are the bias k-step, or with the fold the last lin_z k-step).
    Image prefetch (bit 25; blocks with several lin_z parts — multi-scale — run one block per statement): the NEXT block's first
    gathered lin_z image (16 KiB, kept in the wave's workspace by the view's first block) is fetched back into the wave's B-image
    buffer by 16 LDS-DMA pieces riding in fragment 14 of the two head stages — the buffer is free from there on (the prefix's
    reads are drained, the head k-step's B pair is in registers) and the next reader is the x-stages statement after this one,
    behind its entry wait.  %30 = the image's address in the workspace (s64), %31 = lane*16 (v), %32 = LDS address of the
    buffer (s).  The pieces are younger than weight pieces the next three stages wait for: those stages' counted waits are
    vmcnt(12 / 20 / 12) where the bit is set.
    Stage order (k_pack_mlp follows it): [lin_z k-steps x 2] | bias x 2 | F(0) | F(1) G(0) | ... | F(15) G(14) | G(15), with
    F(c) = the 2 fc_0 stages of chunk c (chunk accumulator += W0[32c..32c+31, :] . relu(x)) and G(c) = its 2 fc_1 stages
    (x += W1[:, 32c..32c+31] . relu(h_c)).  Two chunk accumulators alternate (v40-55 / v72-87): while F(c+1) runs on one, the
    other — finished 64 MFMAs earlier — is converted to relu(h_c) in the MFMA gaps and reloaded with the fc_0.bias rows of
    chunk c+2."""
    E = Emit(dt)
    e = E.e
    setup_cursor(E, "%21", "%22")
    if "drainA" in DIAG:
        e("s_waitcnt vmcnt(0)")
    if "waitA4" in DIAG:
        e("s_waitcnt vmcnt(4)")
    if "sleepA" in DIAG:
        e("s_sleep 4")
    if "barA" in DIAG:
        e("s_waitcnt vmcnt(4)")
        e("s_barrier")
    # Block loop: %29 = lin_z k-steps | blocks WITH the lin_z prefix << 16 | blocks without << 20.  Consecutive blocks of a
    # tile run inside ONE statement when nothing has to happen between them (one lin_z part per block: the B image in LDS
    # serves them all) — a statement boundary costs ~700 cycles (drains, cursor set-up, the first fragments' LDS latency).
    e("s_bfe_u32 s30, %29, 0x40010")                         # offset 16, width 4
    e("s_bfe_u32 s31, %29, 0x40014")                         # offset 20, width 4
    e("s_mov_b32 s33, 0")                                    # byte offset of the block's fc_0.bias rows from %26
    E.loop_begin("8")
    e("s_cmp_eq_u32 s30, 0")
    e("s_cbranch_scc1 5f")
    xstages_core(E, "%28", "%29", "%24", "%23", "%25", "%27")
    E.drain()                                                # both paths reach 9: with nothing in flight
    e("s_sub_u32 s30, s30, 1")
    # %29 bit 24 (projected streams): the block's bias is folded into the LAST lin_z k-step (its columns W_z.Lat + bias: the
    # tap weights sum to 1), which then takes the place of the bias k-step below — the x-stages above ran one k-step fewer
    # and left that k-step's B pair in v16-23 (their read-ahead), so the bias pattern must not overwrite it.
    e("s_bitcmp1_b32 %29, 24")
    e("s_cbranch_scc1 9f")

    def bias_b_operand():
        e("v_mov_b32 v16, %27")
        for i in range(1, 8):
            e(f"v_mov_b32 v{16 + i}, 0")
        e("v_mov_b32 v20, v16")
    bias_b_operand()
    e("s_branch 9f")
    e("5:")
    e("s_sub_u32 s31, s31, 1")
    bias_b_operand()
    e("9:")
    if "drainB" in DIAG:
        e("s_waitcnt vmcnt(0)")
    fixed_bases(E, "%24", "%23")
    e("v_add_u32 v14, s33, %26")
    e("v_mov_b32 v15, %25")
    ACC = (40, 72)

    def hbias(acc, tagp):
        """fc_0.bias rows of the next chunk into a chunk accumulator: register r of (rgl, cg) = bias[32 c + 16 rgl + 4 g + r]."""
        for rgl in range(2):
            for cg in range(2):
                b = acc + 8 * rgl + 4 * cg
                E.ds_read(f"{tagp}{rgl}{cg}", f"ds_read_b128 v[{b}:{b + 3}], v14 offset:{64 * rgl}")
        e("v_add_u32 v14, 0x80, v14")

    hbias(ACC[0], "h")                                       # chunk 0
    E.read_first_frags("v10")
    # ---------------------------------------------------------------- bias k-step + snapshot
    PF = "s_bitcmp1_b32 %29, 25"

    def prefetch_half(half):
        """8 pieces = 8 KiB of the next block's image: source %30 + 8192 half + 4096 j, destination %32 + the same."""
        def hook():
            E.uid = getattr(E, "uid", 100) + 1
            lab = E.uid
            e(PF)
            e(f"s_cbranch_scc0 {lab}f")
            for j in range(2):
                off = 8192 * half + 4096 * j
                e("s_mov_b64 s[34:35], %30")                 # s34 / s35: free here (loop counter / loader temp of other phases)
                if off:
                    e(f"s_add_u32 s34, s34, {off}")
                    e("s_addc_u32 s35, s35, 0")
                e(f"s_add_u32 m0, %32, {off}")
                e("s_nop 0")
                for q in range(4):
                    e("global_load_lds_dwordx4 %31, s[34:35]" + (f" offset:{q * 1024}" if q else "") + " sc1")
            e(f"{lab}:")
        return hook
    if "noprefetch" in DIAG:
        bias_kstep_with_snapshot(E, 0)
        alt = [None, None]
    else:
        bias_kstep_with_snapshot(E, 0, {14: prefetch_half(0)}, {14: prefetch_half(1)}, (PF, 12))
        alt = [(PF, 20), (PF, 12)]

    def mm_fc0(acc, half):
        def mm(f):
            rgl, ks = f & 1, 8 * half + (f >> 1)
            return [f"{E.mfma} v[{acc + 8 * rgl + 4 * cg}:{acc + 8 * rgl + 4 * cg + 3}], {A(f & 7)}, {XB(ks, cg)}, "
                    f"v[{acc + 8 * rgl + 4 * cg}:{acc + 8 * rgl + 4 * cg + 3}]" for cg in range(2)]
        return mm

    def mm_fc1(half):
        def mm(f):
            rg = 16 * half + f
            return [f"{E.mfma} {X(rg, cg)}, {A(f & 7)}, {HB(cg)}, {X(rg, cg)}" for cg in range(2)]
        return mm

    def convert_hooks(acc, reload):
        """relu(h) of the finished accumulator -> v60-67 (hb[cg] dword 2 rgl + p), one register per fragment (fragments
        2..9); then (fragment 15) the next fc_0.bias rows into it."""
        hooks = {}
        for i in range(8):
            cg, rgl, p = i >> 2, (i >> 1) & 1, i & 1
            src = acc + 8 * rgl + 4 * cg + 2 * p
            hooks[2 + i] = (lambda d=60 + 4 * cg + 2 * rgl + p, s=src: E.relu_pack_v(d, s, s + 1))
        if reload:
            hooks[15] = lambda: hbias(acc, "h")
        return hooks

    def F(acc, i, hooks0=None, need_bias=True, vm_alts=(None, None)):
        if need_bias:
            E.need([f"h{rgl}{cg}" for rgl in range(2) for cg in range(2)])
        fixed_stage(E, i, mm_fc0(acc, 0), hooks0, vm_alts[0])
        fixed_stage(E, i + 1, mm_fc0(acc, 1), None, vm_alts[1])

    def G(i):
        fixed_stage(E, i, mm_fc1(0))
        fixed_stage(E, i + 1, mm_fc1(1))

    # chunk 1's fc_0.bias rows into the second accumulator, then F(0) with the last five k-steps of the snapshot in its first stage
    hbias(ACC[1], "g")
    E.need([f"h{rgl}{cg}" for rgl in range(2) for cg in range(2)])
    F(ACC[0], 2, tail_snapshot_hooks(E), need_bias=False, vm_alts=alt)
    # the loop head: outstanding = [g-bias reads?]  make the state explicit: the g reads are waited for here
    E.need([f"g{rgl}{cg}" for rgl in range(2) for cg in range(2)])
    if "drainC" in DIAG:
        e("s_waitcnt vmcnt(0)")
    e("s_mov_b32 s34, 7")
    E.loop_begin("1")
    # c even: F(c+1) on v72.., h_c from v40.. (its conversion + the preload of chunk c+2 ride in F's first stage)
    F(ACC[1], 4, convert_hooks(ACC[0], True), need_bias=False)
    G(6)
    E.need([f"h{rgl}{cg}" for rgl in range(2) for cg in range(2)])
    # c odd: F(c+1) on v40.., h_c from v72..
    F(ACC[0], 8, convert_hooks(ACC[1], True), need_bias=False)
    G(10)
    E.need([f"h{rgl}{cg}" for rgl in range(2) for cg in range(2)])
    e("s_sub_u32 s34, s34, 1")
    e("s_cmp_lg_u32 s34, 0")
    e("s_cbranch_scc1 1b")
    E.loop_end()
    # c = 14: F(15) on v72.., no chunk 16 to preload; G(14); then G(15) from v72..
    F(ACC[1], 60, convert_hooks(ACC[0], False), need_bias=False)
    G(62)
    # G(15): its accumulator's last MFMA is 64 MFMAs back — convert, then the two fc_1 stages
    for i in range(8):
        cg, rgl, p = i >> 2, (i >> 1) & 1, i & 1
        src = ACC[1] + 8 * rgl + 4 * cg + 2 * p
        E.relu_pack_v(60 + 4 * cg + 2 * rgl + p, src, src + 1)
    G(64)
    # ---------------------------------------------------------------- block end: 2 bias + 64 chunk stages consumed
    E.drain()
    e("s_add_u32 s20, s20, 66")
    e("s_and_b32 s20, s20, 3")
    e("s_add_u32 s22, s22, 66")
    e("s_and_b32 s22, s22, 3")
    e("s_add_u32 s33, s33, 0x800")                           # next block's bias rows (512 floats)
    e("s_nop 7")                                             # the last MFMAs of the block have read their B registers
    e("s_or_b32 s35, s30, s31")
    e("s_cmp_lg_u32 s35, 0")
    e("s_cbranch_scc1 8b")
    E.loop_end()
    e("s_nop 15")
    e("s_nop 15")
    exit_cursor(E, 0)
    return E.L


def gen_linout(dt):
    """The last fc_1.bias k-step, then lin_out(relu(x)): 1 stage of 16 fragments (k-steps 0-15, rows 0-3 of each valid), into
    v40-47 (column group cg -> v[40+4cg : 43+4cg], rows 0-3 = registers 0-3 of lanes 0-15).  The 8 results are written to the
    wave's LDS buffer at %26 ([cg][lane] x 16 B) for the caller (lin_out.bias + sigmoid / relu happen there).
    Operands: %0-%15 x tiles, %16-%20 cursor, %21 cfg, %22 stream, %23 ring+wave*4096 (s), %24 ring+lane*16 (v), %25 DMA lane
    offset (v), %26 result address + lane*16 (v), %27 bias B dword 0 (v)."""
    E = Emit(dt)
    e = E.e
    setup_cursor(E, "%21", "%22")
    fixed_bases(E, "%24", "%23")
    e("v_mov_b32 v15, %25")
    e("v_mov_b32 v16, %27")
    for i in range(1, 8):
        e(f"v_mov_b32 v{16 + i}, 0")
    e("v_mov_b32 v20, v16")
    for i in range(8):
        e(f"v_mov_b32 v{40 + i}, 0")
    E.read_first_frags("v10")
    bias_kstep_with_snapshot(E, 0)

    def mm(f):
        return [f"{E.mfma} v[{40 + 4 * cg}:{43 + 4 * cg}], {A(f & 7)}, {XB(f, cg)}, v[{40 + 4 * cg}:{43 + 4 * cg}]" for cg in range(2)]
    # fragment f multiplies k-step f: k-steps 0-10 are converted by the head k-step above, 11-15 in fragments 0-8 here
    fixed_stage(E, 2, mm, tail_snapshot_hooks(E))
    E.drain()
    e("s_nop 15")
    e("s_nop 15")
    e("ds_write_b128 %26, v[40:43]")
    e("ds_write_b128 %26, v[44:47] offset:1024")
    e("s_waitcnt lgkmcnt(0)")
    exit_cursor(E, 3)
    return E.L


def gen_viewspill(dt):
    """Park this view's residual stream in the kernel's 16-bit format (bf16 / fp16): the 256 accumulator registers -> 32 KiB of
    the wave's workspace slot, dwordx4 index (2 t + q) * 64 + lane holding accumulators 16 t + 8 q .. + 7 of the lane as four
    (even, odd) pairs (the reduce reads the same mapping back).  Half the bytes of an fp32 park: with every CU parking at the
    same moment the stores queue on the fabric (13.4 k cycles per park with 256 workgroups against 4.7 k with 32,
    profiles/r03_park_contention.txt), so bytes are time.  The rounding is the one every layer's input already takes (the
    next reader of x is relu -> 16-bit for fc_0); fp16 saturates at +-65504 like the activations (MODE.FP16_OVFL).
    Operands: %0-%15 x tiles (pinned), %16 slot base (s64), %17 lane*16 (v), %18 fp32 park (s; pnr_params.park_fp32).
    Writes v68-71 / v96-223 (rule R1: entry guard).
    %18 != 0: the stream is parked as it is — fp32, 64 KiB, float4 index (4 t + q) * 64 + lane = accumulators 16 t + 4 q .. + 3,
    stored straight from the accumulator registers — for callers that want the reference's fp32 view reduction
    (util.combine_interleaved reduces fp32 activations) at the price of twice the parked bytes."""
    E = Emit(dt)
    e = E.e
    entry_guard(e)
    e("s_nop 15")
    e("s_nop 15")                                            # MFMA write -> accumulator read
    e("s_mov_b64 s[24:25], %16")
    e("s_cmp_lg_u32 %18, 0")
    e("s_cbranch_scc0 7f")
    for t in range(16):
        for q in range(4):
            e(f"global_store_dwordx4 %17, a[{16 * t + 4 * q}:{16 * t + 4 * q + 3}], s[24:25]" + (f" offset:{q * 1024}" if q else "") + PARK_POLICY)
        e("s_add_u32 s24, s24, 0x1000")
        e("s_addc_u32 s25, s25, 0")
    e("s_branch 8f")
    e("7:")
    if dt == "f16":
        e("s_mov_b32 s38, 0x7bff7bff")
        e("s_mov_b32 s35, 0xfbfffbff")
    for t in range(16):
        for q in range(2):
            buf = 96 + 8 * t + 4 * q
            for i in range(4):
                a0 = 16 * t + 8 * q + 2 * i
                e(f"v_accvgpr_read_b32 v{68 + 2 * (i & 1)}, a{a0}")
                e(f"v_accvgpr_read_b32 v{69 + 2 * (i & 1)}, a{a0 + 1}")
                e(f"{E.cvt} v{buf + i}, v{68 + 2 * (i & 1)}, v{69 + 2 * (i & 1)}")
                if dt == "f16" and "satmin" in DIAG:             # round-3 form; MODE.FP16_OVFL clamps in the conversion now
                    e(f"v_pk_min_f16 v{buf + i}, v{buf + i}, s38")
                    e(f"v_pk_max_f16 v{buf + i}, v{buf + i}, s35")
            e(f"global_store_dwordx4 %17, v[{buf}:{buf + 3}], s[24:25]" + (f" offset:{q * 1024}" if q else "") + PARK_POLICY)
        e("s_add_u32 s24, s24, 0x800")
        e("s_addc_u32 s25, s25, 0")
    # No vmcnt wait: the stores read their own buffer registers (written once each), the tiles are free for the next view's
    # LIN_IN at once, and the slot is not read before the reduce — whose entry guard drains the stores.
    e("8:")
    e("s_nop 1")                                             # (fp32 park: a store has read its data two wait states after issue)
    return E.L


def gen_viewreduce(dt):
    """Last view: x = reduce(slot_0 .. slot_{NS-2}, x) (mean or max) from the 16-bit parks, one pass per parked view, all 16
    tiles' loads (2 x dwordx4 each) in flight at once.  Operands: %0-%15 x (pinned), %16 slot_0 base (s64), %17 NS-1 (s),
    %18 combine_max (s), %19 lane*16 (v), %20 1/NS (v), %21 fp32 park (s): the parked streams are fp32 (gen_viewspill)."""
    E = Emit(dt)
    e = E.e

    def combine_pass32(op, scale=False):
        """fp32 parks: a ring of ten 16-register buffers, nine tiles' loads (4 x dwordx4 each) in flight while one is combined."""
        NB, D = 10, 9
        buf = lambda t: 96 + 16 * (t % NB)

        def loads(b):
            for q in range(4):
                e(f"global_load_dwordx4 v[{b + 4 * q}:{b + 4 * q + 3}], %19, s[24:25]" + (f" offset:{q * 1024}" if q else "") + REDUCE_POLICY)
            e("s_add_u32 s24, s24, 0x1000")
            e("s_addc_u32 s25, s25, 0")
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
                if scale:
                    e(f"v_mul_f32 v{tmp}, v{tmp}, %20")
                e(f"v_accvgpr_write_b32 a{16 * t + i}, v{tmp}")

    def combine_pass(op, scale=False):
        for t in range(16):
            for q in range(2):
                buf = 96 + 8 * t + 4 * q
                e(f"global_load_dwordx4 v[{buf}:{buf + 3}], %19, s[24:25]" + (f" offset:{q * 1024}" if q else "") + REDUCE_POLICY)
            e("s_add_u32 s24, s24, 0x800")
            e("s_addc_u32 s25, s25, 0")
        for t in range(16):
            e(f"s_waitcnt vmcnt({2 * (15 - t)})")
            for j in range(8):                               # dword j of the tile: accumulators 16 t + 2 j, + 1
                src = 96 + 8 * t + j
                for half in range(2):
                    tmp, acc = 68 + half, 16 * t + 2 * j + half
                    if dt == "bf16":
                        e(f"v_lshlrev_b32 v{70 + half}, 16, v{src}" if half == 0 else f"v_and_b32 v{70 + half}, 0xffff0000, v{src}")
                    else:
                        e(f"v_cvt_f32_f16 v{70 + half}, v{src}" if half == 0 else
                          f"v_cvt_f32_f16_sdwa v{70 + half}, v{src} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
                    e(f"v_accvgpr_read_b32 v{tmp}, a{acc}")
                    e(f"{op} v{tmp}, v{tmp}, v{70 + half}")
                    if scale:
                        e(f"v_mul_f32 v{tmp}, v{tmp}, %20")
                    e(f"v_accvgpr_write_b32 a{acc}, v{tmp}")

    entry_guard(e)                                           # rule R1: the passes write v68-71 / v96-255
    e("s_nop 15")
    e("s_nop 15")
    e("s_mov_b64 s[26:27], %16")
    e("s_mov_b32 s34, %17")

    def whole_reduce(cp, base):
        """labels base+1 .. base+6; cp = the combine pass of the park format"""
        L1, L2, L5, L6 = base + 1, base + 2, base + 5, base + 6
        e("s_cmp_lg_u32 %18, 0")
        e(f"s_cbranch_scc1 {L5}f")
        e("s_cmp_eq_u32 s34, 1")                             # ---- sum over parked views; the last pass also scales by 1/NS
        e(f"s_cbranch_scc1 {L2}f")                           #      (same order of operations as sum-then-scale: identical bits)
        e(f"{L1}:")
        e("s_mov_b64 s[24:25], s[26:27]")
        cp("v_add_f32")
        e("s_add_u32 s26, s26, 0x10000")
        e("s_addc_u32 s27, s27, 0")
        e("s_sub_u32 s34, s34, 1")
        e("s_cmp_lg_u32 s34, 1")
        e(f"s_cbranch_scc1 {L1}b")
        e(f"{L2}:")
        e("s_mov_b64 s[24:25], s[26:27]")
        cp("v_add_f32", scale=True)
        e(f"s_branch {L6}f")
        e(f"{L5}:")                                           # ---- max over parked views
        e("s_mov_b64 s[24:25], s[26:27]")
        cp("v_max_f32")
        e("s_add_u32 s26, s26, 0x10000")
        e("s_addc_u32 s27, s27, 0")
        e("s_sub_u32 s34, s34, 1")
        e("s_cmp_lg_u32 s34, 0")
        e(f"s_cbranch_scc1 {L5}b")
        e(f"{L6}:")
    e("s_cmp_lg_u32 %21, 0")
    e("s_cbranch_scc1 30f")
    whole_reduce(combine_pass, 10)
    e("s_branch 40f")
    e("30:")
    whole_reduce(combine_pass32, 20)
    e("40:")
    e("s_nop 7")
    return E.L


M0_HOLD = 0        # instructions an LDS-DMA needs before the next write of M0: none, it reads M0 at issue (rule R3)


def _vdest(line):
    """Physical VGPRs (clobbered registers, named literally) the instruction writes; AGPRs are statement operands."""
    m = re.match(r"\s*([a-z_0-9]+)\s+(.*)", line)
    if not m:
        return []
    op, rest = m.group(1), m.group(2)
    writes = (op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_accvgpr_write")) or \
        op.startswith("ds_read") or (op.startswith("global_load_dword") and not op.startswith("global_load_lds"))
    if not writes:
        return []
    d = rest.split(",")[0].strip()
    m1 = re.fullmatch(r"v(\d+)", d)
    m2 = re.fullmatch(r"v\[(\d+):(\d+)\]", d)
    if m1:
        return [int(m1.group(1))]
    if m2:
        return list(range(int(m2.group(1)), int(m2.group(2)) + 1))
    return []                                                # %N operand or an AGPR


def audit_statement(name, lines):
    """Rules R1-R4 of the header on the final text of one statement (straight-line scan: branches only skip forward or loop
    over stage bodies that satisfy the rules themselves)."""
    seen_vm = seen_lgkm = False
    first_write = None
    for i, l in enumerate(lines):
        if l.startswith("s_waitcnt") and "vmcnt(0)" in l:
            seen_vm = True
        if l.startswith("s_waitcnt") and "lgkmcnt(0)" in l:
            seen_lgkm = True
        if _vdest(l):
            first_write = i
            break
    if first_write is not None:
        assert seen_vm and seen_lgkm, f"{name}: R1 — '{lines[first_write]}' writes a clobbered VGPR before the entry guard"
    # R2: the last register-writing LDS read / global load is followed by a full wait of its counter
    last = {"lgkm": None, "vm": None}
    for i, l in enumerate(lines):
        if l.startswith("ds_read"):
            last["lgkm"] = i
        if l.startswith("global_load_dword") and _vdest(l):
            last["vm"] = i
    if last["lgkm"] is not None:
        assert any(l.startswith("s_waitcnt") and "lgkmcnt(0)" in l for l in lines[last["lgkm"]:]), f"{name}: R2 — LDS read pending at exit"
    if last["vm"] is not None:
        assert any(l.startswith("s_waitcnt") and "vmcnt(0)" in l for l in lines[last["vm"]:]), f"{name}: R2 — register load pending at exit"
    # R3: M0 writes vs LDS-DMA
    for i, l in enumerate(lines):
        if l.startswith("global_load_lds"):
            # the nearest M0 write above it must be >= 1 wait state away
            for k in range(i - 1, -1, -1):
                if lines[k].startswith(("s_mov_b32 m0", "s_add_u32 m0")):
                    assert k <= i - 2, f"{name}: R3 — no wait state between '{lines[k]}' and the DMA"
                    break
                if lines[k].startswith("global_load_lds") or lines[k].endswith(":"):
                    break
            for k in range(i + 1, min(i + 1 + M0_HOLD, len(lines))):
                assert not lines[k].startswith("s_mov_b32 m0"), f"{name}: R3 — M0 rewritten {k - i} instruction(s) behind a DMA"


    # R4: wait states behind the last MFMA
    mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
    if mf:
        states = 0
        for l in lines[mf[-1] + 1:]:
            if l.endswith(":") or l.startswith((";", "//")):
                continue
            states += int(l.split()[1]) + 1 if l.startswith("s_nop") else 1
        assert states >= MFMA_EXIT_STATES, f"{name}: R4 — only {states} wait states behind the last MFMA"


MFMA_EXIT_STATES = 24


def audit_all(path):
    text = open(path).read()
    n = 0
    for m in re.finditer(r"#define (PNR_\w+_ASM\w*) \\\n((?:    \".*\\n\\t\" \\\n)+)", text):
        lines = re.findall(r'    "(.*)\\n\\t" \\', m.group(2))
        audit_statement(m.group(1), lines)
        n += 1
    assert n == 10, f"audited {n} statements, expected 10"


def main():
    out = os.environ.get("PNR_ASM_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pixel_nerf_multiscale_amd", "csrc", "resblock_asm.inc")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_resblock_asm.py — do not edit.  See that file for the register contract.\n")
        for dt, name, fn in (("bf16", "PNR_RESBLOCK_ASM_BF16", gen), ("f16", "PNR_RESBLOCK_ASM_F16", gen),
                             ("bf16", "PNR_XSTAGES_ASM_BF16", gen_xstages), ("f16", "PNR_XSTAGES_ASM_F16", gen_xstages),
                             ("bf16", "PNR_LINOUT_ASM_BF16", gen_linout), ("f16", "PNR_LINOUT_ASM_F16", gen_linout),
                             ("bf16", "PNR_VIEWSPILL_ASM_BF16", gen_viewspill), ("f16", "PNR_VIEWSPILL_ASM_F16", gen_viewspill),
                             ("bf16", "PNR_VIEWREDUCE_ASM_BF16", gen_viewreduce), ("f16", "PNR_VIEWREDUCE_ASM_F16", gen_viewreduce)):
            lines = fn(dt)
            if "nobarrier" in DIAG:
                lines = [l for l in lines if l != "s_barrier"]
            if "nowait" in DIAG:
                lines = [("s_nop 0" if l == "s_waitcnt vmcnt(4)" else l) for l in lines]
            if "fullwait" in DIAG or "drain" in DIAG:
                pass
            if "nosat" in DIAG:
                lines = [l for l in lines if not l.startswith("v_pk_min_i16")]
            if "fullwait" in DIAG:
                lines = [("s_waitcnt vmcnt(0)" if l == "s_waitcnt vmcnt(4)" else l) for l in lines]
            if "ldswait" in DIAG:
                lines = [("s_waitcnt lgkmcnt(0)" if l.startswith("s_waitcnt lgkmcnt(") else l) for l in lines]
            if "dmaplain" in DIAG:      # timing only: ordinary loads into (garbage) registers instead of LDS-DMA
                lines = [l.replace("global_load_lds_dwordx4 v15,", "global_load_dwordx4 v[68:71], v15,") for l in lines]
            if "dmaquarter" in DIAG:
                lines = [l.replace("global_load_lds_dwordx4", "global_load_lds_dword") for l in lines]
            if "nodma" in DIAG:
                lines = [l for l in lines if not l.startswith("global_load_lds")]
                lines = [("s_nop 0" if l == "s_waitcnt vmcnt(4)" else l) for l in lines]
            f.write(f"#define {name} \\\n")
            for l in lines:
                f.write(f'    "{l}\\n\\t" \\\n')
            f.write('    ""\n\n')
        clob = ["memory", "scc", "vcc"] + [f"v{i}" for i in list(range(10, 24)) + list(range(40, 56)) + list(range(60, 88)) + list(range(96, 256))] \
            + [f"s{i}" for i in list(range(20, 32)) + list(range(33, 44))]
        f.write("#define PNR_RESBLOCK_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + "\n")
    audit_all(out)
    # audit: every physical v/s register the text names must be declared clobbered (operands are %N references)
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
