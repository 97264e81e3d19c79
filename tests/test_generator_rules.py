"""The hand-scheduled statements of the fused kernel are generated (tools/gen_resblock_asm.py).  CPU checks: the committed
.inc is what the generator produces today, and the generator's audit enforces the statement rules R1-R4 of its header —
R1 is the root cause of round 2's fp16 race (a pending compiler load landing in a register the statement had already written)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_resblock_asm", os.path.join(ROOT, "tools", "gen_resblock_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_inc_is_current(tmp_path, monkeypatch):
    out = tmp_path / "resblock_asm.inc"
    monkeypatch.setenv("PNR_ASM_OUT", str(out))
    monkeypatch.delenv("PNR_ASM_DIAG", raising=False)
    _gen().main()                    # runs audit_all on what it wrote
    committed = open(os.path.join(ROOT, "pixel_nerf_multiscale_amd", "csrc", "resblock_asm.inc")).read()
    assert out.read_text() == committed, "regenerate: python tools/gen_resblock_asm.py"


def test_audit_rejects_rule_violations():
    g = _gen()
    ok = ["s_waitcnt vmcnt(0)", "s_waitcnt lgkmcnt(0)", "v_mov_b32 v10, %0", "ds_read_b128 v[16:19], v10", "s_waitcnt lgkmcnt(0)"]
    g.audit_statement("ok", ok)
    with pytest.raises(AssertionError, match="R1"):      # writes a clobbered VGPR before the entry guard
        g.audit_statement("no_guard", ok[2:])
    with pytest.raises(AssertionError, match="R1"):      # vmcnt(4) is not the guard (the reload is the youngest operation)
        g.audit_statement("counted", ["s_waitcnt vmcnt(4)", "s_waitcnt lgkmcnt(0)"] + ok[2:])
    with pytest.raises(AssertionError, match="R2"):      # LDS read pending at exit
        g.audit_statement("pending", ok[:4])
    # M0 rewritten right behind a piece is fine (a piece reads M0 at issue: tools/dev/ubench/waw_ubench.hip part C)
    g.audit_statement("m0", ok + ["s_mov_b32 m0, s40", "s_nop 0", "global_load_lds_dwordx4 v15, s[24:25]", "s_mov_b32 m0, s39"])
    with pytest.raises(AssertionError, match="R3"):      # no wait state between the M0 write and the DMA
        g.audit_statement("m0b", ok + ["s_mov_b32 m0, s40", "global_load_lds_dwordx4 v15, s[24:25]", "s_nop 7", "s_nop 7"])
    # R4: the compiler pads nothing behind ';;#ASMEND' — the last MFMA needs its wait states inside the statement
    mf = ok + ["v_mfma_f32_16x16x32_bf16 a[0:3], v[96:99], v[128:131], a[0:3]"]
    with pytest.raises(AssertionError, match="R4"):
        g.audit_statement("mfma_tail", mf + ["s_nop 7"])
    g.audit_statement("mfma_tail_ok", mf + ["s_nop 15", "s_nop 7"])
    # operands (%N) and accumulator tiles are the compiler's business, not the audit's
    g.audit_statement("operands", ["v_mov_b32 %0, 1", "v_accvgpr_write_b32 a3, %1"])


def test_fused_kernel_keeps_nothing_in_scratch(tmp_path):
    """The compiler half of rule R1: a value hipcc keeps in scratch across the statements comes back through a
    scratch_load that can be pending when a statement starts (the fp16 race) and whose s_waitcnt vmcnt(0) drains the LDS-DMA
    in flight.  The C++ around the statements is written so that no instantiation of the fused kernel has any scratch
    traffic at all (launch-uniform tests and divisors are formed where they are used: flag(), lane_id(), div_pts in
    point_mfma.hip).  This compiles the kernel to ISA and checks exactly that."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    src = os.path.join(ROOT, "pixel_nerf_multiscale_amd", "csrc", "point_mfma.hip")
    out = tmp_path / "point_mfma.s"
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out), src],
                   check=True, capture_output=True, timeout=600)
    text = out.read_text().splitlines()
    kernels, name = {}, None
    for ln in text:
        s = ln.split(";")[0].strip()
        if s.endswith(":") and "k_point_mfma" in s and not s.startswith("."):
            name = s[:-1]
            kernels[name] = []
        elif s.startswith(".size") and name and name in s:
            name = None
        elif name:
            kernels[name].append("#ASMSTART" if "#ASMSTART" in ln else "#ASMEND" if "#ASMEND" in ln else s)
    assert len(kernels) == 4, list(kernels)            # {bf16, fp16} x {one view, several views}
    for k, body in kernels.items():
        bad = [s for s in body if s.startswith(("scratch_", "buffer_load", "buffer_store")) and not s.startswith("buffer_wbl2")]
        assert not bad, (k, bad[:4])
        # the pinned accumulator tiles a[0:255] belong to the statements: no compiler v_accvgpr_* move outside them
        # (cdna_hip_programming.md §5.7 item 4)
        inside, moved = False, []
        for s in body:
            if "#ASMSTART" in s:
                inside = True
            elif "#ASMEND" in s:
                inside = False
            elif not inside and s.startswith("v_accvgpr"):
                moved.append(s)
        assert not moved, (k, moved[:4])
    # the code-object metadata agrees: no VGPR spilled (SGPRs spill into VGPR lanes, not memory)
    meta = [ln.strip() for ln in text if ".vgpr_spill_count:" in ln or (".name:" in ln and "k_point_mfma" in ln)]
    names = [i for i, ln in enumerate(text) if ".name:" in ln and "k_point_mfma" in ln]
    assert len(names) == 4, meta
    for i in names:
        spill = next(ln for ln in text[i:i + 12] if ".vgpr_spill_count:" in ln)
        assert spill.split(":")[1].strip() == "0", (text[i].strip(), spill.strip())
