"""The hand-scheduled statements of the fused kernel are generated (tools/gen_resblock_asm.py).  CPU checks: the committed
.inc is what the generator produces today, and the generator's audit enforces the statement rules R1-R3 of its header —
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
    # operands (%N) and accumulator tiles are the compiler's business, not the audit's
    g.audit_statement("operands", ["v_mov_b32 %0, 1", "v_accvgpr_write_b32 a3, %1"])
