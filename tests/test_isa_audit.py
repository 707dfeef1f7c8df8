"""build.py's guard against the hipcc spill-placement miscompile: a VGPR spill store emitted in
a join ("Flow") block ahead of the `s_or_b64 exec` that reopens it."""
import textwrap

from nann_amd import isa_audit

BAD = textwrap.dedent("""\
    _Z6kernelv:
    \ts_load_dwordx2 s[0:1], s[4:5], 0x0
    .LBB0_327:
    \ts_andn2_b64 exec, exec, s[4:5]
    \ts_cbranch_execnz .LBB0_327
    .LBB0_328:                              ; %Flow2337
    \ts_mov_b64 s[26:27], 0x800
    \tscratch_store_dwordx2 off, v[56:57], off offset:160 ; 8-byte Folded Spill
    \ts_or_b64 exec, exec, s[2:3]
    \ts_barrier
    """)

GOOD = textwrap.dedent("""\
    _Z6kernelv:
    \tscratch_store_dwordx2 off, v[2:3], off offset:56 ; 8-byte Folded Spill
    .LBB0_10:                               ; %Flow12
    \ts_or_b64 exec, exec, s[2:3]
    \tscratch_store_dwordx2 off, v[56:57], off offset:160 ; 8-byte Folded Spill
    .LBB0_11:
    \ts_and_saveexec_b64 s[0:1], vcc
    \tscratch_load_dwordx2 v[2:3], off, off offset:56 ; 8-byte Folded Reload
    \ts_or_b64 exec, exec, s[0:1]
    """)


def test_flags_spill_ahead_of_exec_restore(tmp_path):
    f = tmp_path / "bad.s"
    f.write_text(BAD)
    hits = isa_audit.flow_hits(str(f))
    assert len(hits) == 1 and hits[0][0] == "_Z6kernelv" and "offset:160" in hits[0][3]


def test_accepts_spills_after_the_restore_or_inside_a_masked_region(tmp_path):
    f = tmp_path / "good.s"
    f.write_text(GOOD)
    assert isa_audit.flow_hits(str(f)) == []
    assert isa_audit.audit(str(f)) == []
