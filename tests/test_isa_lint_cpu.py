"""CPU check of the MFMA -> VALU wait states of every gfx950 kernel (tools/isa_lint.py).

The hardware does not interlock a VALU / LDS / VMEM access to a register a matrix instruction is still writing.  hipcc pads the
instructions it emits itself, but not the ones inside `asm volatile` statements -- round 5 shipped (and fixed by a hand-counted
s_nop) a wrong-answer bug of that kind in the inner-product sweeps.  The requirement is read off hipcc's own
code for a two-instruction probe; the walker follows both arms of every branch."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_required_wait_states_are_what_the_architecture_documents():
    import isa_lint
    if not os.path.exists(isa_lint.HIPCC):
        pytest.skip("hipcc not available")
    req = isa_lint.required_wait_states()
    # (passes + 2, + 1 on gfx950 for the packed-input forms: 8-pass f16 32x32x16 -> 11 idle states, 16-pass f32 32x32x2 -> 18)
    assert 8 <= req["v_mfma_f32_32x32x16_f16"] <= 20 and 12 <= req["v_mfma_f32_32x32x2_f32"] <= 24, req


def test_no_access_to_an_mfma_destination_inside_its_wait_states(kernel_builds):
    import isa_lint
    total_asm = 0
    for name, (asm, _) in kernel_builds.items():
        bad, closest = isa_lint.lint_asm(asm)
        assert not bad, (name, {k[:100]: v[:3] for k, v in list(bad.items())[:4]})
        need = isa_lint.required_wait_states()
        for (op, in_asm), elapsed in closest.items():
            if in_asm:
                assert elapsed >= need[op], (name, op, elapsed)
                total_asm += 1
            else:
                # hipcc's own accesses sit AT the requirement on straight-line code (which is what calibrates the walker); a few
                # multi-branch paths come out one or two states short by this count (tools/isa_lint.py header)
                assert elapsed >= need[op] - 4, (name, op, elapsed)
    assert total_asm >= 2  # (flat_filter.hip and ivf_lm_filter.hip read accumulators from asm statements: the lint sees them)


def test_the_lint_catches_round_5s_hazard_when_the_guard_is_removed(kernel_builds, tmp_path):
    """Sensitivity: with the hand-written `s_nop 15; s_nop 3` guards deleted from the assembly the asm v_max3 of the inner-product
    sweeps sits inside the wait states of the last MFMA -- the bug round 5 found by accident -- and the lint reports it."""
    import isa_lint
    src = open(kernel_builds["ivf_lm_filter.hip"][0]).read()
    cut = re.sub(r";;#ASMSTART\n\ts_nop 15\n\ts_nop 3\n\t;;#ASMEND\n", "", src)
    assert len(cut) < len(src)
    p = tmp_path / "noguard.s"
    p.write_text(cut)
    bad, closest = isa_lint.lint_asm(str(p))
    assert bad, "the lint does not see the unguarded asm reader"
    assert all("ivf_lmf_" in k for k in bad)
    # only inner-product instantiations (METRIC = 0: the first template argument) lose rows without the guard; the L2 ones read the
    # accumulators through a compiler-visible v_pk_fma first
    assert all(re.search(r"kernelILi0E", k) for k in bad), [k[:60] for k in bad if not re.search(r"kernelILi0E", k)][:5]
    assert any(f[4] and "v_max3_f32" in f[5] for fs in bad.values() for f in fs)
