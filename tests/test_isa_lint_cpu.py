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


def _calls_and_lds_casts(asm_path):
    calls = casts = 0
    with open(asm_path) as f:
        for line in f:
            s = line.split(";")[0]
            calls += "s_swappc_b64" in s
            casts += "src_shared_base" in s
    return calls, casts


def test_lds_is_never_reached_through_flat_instructions(kernel_builds):
    """Root cause of round 5's "equivalent code faults" (wg_select.h): hipcc kept wg_select_kth OUT OF LINE; an
    out-of-line function sees its LDS arguments as generic pointers, so the histogram of the radix select was zeroed, incremented
    and read with FLAT instructions, and a no-return flat_atomic_add that lands in LDS is not complete when the `s_waitcnt
    lgkmcnt(0)` in front of the barrier lets the wave through -- the scan then misses increments, the k-th key comes out too large,
    more than k keys survive and overrun the partial-result slot.  The same lowering happens to `volatile` accesses through a generic
    pointer.  Both leave fingerprints in the ISA: a call (`s_swappc_b64`) and an LDS -> generic pointer cast (`src_shared_base`).
    No kernel file may contain either."""
    for name, (asm, _) in kernel_builds.items():
        calls, casts = _calls_and_lds_casts(asm)
        assert calls == 0, "%s: %d out-of-line device function calls (a helper lost its __forceinline__?)" % (name, calls)
        assert casts == 0, "%s: %d LDS pointers cast to generic (volatile access without lds_volatile(), common.h?)" % (name, casts)


def test_the_lint_sees_the_out_of_line_build_that_faulted(tmp_path):
    """ivf_fused.hip built the way it was when it faulted (wg_select_kth out of line: -DFAISS_AMD_WGS_OUTOFLINE_REPRO, the build
    tools/wgs_fault_repro.sh runs on the GPU) shows both fingerprints."""
    import isa_lint
    if not os.path.exists(isa_lint.HIPCC):
        pytest.skip("hipcc not available")
    out = str(tmp_path / "outofline.s")
    isa_lint.compile_asm(os.path.join(isa_lint.CSRC, "ivf_fused.hip"), out, extra=["-DFAISS_AMD_WGS_OUTOFLINE_REPRO"])
    calls, casts = _calls_and_lds_casts(out)
    assert calls >= 50 and casts >= 10, (calls, casts)
    text = open(out).read()
    assert "flat_atomic_add" in text and "wg_select_kth" in text
