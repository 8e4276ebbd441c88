"""CPU check of the register / scratch budget of the gfx950 kernels (hipcc cross-compiles without a GPU): every .hip file is
compiled with -Rpass-analysis=kernel-resource-usage and the numbers are held against the budgets the designs rest on.
What this catches: a change that makes the register allocator spill in a hot loop (a lambda that stops being inlined
sends the MFMA operands of flat_filter_kernel to scratch -- round 2), or that costs a kernel the
occupancy its LDS / latency-hiding plan assumes (two 512-thread IVFPQ workgroups per CU = at most 128 VGPRs)."""
import re

import pytest


def _usage(remarks):
    kernels, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r" AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return kernels


@pytest.fixture(scope="module")
def usage(kernel_builds):
    allk = {}
    for name, (_, remarks) in kernel_builds.items():
        allk.update(_usage(remarks))
    assert len(allk) > 60
    return allk


def _pick(usage, *subs):
    got = {k: v for k, v in usage.items() if all(s in k for s in subs)}
    assert got, subs
    return got


def test_no_kernel_spills_beyond_the_known_cold_paths(usage):
    # the reservoir cut of the IVFFlat scan and one scalar-quantizer variant spill a few registers in their (rare)
    # selection path; nothing else may touch scratch at all
    # (list-major scan: a few per-item invariants are reloaded once per work item, outside the tile loop)
    # (ivf_lm_pq_kernel: 128 registers of B operands + 32 of accumulators under the 256-register ceiling of two waves
    # per SIMD: per-item values and the temporaries of staging / epilogue live in scratch, nothing inside the operand /
    # MFMA pipeline of a block -- test_list_major_scan checks the bench shape's instantiations more tightly)
    # (wave_select_kernel: scalar registers saved around the memory-streaming fallback loops; the 128 key registers stay)
    # (filter sweeps, ivf_lm_filter.hip: the instantiations beside the bench shapes -- rows of more than 128 coordinates, dsub = 1,
    # collect with a selector -- keep a few per-item values in scratch; test_filter_sweeps pins the bench shapes at zero.  Round 6:
    # the flush of the parked candidates holds four returning atomics per lane at once; beside 128 VGPRs of B operands (rows of
    # 257 .. 512 coordinates) that costs the collect sweep up to 68 bytes of scratch around the flush, once per 256 candidates)
    allowed = {"ivfflat_fused_kernel": 64, "ivfsq_fused_kernel": 48, "ivf_lm_scan_kernel": 48, "ivf_lm_pq_kernel": 320,
               "wave_select_kernel": 32, "ivf_lmf_flat_kernel": 72, "ivf_lmf_pq_kernel": 136}
    for name, u in usage.items():
        limit = max([v for k, v in allowed.items() if k in name] or [0])
        assert u["scratch"] <= limit, (name, u)


def test_flat_filter_kernel_budget(usage):
    # 8 waves x 128 queries: 128 VGPRs of query operands + 64 of accumulators must stay in registers, two waves per SIMD
    for name, u in _pick(usage, "flat_filter_kernel", "ELi4E").items():
        assert u["scratch"] == 0 and u["vgpr"] <= 256 and u["occupancy"] >= 2, (name, u)
    # 4 waves x 64 queries, two workgroups per CU
    for name, u in _pick(usage, "flat_filter_kernel", "ELi2E").items():
        assert u["scratch"] == 0 and u["occupancy"] >= 2, (name, u)


def test_ivf_scan_kernels_keep_their_occupancy(usage):
    # two 512-thread workgroups per CU (one query's table build under the other's gathers): 4 waves per SIMD
    for name, u in _pick(usage, "ivfpq_fused_kernel").items():
        assert u["scratch"] == 0 and u["occupancy"] >= 4, (name, u)
    for name, u in _pick(usage, "ivfsq_fused_kernel").items():
        assert u["occupancy"] >= 4, (name, u)
    for name, u in _pick(usage, "ivfflat_fused_kernel").items():
        assert u["vgpr"] <= 128 and u["occupancy"] >= 4, (name, u)  # 1024-thread workgroups
    for name, u in _pick(usage, "ivf_finish_kernel").items():
        assert u["scratch"] == 0 and u["occupancy"] >= 8, (name, u)


def test_exact_scan_and_helpers(usage):
    for name, u in _pick(usage, "flat_scan_kernel").items():
        assert u["scratch"] == 0 and u["occupancy"] >= 2, (name, u)
    for name, u in _pick(usage, "wave_select_kernel").items():  # 64 keys per lane in registers, three waves per SIMD
        assert u["vgpr"] <= 168 and u["occupancy"] >= 3, (name, u)
    for sub in ("flat_rerank_kernel", "flat_tighten_kernel", "select_k_kernel", "selector_mask_kernel",
                "flat_general_kernel"):
        for name, u in _pick(usage, sub).items():
            assert u["scratch"] == 0, (name, u)


def test_list_major_scan(usage):
    """IVFPQ: three 4-wave workgroups per CU (168 registers); IVFFlat and the scalar quantizer: two"""
    picked = _pick(usage, "ivf_lm_scan_kernel")
    assert len(picked) == 24  # metric x kind (IVFFlat, IVFPQ, scalar quantizer) x pass x (dpad == 128)
    for name, u in picked.items():
        assert u["scratch"] <= 48 and u["occupancy"] >= 2, (name, u)
    # the register-fed pass-2 kernel of IVFFlat: 64 + 64 registers of operands per lane, two waves per SIMD, no scratch;
    # the scalar quantizer's instantiations (8-bit, 4-bit, 6-bit, fp16 codes: 16 / 16 / 32 / 32 registers of codes) likewise
    picked = _pick(usage, "ivf_lm_flat_reg_kernel")
    assert len(picked) == 40  # metric x (dpad == 128) x (fp32 rows, four code types) x pass
    for name, u in picked.items():
        assert u["scratch"] == 0 and u["occupancy"] >= 2, (name, u)
    for name, u in _pick(usage, "ivf_lm_pq_kernel").items():
        assert u["occupancy"] >= 2, (name, u)
    for name, u in _pick(usage, "ivf_lm_pq_kernel", "ELi2ELb1E").items():  # dsub = 2, d = 128: PQ64 of the bench
        assert u["scratch"] <= 192, (name, u)


def test_filter_sweeps(usage):
    """ivf_lm_filter.hip (the file with the dominant kernel of every IVF leg was the one file this test did not
    compile).  The sweeps sit at 235-256 VGPRs by design -- 96 of B operands, 48 of accumulators, the A ring -- and round 5 lost
    three experiments to spills there.  Template arguments: <METRIC, MODE (1 sweep 1, 2 sweep 2, 3 estimate dump), query blocks,
    k-steps | dsub, ...>; the bench shapes are L2, three query blocks, 8 k-steps (IVFFlat d = 128) / dsub 2 (PQ64 over d = 128)."""
    flat = _pick(usage, "ivf_lmf_flat_kernel")
    pq = _pick(usage, "ivf_lmf_pq_kernel")
    assert len(flat) >= 60 and len(pq) >= 30
    for name, u in list(flat.items()) + list(pq.items()):
        assert u["vgpr"] <= 256 and u["occupancy"] >= 2, (name, u)  # two waves per SIMD: the other wave is the latency cover
    def targs(name):  # template arguments of an instantiation: ILi1ELi2ELi3ELi8ELb1ELb0ELb0E -> [1, 2, 3, 8, 1, 0, 0]
        return [int(x) for x in re.findall(r"L[ib](n?\d+)E", name.split("kernelI")[1].split("EEvNS_")[0] + "E")]

    for name, u in flat.items():
        metric, mode, nqb, ks, full, sel, pairb = targs(name)[:7]
        # IVFFlat / scalar quantizer, rows of <= 128 coordinates (KS = 8), the two sweeps, no selector: nothing in scratch
        if mode in (1, 2) and nqb == 3 and ks == 8 and not sel:
            assert u["scratch"] == 0, (name, u)
    for name, u in pq.items():
        metric, mode, nqb, ds = targs(name)[:4]
        # IVFPQ with dsub = 2, 4, 8 (PQ64 / PQ32 / PQ16 over d = 128): nothing in scratch, selector or not
        if mode in (1, 2) and ds in (2, 4, 8):
            assert u["scratch"] == 0, (name, u)
    # the kernels between the sweeps: prepare / bound / tighten / rerank never spill
    for sub in ("lmf_bound_kernel", "lmf_tighten_kernel", "lmf_pq_prepare_kernel", "lmf_sq_prepare_kernel", "lmf_rerank_flat_kernel",
                "lmf_rerank_pq64_kernel"):
        for name, u in _pick(usage, sub).items():
            assert u["scratch"] == 0, (name, u)
