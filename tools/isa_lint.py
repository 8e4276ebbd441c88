#!/usr/bin/env python
"""tools/isa_lint.py -- MFMA -> VALU wait-state lint of the gfx950 kernels (no GPU needed: hipcc -S cross-compiles).

Why: gfx950 does not interlock a VALU / LDS / VMEM access to a VGPR that a matrix instruction is still writing; the compiler's
hazard recogniser inserts the wait states for the instructions IT emits, but it does not look into `asm volatile` statements.
Round 5 found a wrong-answer bug of exactly this kind (an asm `v_max3_f32` right behind the last `v_mfma` of the inner-product
sweeps); the fix is a hand-counted `s_nop`, and nothing checked it.  This tool does:

  * `required_wait_states()` compiles a two-instruction probe per MFMA opcode and reads off how many wait states hipcc itself puts
    between the MFMA and the first VALU that touches its destination (the self-calibrated requirement, no table to keep);
  * `lint_file()` walks every kernel of a source file: from each `v_mfma` it follows the control flow (both arms of conditional
    branches) for that many wait states -- every instruction is one wait state, `s_nop N` is N + 1, the model of LLVM's
    GCNHazardRecognizer -- and reports any non-MFMA instruction that reads or writes a register of the MFMA's destination
    earlier.  Accesses inside `;;#ASMSTART` blocks are the ones the lint exists for and the only ones it FAILS on; compiler-
    emitted accesses are listed with `in_asm` False for information (hipcc's recogniser looks back a bounded distance through
    the control flow: ivf_lm_flat_reg_kernel has a three-branch path from the last f32 MFMA of a block to a v_readfirstlane that
    is 17 of 18 states long by this count -- every taken branch on it costs more than the one state it is counted as).

tests/test_isa_lint_cpu.py runs it over every .hip file of the library."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "faiss_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]

PROBE = r"""
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
extern "C" __global__ void probe_f16(const half8* a, const half8* b, float* out) {
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    out[threadIdx.x] = acc[0] + acc[15];
}
extern "C" __global__ void probe_f32(const float* a, const float* b, float* out) {
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    out[threadIdx.x] = acc[0] + acc[15];
}
"""

_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(operand):
    """set of ('v'|'a', n) named by one operand string"""
    out = set()
    for m in _REG.finditer(operand):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


class Ins:
    __slots__ = ("op", "operands", "in_asm", "line", "text")

    def __init__(self, op, operands, in_asm, line, text):
        self.op, self.operands, self.in_asm, self.line, self.text = op, operands, in_asm, line, text


def compile_asm(src, out_s, extra=()):
    """hipcc -S of one source file; returns the compiler's stderr (the kernel-resource-usage remarks when asked for)"""
    r = subprocess.run([HIPCC] + FLAGS + list(extra) + ["-S", "--cuda-device-only", "-o", out_s, src], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    return r.stderr


def parse_kernels(asm_path):
    """{kernel symbol: (instructions, {label: index of the first instruction behind it})}"""
    kernels, cur, labels, in_asm, name = {}, None, None, False, None
    with open(asm_path) as f:
        for ln, raw in enumerate(f, 1):
            s = raw.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            m = re.match(r"^([A-Za-z_][\w$.]*):", raw)
            if m and not m.group(1).startswith(".L"):
                name = m.group(1)
                cur, labels = [], {}
                kernels[name] = (cur, labels)
                continue
            if cur is None:
                continue
            if s.startswith(".Lfunc_end"):
                cur = None
                continue
            m = re.match(r"^(\.L[\w$]+):", s)
            if m:
                labels[m.group(1)] = len(cur)
                continue
            s = s.split(";")[0].strip()
            if not s or s.startswith("."):
                continue
            parts = s.split(None, 1)
            ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
            cur.append(Ins(parts[0], ops, in_asm, ln, s))
    return {k: v for k, v in kernels.items() if any(i.op.startswith("s_endpgm") for i in v[0])}


# quad-cycle passes an MFMA holds its SIMD's matrix pipe for (MI355X_MICROARCH.md: 32x32x16 f16 issues back to back every 32
# cycles, 32x32x2 f32 every 64).  Instructions of one wave issue in order, so an instruction BEHIND a later MFMA of the same wave
# cannot issue before that MFMA was accepted by the pipe, i.e. before the earlier MFMA has been through all its passes: a later
# MFMA is worth its predecessor's pass count in wait states, not one.  (LLVM's recogniser counts it as one and is therefore
# more conservative on straight-line code; on the arm of an s_cbranch_execz that skips hipcc's own s_nop it relies on the same
# fact -- flat_filter_kernel has such arms: three MFMAs, three SALU, then a VALU write 7 "LLVM states" behind the MFMA.)
PASSES = {"v_mfma_f32_32x32x16_f16": 8, "v_mfma_f32_32x32x2_f32": 16}


def wait_states(ins):
    if ins.op == "s_nop":
        return int(ins.operands[0], 0) + 1
    if ins.op.startswith("v_mfma"):
        return PASSES.get(ins.op, 2)
    return 1


def touched(ins):
    """registers an instruction reads or writes (all of its vector operands)"""
    out = set()
    for o in ins.operands:
        out |= regs_of(o)
    return out


def _walk(instrs, labels, start, dest, need, report, origin, seen):
    """follow the control flow from instruction index `start` with `need` wait states still to cover"""
    stack = [(start, need)]
    while stack:
        i, left = stack.pop()
        while left > 0 and i < len(instrs):
            if seen.get(i, 0) >= left:
                break
            seen[i] = left
            ins = instrs[i]
            if ins.op.startswith("v_mfma") or ins.op.startswith("v_smfma"):
                # an MFMA that overwrites / accumulates into the same registers is ordered by the matrix pipe itself (and is
                # tracked as an origin of its own); reading them as A / B operands is not
                ab = set()
                for o in ins.operands[1:3]:
                    ab |= regs_of(o)
                if ab & dest:
                    report(origin, ins, need - left)
                if regs_of(ins.operands[0]) & dest:
                    break
            elif touched(ins) & dest:
                report(origin, ins, need - left)
            if ins.op == "s_endpgm" or ins.op.startswith("s_setpc") or ins.op.startswith("s_swappc"):
                break
            left -= wait_states(ins)
            if ins.op == "s_branch":
                i = labels.get(ins.operands[0], len(instrs))
                continue
            if ins.op.startswith("s_cbranch"):
                t = labels.get(ins.operands[-1])
                if t is not None and left > 0:
                    stack.append((t, left))
            i += 1


def lint_kernel(instrs, labels, need_of):
    """list of (mfma line, offending line, wait states elapsed, needed, in_asm, text)"""
    found = []
    for i, ins in enumerate(instrs):
        if not ins.op.startswith("v_mfma"):
            continue
        need = need_of(ins.op)
        dest = regs_of(ins.operands[0])

        def report(origin, bad, elapsed, need=need):
            found.append((origin.line, bad.line, elapsed, need, bad.in_asm, bad.text))

        _walk(instrs, labels, i + 1, dest, need, report, ins, {})
    return found


def closest_access(instrs, labels, horizon=40):
    """per MFMA opcode: the smallest number of wait states between an MFMA and the first access to its destination, split
    into compiler-emitted and asm-emitted accesses (diagnostics: how close to the requirement each kind comes)"""
    best = {}
    for i, ins in enumerate(instrs):
        if not ins.op.startswith("v_mfma"):
            continue
        dest = regs_of(ins.operands[0])

        def report(origin, bad, elapsed):
            key = (origin.op, bad.in_asm)
            if key not in best or elapsed < best[key]:
                best[key] = elapsed

        _walk(instrs, labels, i + 1, dest, horizon, report, ins, {})
    return best


_required = None


def required_wait_states():
    """{mfma opcode: wait states hipcc puts between it and the first VALU access to its destination}"""
    global _required
    if _required is None:
        with tempfile.TemporaryDirectory() as tmp:
            src = os.path.join(tmp, "probe.hip")
            with open(src, "w") as f:
                f.write(PROBE)
            compile_asm(src, os.path.join(tmp, "probe.s"))
            ks = parse_kernels(os.path.join(tmp, "probe.s"))
        req = {}
        for name, (instrs, labels) in ks.items():
            for (op, in_asm), elapsed in closest_access(instrs, labels).items():
                req[op] = elapsed
        assert len(req) == 2, req
        _required = req
    return _required


def need_of_factory():
    req = required_wait_states()
    worst = max(req.values())
    return lambda op: req.get(op, worst)


def lint_asm(asm_path, asm_only=True):
    need_of = need_of_factory()
    out, closest = {}, {}
    for name, (instrs, labels) in parse_kernels(asm_path).items():
        f = [x for x in lint_kernel(instrs, labels, need_of) if x[4] or not asm_only]
        if f:
            out[name] = f
        for key, v in closest_access(instrs, labels).items():
            if key not in closest or v < closest[key]:
                closest[key] = v
    return out, closest


def main(files):
    print("required wait states (from hipcc's own probe code):", required_wait_states())
    rc = 0
    with tempfile.TemporaryDirectory() as tmp:
        for name in files:
            s = os.path.join(tmp, os.path.basename(name) + ".s")
            compile_asm(os.path.join(CSRC, os.path.basename(name)), s)
            bad, closest = lint_asm(s)
            print("%s: %d kernels with findings; closest accesses %s" % (name, len(bad), {"%s/%s" % (k[0][7:], "asm" if k[1] else "cc"): v
                                                                                           for k, v in sorted(closest.items())}))
            for k, f in bad.items():
                rc = 1
                for origin, line, elapsed, need, in_asm, text in f[:5]:
                    print("   %s: line %d (%s) touches the destination of the MFMA at line %d after %d of %d wait states: %s" % (
                        k[:90], line, "asm" if in_asm else "compiler", origin, elapsed, need, text))
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or ["ivf_lm_filter.hip", "flat_filter.hip"]))
