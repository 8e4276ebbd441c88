#!/usr/bin/env python
"""tools/isa_diff.py REV FILE.hip [FILE.hip ...] -- which gfx950 kernels of a source file changed since git revision REV?

Cross-compiles the file at REV (with the headers of REV) and in the working tree (hipcc -S, no GPU needed), strips labels /
branches / comments and compares the instruction streams kernel by kernel.  Used at the end of round 2 to show that the
experiments added after the last GPU run (staggered filter schedule, extra metrics, small-database kernel) left every
default instantiation instruction-identical to the measured build.  Template parameters appended with a default value
(...ELb0E) are folded so that renamed instantiations still match."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "faiss_amd", "csrc")
HDRS = ["common.h", "kernels.h", "wg_select.h", "wave_select.h"]


def asm(src_dir, name, out):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                           "-S", "--cuda-device-only", "-o", out, os.path.join(src_dir, name)], stderr=subprocess.DEVNULL)
    ks, cur = {}, None
    for line in open(out):
        m = re.match(r"^(_ZN9faiss_amd\S+):", line)
        if m and "kernel" in m.group(1):
            cur = re.sub(r"ELb0E(EEvNS_)", r"E\1", m.group(1))
            ks[cur] = []
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
        if cur is None:
            continue
        s = line.split(";")[0].strip()
        if s and not s.startswith(".") and not s.startswith("s_cbranch") and not s.startswith("s_branch"):
            ks[cur].append(re.sub(r"\.LBB[0-9_]+", "L", s))
    return ks


def main(rev, files):
    with tempfile.TemporaryDirectory() as tmp:
        for h in HDRS:
            with open(os.path.join(tmp, h), "wb") as f:
                f.write(subprocess.check_output(["git", "-C", ROOT, "show", "%s:faiss_amd/csrc/%s" % (rev, h)]))
        for name in files:
            name = os.path.basename(name)
            with open(os.path.join(tmp, name), "wb") as f:
                f.write(subprocess.check_output(["git", "-C", ROOT, "show", "%s:faiss_amd/csrc/%s" % (rev, name)]))
            old = asm(tmp, name, os.path.join(tmp, "old.s"))
            new = asm(CSRC, name, os.path.join(tmp, "new.s"))
            same = [k for k in old if new.get(k) == old[k]]
            print("%s: %d kernels at %s, %d identical, %d changed, %d removed, %d new" % (
                name, len(old), rev, len(same), sum(1 for k in old if k in new and new[k] != old[k]),
                sum(1 for k in old if k not in new), sum(1 for k in new if k not in old)))
            for k in old:
                if k in new and new[k] != old[k]:
                    print("   changed:", k[:110], len(old[k]), "->", len(new[k]), "instructions")
            for k in new:
                if k not in old:
                    print("   new:    ", k[:110])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
