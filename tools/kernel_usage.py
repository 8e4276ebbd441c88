"""tools/kernel_usage.py <file.hip> -- registers / scratch / occupancy of every kernel in a source file (hipcc -Rpass-analysis)"""
import re, subprocess, sys, os, tempfile
src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(tempfile.gettempdir(), "ku.o")],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(faiss_amd::\w+\)$", "", cur).replace("faiss_amd::", "")
        rows[cur] = {}
        continue
    for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur:
            rows[cur][key] = int(m.group(1))
for k, v in rows.items():
    print("%-70s %s" % (k[:70], " ".join("%s=%d" % kv for kv in v.items())))
