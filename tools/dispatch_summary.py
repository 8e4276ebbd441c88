#!/usr/bin/env python
"""tools/dispatch_summary.py -- per-dispatch durations of the named kernels from a rocprofv3 --kernel-trace CSV.
usage: dispatch_summary.py TRACE_DIR OUT.csv kernel_substr [kernel_substr ...]"""
import csv, glob, os, sys


def main(d, out, subs):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(s in r["Kernel_Name"] for s in subs):
                rows.append((r["Kernel_Name"][:90], int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                             r.get("VGPR_Count", ""), r.get("SGPR_Count", ""), r.get("LDS_Block_Size", ""),
                             r.get("Grid_Size", ""), r.get("Workgroup_Size", "")))
    rows.sort(key=lambda r: r[1])
    with open(out, "w") as fo:
        fo.write("kernel,start_ns,duration_ns,vgpr,sgpr,lds_bytes,grid,workgroup\n")
        for r in rows:
            fo.write(",".join(str(x).replace(",", ";") for x in r) + "\n")
    by = {}
    for r in rows:
        by.setdefault(r[0], []).append(r[2])
    for k, v in by.items():
        big = [x for x in v if x > 0.5 * max(v)]
        print("%-90s dispatches=%d (>= half of the longest: %d, mean %.1f us)" % (k, len(v), len(big), sum(big) / len(big) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
