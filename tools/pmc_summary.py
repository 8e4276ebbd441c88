#!/usr/bin/env python
"""tools/pmc_summary.py -- per-kernel means of rocprofv3 --pmc counters (CSV output), as text + JSON.

usage: python tools/pmc_summary.py OUT.txt OUT.json dir1[:substr] [dir2[:substr] ...]
Each dir is one rocprofv3 pass (`--pmc ... --kernel-trace --output-format csv -d dir`); with `:substr` only the
kernels whose name contains substr are taken from that pass (a pass of another workload launches small instances
of the same kernel templates).  Kernels that ran
for less than 20 us on average are left out.  FETCH_SIZE / WRITE_SIZE are reported in KiB as rocprofv3
prints them; `hbm_read_bytes_corrected` applies the gfx950 correction for 16-byte-per-lane reads
(x2, /opt/skills/guides/MI355X_MICROARCH.md, HBM section).
"""
import csv, glob, json, os, sys
from collections import defaultdict


def main(out_txt, out_json, dirs):
    acc = defaultdict(lambda: [0.0, 0, 0.0])  # (kernel, counter) -> sum value, launches, sum duration
    for d in dirs:
        d, _, only = d.partition(":")
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if only and only not in r["Kernel_Name"]:
                    continue
                k = (r["Kernel_Name"], r["Counter_Name"])
                a = acc[k]
                a[0] += float(r["Counter_Value"]); a[1] += 1
                a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    lines, js = [], {}
    for (kern, ctr), (v, n, dur) in sorted(acc.items()):
        if dur / n < 20000:
            continue
        lines.append("%-100s %-26s launches=%d mean=%g avg_duration_ns=%.0f" % (kern[:100], ctr, n, v / n, dur / n))
        e = js.setdefault(kern, {"avg_duration_ns": dur / n, "launches": n})
        e[ctr] = v / n
        if ctr == "FETCH_SIZE":
            e["hbm_read_bytes_corrected"] = 2.0 * 1024.0 * v / n
    hdr = ["# rocprofv3 --pmc passes on MI355X (gfx950), one counter group per run; means over the launches of each kernel",
           "# FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; FETCH_SIZE under-reports 16-byte-per-lane streaming reads by 2x",
           "# on gfx950 (MI355X_MICROARCH.md, HBM section) -> hbm_read_bytes_corrected = 2 x 1024 x raw in the JSON next to this file", ""]
    open(out_txt, "w").write("\n".join(hdr + lines) + "\n")
    json.dump(js, open(out_json, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
