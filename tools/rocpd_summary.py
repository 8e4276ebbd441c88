#!/usr/bin/env python
"""tools/rocpd_summary.py -- per-kernel statistics from a rocprofv3 rocpd (sqlite) database.

ROCm 7.2's rocprofv3 writes `<name>_results.db` by default; this prints the equivalent of the
`--stats` kernel table (calls, total / average / min / max duration, share of GPU time) as
text so the summary can be committed under profiles/.

usage: python tools/rocpd_summary.py gpurun_out/prof_x/x_results.db [> profiles/xxx.txt]
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(
        "select %s, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
        "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = float(sum(r[2] for r in rows)) or 1.0
    print("# source: %s" % path)
    print("%-88s %7s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) <= 88 else name[:85] + "..."
        print("%-88s %7d %14d %12.0f %12d %12d %6.2f%%" % (short, n, tot, avg, mn, mx, 100.0 * tot / total))
    # register / LDS footprint per kernel when the view exposes it
    want = [c for c in ("vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_size", "scratch_size", "workgroup_size",
                        "grid_size") if c in cols]
    if want:
        print("\n# per-kernel resources (first dispatch): " + ", ".join(want))
        for (name,) in cur.execute("select distinct %s from kernels" % name_col).fetchall():
            r = cur.execute("select %s from kernels where %s = ? limit 1" % (", ".join(want), name_col),
                            (name,)).fetchone()
            short = name if len(name) <= 88 else name[:85] + "..."
            print("%-88s %s" % (short, " ".join("%s=%s" % (c, v) for c, v in zip(want, r))))


if __name__ == "__main__":
    main(sys.argv[1])
