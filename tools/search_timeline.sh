#!/bin/bash
# tools/search_timeline.sh SCRIPT NB OUT -- rocprofv3 kernel trace of tools/SCRIPT 3 NB; prints the LAST search as a timeline
# (start offset, duration, gap to the previous dispatch) -> gpurun_out/OUT
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tl_tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl_tmp -o t -- python $R/tools/$1 3 $2 > $O/tl_tmp.log 2>&1
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/tl_tmp/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in csv.DictReader(open(f))]
for f in glob.glob("$O/tl_tmp/**/*memory_copy_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") ) for r in csv.DictReader(open(f))]
rows.sort()
plans = [i for i, r in enumerate(rows) if "lm_plan_kernel" in r[2]]
# the last search: from the first dispatch after the previous search's last kernel ... take [plans[-2] .. plans[-1]] shifted to start at the prep
a, b = plans[-2], plans[-1]
seg = rows[a:b]
t0 = seg[0][0]
with open("$O/$3", "w") as fo:
    fo.write("# one search of tools/$1 3 $2 (from one lm_plan_kernel to the next): start us, duration us, idle gap before us, name\n")
    prev_end = seg[0][0]
    for s, e, n in seg:
        fo.write("%9.1f %8.1f %7.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n))
        prev_end = max(prev_end, e)
    fo.write("# span %.1f us, busy %.1f us\n" % ((seg[-1][1] - t0) / 1e3, sum(e - s for s, e, n in seg) / 1e3))
PY
rm -rf $O/tl_tmp
cat $O/$3
