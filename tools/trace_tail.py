#!/usr/bin/env python3
"""Average duration of the LAST k calls' dispatches of every kernel whose name contains `pattern`, from a rocprofv3
--kernel-trace CSV directory:  python tools/trace_tail.py <dir> <k> <pattern>.  With tools/plan_probe.py --only cold the
last k calls of the process are its cold ones, so this splits a cold transform into its passes (kernels in launch order; a
kernel that serves several passes of one call is averaged over all of them)."""
import csv, glob, os, sys
d, k, pat = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if pat in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
by = {}
for s, e, n in rows:
    by.setdefault(n, []).append(e - s)
if not by:
    sys.exit("no dispatch of *%s* in %s" % (pat, d))
fewest = min(len(v) for v in by.values())
for n, sel in by.items():
    per = max(1, round(len(sel) / fewest))
    t = sel[-k * per:]
    print("   %-100s x%d per call   avg %8.2f us   (min %.2f, max %.2f over the last %d dispatches)" %
          (n.split("(")[0][-100:], per, sum(t) / len(t) / 1e3, min(t) / 1e3, max(t) / 1e3, len(t)))
