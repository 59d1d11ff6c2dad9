"""Print the kernel timeline of the last step (or, with a third argument, of n launches that many before the end) in a rocprofv3 kernel-trace csv
(start/end relative to the step start, per kernel, gap to the previous kernel's end) to see what overlaps
and what a kernel boundary costs."""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        m = re.search(r"(k_[a-z0-9_]+)", name)
        if not m:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1)))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # that many launches from the end are left out (the run's tail)
tail = rows[-(n + skip):-skip] if skip else rows[-n:]
t0 = tail[0][0]
prev_end = None
for s, e, k in tail:
    gap = "" if prev_end is None else "  gap %6.1f" % ((s - prev_end) / 1e3)
    print("%-24s start %9.1f us  end %9.1f us  dur %7.1f%s" % (k, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap))
    prev_end = e
