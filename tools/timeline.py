"""Print the kernel timeline of the last step in a rocprofv3 kernel-trace csv
(start/end relative to the step start, per kernel) to see what overlaps."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        short = [k for k in ("k_tx_plan", "k_copy", "k_rx_plan", "k_rx_apply") if k in name]
        if not short:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short[0]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tail = rows[-n:]
t0 = tail[0][0]
for s, e, k in tail:
    print("%-11s start %9.1f us  end %9.1f us  dur %7.1f" % (k, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
