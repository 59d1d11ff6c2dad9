"""Summarise rocprofv3 counter-collection CSVs (one --pmc pass per sub-directory of the
given directory) into one JSON: per kernel, the average counter value per launch.
FETCH_SIZE / WRITE_SIZE are KiB per dispatch; FETCH_SIZE is doubled into hbm_read_bytes
(gfx950 note in MI355X_MICROARCH.md), WRITE_SIZE is taken as reported."""
import collections
import csv
import glob
import json
import os
import re
import sys

root, out_path, command = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (sources_sha16: which kernels these counters were collected from)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"k_\w+|__amd_rocclr_\w+", row["Kernel_Name"])
        name = m.group(0) if m else row["Kernel_Name"][:40]
        a = agg[name][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
        vals[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
kernels = {}
for name, ctrs in sorted(agg.items()):
    k = {}
    for c, (n, v) in sorted(ctrs.items()):
        k[c + "_avg"] = round(v / n, 2)
        k["launches_" + c] = n
    if "FETCH_SIZE" in ctrs and "WRITE_SIZE" in ctrs:
        rd = 2 * 1024 * ctrs["FETCH_SIZE"][1] / ctrs["FETCH_SIZE"][0]
        wr = 1024 * ctrs["WRITE_SIZE"][1] / ctrs["WRITE_SIZE"][0]
        k["hbm_read_bytes"], k["hbm_write_bytes"], k["hbm_traffic_bytes"] = int(rd), int(wr), int(rd + wr)
        # The launches of a kernel differ in size (warm-ups, the short last round of a step, the
        # connection set-up): the figure to set against the algorithmic bytes of ONE full launch is
        # the mean over the launches within 10 % of the largest one, per counter.
        full = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            top = max(vals[name][c])
            sel = [v for v in vals[name][c] if v >= 0.9 * top]
            full[c] = (len(sel), sum(sel) / len(sel))
        k["launches_full_size"] = {"FETCH_SIZE": full["FETCH_SIZE"][0], "WRITE_SIZE": full["WRITE_SIZE"][0]}
        k["hbm_read_bytes_full_size_launch"] = int(2 * 1024 * full["FETCH_SIZE"][1])
        k["hbm_write_bytes_full_size_launch"] = int(1024 * full["WRITE_SIZE"][1])
        k["hbm_traffic_bytes_full_size_launch"] = (k["hbm_read_bytes_full_size_launch"]
                                                   + k["hbm_write_bytes_full_size_launch"])
    for a, b, label in (("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "frac_wave_time_parked"),
                        ("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "frac_wave_time_issuing"),
                        ("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "frac_wave_time_issue_stalled")):
        if a in ctrs and b in ctrs and ctrs[b][1] > 0:
            k[label] = round(ctrs[a][1] / ctrs[b][1], 4)
    kernels[name] = k
json.dump({"command": command,
           "sources_sha16": bench.sources_sha16(),
           "units": "FETCH_SIZE / WRITE_SIZE: KiB per dispatch; SQ_*: quad-cycles summed over waves; "
                    "averages per launch over the whole run; *_full_size_launch: over the launches within 10 % "
                    "of the kernel's largest (what one full drain / gather moves)",
           "gfx950_correction": "FETCH_SIZE reports 1/2 of a wide coalesced streaming read -> doubled in "
                                "hbm_read_bytes; WRITE_SIZE uncalibrated on gfx950, taken as reported",
           "kernels": kernels}, open(out_path, "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a.startswith(("hbm_traffic", "frac_", "launches_full"))} for k, v in kernels.items()
                  if k.startswith("k_")}))
