#!/usr/bin/env python3
"""Kernel durations and the gaps between consecutive kernels of a rocprofv3 --kernel-trace csv (one stream)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]                      # the second half: steady state
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    name = b["Kernel_Name"].split("(")[0][-40:]
    dur[name].append(int(b["End_Timestamp"]) - int(b["Start_Timestamp"]))
    gap[name].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for k in dur:
    d, g = sorted(dur[k]), sorted(gap[k])
    print("%-42s n %5d  duration p50 %7.2f us  mean %7.2f | gap in front p50 %7.2f us mean %7.2f p90 %7.2f" % (
        k, len(d), d[len(d) // 2] / 1e3, sum(d) / len(d) / 1e3, g[len(g) // 2] / 1e3, sum(g) / len(g) / 1e3, g[int(len(g) * 0.9)] / 1e3))
