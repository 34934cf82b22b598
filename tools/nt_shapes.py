#!/usr/bin/env python3
"""gemm_nt / sum_parts dispatches of a rocprofv3 --kernel-trace CSV grouped by grid: calls, average microseconds, total ms.
    python tools/nt_shapes.py <kernel_trace.csv> [substring ...]"""
import collections, csv, sys
subs = sys.argv[2:] or ["gemm_nt", "sum_parts"]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if any(s in n for s in subs):
        wx = int(r["Workgroup_Size_X"])
        agg[(n.split("(")[0][-24:], int(r["Grid_Size_X"]) // wx, r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(k, len(v), round(sum(v) / len(v), 1), round(sum(v) / 1e3, 2))
    tot += sum(v)
print("total ms", round(tot / 1e3, 2))
