#!/usr/bin/env python3
"""Per-(kernel, grid) average durations from a rocprofv3 --kernel-trace CSV: the conv / wgrad kernels are one template
instantiation for many layers, so the per-name average of --stats mixes layers; a layer is identified by its grid.
usage: scripts/trace_by_grid.py <kernel_trace.csv> [name substring ...]"""
import csv, sys, collections
path, pats = sys.argv[1], sys.argv[2:] or ["conv_igemm_kernel", "wgrad_kernel"]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"]
    if not any(p in name for p in pats):
        continue
    grid = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]), int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_Z"]))
    agg[(name[:70], grid, int(r["LDS_Block_Size"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-70s %-18s %7s %5s %9s %9s %9s" % ("kernel", "workgroups", "LDS", "calls", "avg_us", "min_us", "max_us"))
for (name, grid, lds), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-70s %-18s %7d %5d %9.1f %9.1f %9.1f" % (name, "x".join(map(str, grid)), lds, len(v), sum(v) / len(v), min(v), max(v)))
