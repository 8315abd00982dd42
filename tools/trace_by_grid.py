"""Per-launch durations of the library's kernels bucketed by (kernel, grid size) from a rocprofv3 --kernel-trace CSV directory:
which LEVEL (grid) of a kernel costs what.  usage: trace_by_grid.py <dir> [name filter] [min total us]"""
import collections, csv, glob, os, sys
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else "r3dm::"; min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)]
rows = [r for p in f for r in csv.DictReader(open(p))]
b = collections.defaultdict(list)
for r in rows:
    if flt not in r["Kernel_Name"]:
        continue
    g = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]) // max(int(r["Workgroup_Size_Y"]), 1), int(r["Grid_Size_Z"]) // max(int(r["Workgroup_Size_Z"]), 1))
    b[(r["Kernel_Name"].split("(")[0].replace("r3dm::", ""), g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = collections.defaultdict(float)
for (k, g), v in b.items():
    tot[k] += sum(v)
print(f"{'kernel':34s} {'grid (workgroups)':>22s} {'calls':>6s} {'total us':>10s} {'avg us':>9s} {'min us':>9s}")
for (k, g), v in sorted(b.items(), key=lambda kv: (-tot[kv[0][0]], -sum(kv[1]))):
    if sum(v) >= min_us:
        print(f"{k[:34]:34s} {str(g):>22s} {len(v):6d} {sum(v):10.1f} {sum(v) / len(v):9.1f} {min(v):9.1f}")
