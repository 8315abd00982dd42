"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch.
With --by-grid the dispatches of a kernel are bucketed by their grid size (one row per evolution level / batch size)."""
import csv, glob, sys, collections
args = [a for a in sys.argv[1:] if not a.startswith("--")]
by_grid = "--by-grid" in sys.argv
for d in args:
    for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(path) as f:
            for row in csv.DictReader(f):
                key = row["Kernel_Name"].split("(")[0][-60:]
                if by_grid:
                    key += " grid=" + row.get("Grid_Size", "?")
                agg[(row["Kernel_Name"], key)][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for (k, name), v in sorted(agg.items(), key=lambda kv: -max(sum(x) for x in kv[1].values())):
            if "r3dm" in k:
                print(name, " ".join(f"{c}={sum(x)/len(x):.6g}(n={len(x)})" for c, x in sorted(v.items())))
