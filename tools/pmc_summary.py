"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(path) as f:
            for row in csv.DictReader(f):
                agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            if "r3dm" in k:
                name = k.split("(")[0][-60:]
                print(name, " ".join(f"{c}={sum(x)/len(x):.6g}(n={len(x)})" for c, x in sorted(v.items())))
