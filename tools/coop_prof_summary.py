"""Sums of the per-pair phase table the developer build prints with R3DM_COOP_PROF=1 (stderr of tools/filter_coop_perf.py):
   python tools/coop_prof_summary.py <stderr file> [last N lines per kind]"""
import re, sys
import numpy as np
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 0
keys = ["wall", "init", "solve", "form+publish", "slices", "arrive", "bounds", "full", "walk", "batches", "models", "task delay sum", "max", "longest slice"]
for kind in "FEH":
    rows = [l for l in open(sys.argv[1]) if l.startswith("coop " + kind + " ")]
    if n_last: rows = rows[-n_last:]
    if not rows: continue
    acc = {k: [] for k in keys}
    for l in rows:
        for k in keys:
            m = re.search(re.escape(k) + r" (\d+)", l)
            acc[k].append(int(m.group(1)) if m else 0)
    print(kind, len(rows), "pairs:", {k: (int(np.sum(v)), int(np.max(v))) for k, v in acc.items()})
