"""Summarise a rocprofv3 results .db (ROCm 7 rocpd format) into a short, committable text table."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
tot = sum(r[2] for r in rows)
print(f"# source: {sys.argv[1]}  (durations in MICROSECONDS, from rocprofv3 --kernel-trace --stats)")
print(f"{'kernel':<70} {'calls':>7} {'total':>16} {'avg':>16} {'pct':>7}")
other = 0.0
for name, calls, total, avg, pct in rows:
    if name.startswith(("r3dm::", "void r3dm::")) or pct >= 0.5:
        print(f"{name[:70]:<70} {calls:>7} {total:>16.1f} {avg:>16.1f} {pct:>7.2f}")
    else:
        other += total
print(f"{'(all other kernels: torch data generation, memcpy, fills)':<70} {'':>7} {other:>16.1f} {'':>16} {100*other/tot:>7.2f}")
