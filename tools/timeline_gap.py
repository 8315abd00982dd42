"""Kernel timeline around a named kernel from a rocprofv3 --kernel-trace CSV directory: what ran in the N ms before each launch of it
and how long the GPU sat idle.  python tools/timeline_gap.py <dir> <kernel substring> [window_ms]"""
import csv, glob, sys
d, name = sys.argv[1], sys.argv[2]
win = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 80e6
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
for i, (s, e, k) in enumerate(rows):
    if name in k:
        prev = [x for x in rows[:i] if x[1] > s - win]
        busy = sum(min(x[1], s) - max(x[0], s - win) for x in prev)
        last_end = max((x[1] for x in rows[:i]), default=s)
        names = {}
        for x in prev: names[x[2][:40]] = names.get(x[2][:40], 0) + (x[1] - x[0]) / 1e6
        top = sorted(names.items(), key=lambda t: -t[1])[:4]
        print(f"{k[:40]} start: idle since last kernel end {(s - last_end) / 1e6:.2f} ms; busy in the {win / 1e6:.0f} ms before: {busy / 1e6:.1f} ms; top: {top}")
        if len(sys.argv) > 4:
            print("   last launches before it (start, end relative to its start, ms):")
            for x in rows[max(0, i - int(sys.argv[4])):i]:
                print(f"     {(x[0] - s) / 1e6:9.3f} {(x[1] - s) / 1e6:9.3f}  {x[2][:60]}")
