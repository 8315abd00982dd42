"""GPU occupancy of the stage's features phase from a rocprofv3 --kernel-trace CSV directory: the phase of a step runs from the first
detector kernel (ak_*) behind a match kernel to the last liop_kernel before the next match kernel; prints, per step, the phase's
wall time on the GPU clock, the union of all kernel intervals (time with at least one kernel running), the sum of kernel durations
and how many kernels ran side by side on average -- is the phase short of kernels (host gaps) or of GPU (contention)?
usage: trace_busy.py <dir>"""
import csv, glob, os, sys
d = sys.argv[1]
rows = [r for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True) for r in csv.DictReader(open(p))]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
def is_feat(n): return "ak_" in n or "liop" in n
def is_match(n): return "l2_knn2" in n or "acransac" in n
# split into phases: maximal runs of time in which feature kernels occur, separated by match / filter kernels
phases, cur = [], []
for s, e, n in ev:
    if is_match(n):
        if cur: phases.append(cur); cur = []
    elif is_feat(n) or cur:
        cur.append((s, e, n))
if cur: phases.append(cur)
for k, ph in enumerate(phases):
    fe = [x for x in ph if is_feat(x[2])]
    if len(fe) < 100: continue
    t0, t1 = min(x[0] for x in fe), max(x[1] for x in fe)
    iv = sorted((x[0], x[1]) for x in ph if x[0] < t1)
    union, cs, ce = 0, iv[0][0], iv[0][1]
    gaps = []
    for s, e in iv[1:]:
        if s > ce:
            union += ce - cs; gaps.append(s - ce); cs, ce = s, e
        else:
            ce = max(ce, e)
    union += ce - cs
    tot = sum(e - s for s, e in iv)
    big = sorted(gaps, reverse=True)[:5]
    print(f"phase {k}: {len(iv)} kernels, wall {(t1 - t0) / 1e6:.2f} ms, some kernel running {union / 1e6:.2f} ms ({union / (t1 - t0):.2f}), "
          f"sum of durations {tot / 1e6:.2f} ms (x{tot / union:.2f} side by side), idle gaps {len(gaps)} totalling {sum(gaps) / 1e6:.2f} ms, largest {[round(g / 1e3) for g in big]} us")
