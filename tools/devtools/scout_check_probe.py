"""tools/devtools/scout_check_probe.py -- the AC-RANSAC scout pass against the full evaluation on C2's 790 short pairs, developer build:
R3DM_FILTER_CHECK=1 skips nothing and checks every model; R3DM_FILTER_SCOUT=3 adds the scout's promises to the checks (count within the
bound, NFA bound: invariants 9-11 of kernels_filter.hip), =5 the same with the scout dividing exactly.  Round 6 found with it that the
scout's cross-lane reads of its running counts could be scheduled ahead of their write-back (a bound that was not one)."""
import os, sys, re, subprocess
here = os.path.dirname(os.path.abspath(__file__))
for sc in ("3", "5"):
    r = subprocess.run([sys.executable, os.path.join(here, "..", "filter_short_pairs.py"), "--reps", "1"],
                       env=dict(os.environ, R3DM_FILTER_SCOUT=sc, R3DM_FILTER_CHECK="1"), capture_output=True, text=True)
    out = r.stdout + r.stderr
    print("R3DM_FILTER_SCOUT=" + sc, "R3DM_FILTER_CHECK=1 -> rc", r.returncode, "(every model of F, E, H checked)" if r.returncode == 0 else "")
    i = out.find("filter invariant")
    if i >= 0:
        print("  " + out[i:i + 300].splitlines()[0])
    for ln in out.splitlines():
        if ln.startswith('{"kind"'):
            print("  " + ln[:140])
