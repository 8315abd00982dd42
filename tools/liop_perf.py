"""LIOP throughput: GPU kernel vs the reference's own vl_liop.c (oracle/_ref) on this box's host."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from regard3d_amd import api
from oracle import pyoracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
g = torch.Generator(device="cuda"); g.manual_seed(1)
img = torch.rand((n, 1, 41, 41), generator=g, device="cuda")
k = torch.tensor([1, 4, 6, 4, 1], device="cuda", dtype=torch.float32); k = (k[:, None] * k[None, :]); k /= k.sum()
P = torch.nn.functional.conv2d(img, k[None, None], padding=2)[:, 0].contiguous()
torch.cuda.synchronize()
c = api.Context(0)
for rep in range(3):
    t = time.time(); d, nt = c.liop_describe_patches(P); wall = time.time() - t
    ms = c.stats().ms_liop_kernel
    print(json.dumps({"rep": rep, "patches": n, "kernel_ms": ms, "kpts_per_s_kernel": n / (ms * 1e-3), "wall_ms": wall * 1e3, "resorted": nt,
                      "lds_bound_note": "6.7 KB in / 576 B out per patch: HBM GB/s = %.1f" % ((n * (6724 + 576)) / (ms * 1e-3) / 1e9)}))
m = 4096
hp = P[:m].cpu().numpy()
t = time.time(); ref = O.ref_liop(hp) if O.ref_liop_lib() is not None else O.liop_describe(hp); tc = time.time() - t
print(json.dumps({"cpu_reference_1thread_kpts_per_s": m / tc, "kind": "reference" if O.ref_liop_lib() is not None else "port", "equal": bool(np.array_equal(ref, d[:m]))}))
