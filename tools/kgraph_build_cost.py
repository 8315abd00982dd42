"""Cost of the exact K-NN graph index (the product's KGraph builder) by view size, up to R3DM_KGRAPH_MAX_ROWS:
   python tools/kgraph_build_cost.py            (the row 'NN-descent on the GPU' is closed with this table)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from regard3d_amd import api
c = api.Context(0)
rng = np.random.default_rng(1)
print("rows      build ms (HIP events)   ms per 1e9 row pairs   index bytes")
for n in (8192, 16384, 32768, 65536, 131072):
    a = np.rint(np.clip(rng.gamma(0.5, 60.0, (n, 128)), 0, 255)).astype(np.float32)
    c.clear_images(); c.set_image(0, a, None, 4000, 3000)
    best = None
    for _ in range(2):
        c.drop_indices()
        adj, deg = c.kgraph_index(0, n, 24)
        ms = c.stats().ms_ann_build
        best = ms if best is None or ms < best else best
    print(f"{n:7d}   {best:10.2f}              {best / (n * n / 1e9):8.2f}             {c.view_info(0)[1]:12d}", flush=True)
