#!/usr/bin/env python
"""tools/register_perf.py -- what registering a collection costs (r3dm_set_image / r3dm_set_images), from where the rows are:
pageable host memory (numpy), page-locked host memory (torch pin_memory) and device memory (torch cuda tensors).

SURVEY.md section 8(d) starts the metric's clock at "descriptors resident in host RAM": this is the step between that and the
first match call (the reference: Regions_Provider::load, /root/reference/src/R3DComputeMatches.cpp:2040,2094-2095).

  python tools/register_perf.py [--images 200] [--feat 8192] [--kind sift|siftu8|liopc|akaze] [--dev] [--reps 5]
--dev loads the developer build.  Prints one line per (source, entry point) with ms per collection, GB/s and the HBM the
collection holds afterwards.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from regard3d_amd import api, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=200)
    ap.add_argument("--feat", type=int, default=8192)
    ap.add_argument("--kind", default="sift")
    ap.add_argument("--dev", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--split", action="store_true", help="r3dm_set_split_mfma on before the views are registered: rows + count tiles are staged behind every view")
    ap.add_argument("--integer", action="store_true", help="r3dm_set_integer_mfma on before the views are registered (its bf16 tiles wait for the first match call all the same)")
    a = ap.parse_args()
    if a.dev:
        api.use_developer_library()
    dev = torch.device("cuda", 0)
    kind = {"siftu8": "sift"}.get(a.kind, a.kind)
    descs, xys, _ = synth.make_scene_torch(a.images, a.feat, seed=2002, device=dev, kind=kind)
    if a.kind == "siftu8":
        descs = descs.to(torch.uint8)
    binary = a.kind == "akaze"
    torch.cuda.synchronize()
    hd = [descs[i].cpu().numpy().copy() for i in range(a.images)]
    hx = [xys[i].cpu().numpy().copy() for i in range(a.images)]
    pd = [descs[i].cpu().pin_memory() for i in range(a.images)]
    px = [xys[i].cpu().pin_memory() for i in range(a.images)]
    raw_bytes = sum(d.nbytes for d in hd)
    ids = list(range(a.images))
    ctx = api.Context(0)
    if a.split:
        ctx.set_split_mfma(True)
    if a.integer:
        ctx.set_integer_mfma(True)

    def free_hbm():
        return torch.cuda.mem_get_info(0)[0]

    def run(name, fn):
        best = None
        for _ in range(a.reps):
            ctx.clear_images()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            t_ret = time.perf_counter() - t0
            ctx.images_wait()
            t = time.perf_counter() - t0
            best = (t, t_ret) if best is None or t < best[0] else best
        print(f"{name:46s} {best[0] * 1e3:8.2f} ms resident ({best[1] * 1e3:8.2f} ms until the call returned)  "
              f"{raw_bytes / best[0] / 1e9:6.1f} GB/s of raw rows  {best[0] / a.images * 1e6:7.1f} us per view", flush=True)

    print(f"# {a.images} views x {a.feat} x {descs.shape[2]} {descs.dtype}, raw rows {raw_bytes / 1e6:.1f} MB, library {api.LIB_PATH}")
    run("pageable numpy, r3dm_set_images", lambda: ctx.set_images(ids, hd, hx, synth.WIDTH, synth.HEIGHT, binary=binary))
    run("pageable numpy, r3dm_set_image per view", lambda: [ctx.set_image(i, hd[i], hx[i], synth.WIDTH, synth.HEIGHT, binary=binary) for i in ids])
    run("page-locked host, r3dm_set_images", lambda: ctx.set_images(ids, pd, px, synth.WIDTH, synth.HEIGHT, binary=binary))
    run("device tensors, r3dm_set_images", lambda: ctx.set_images(ids, [descs[i] for i in ids], [xys[i] for i in ids], synth.WIDTH, synth.HEIGHT, binary=binary))
    run("device tensors, r3dm_set_image per view", lambda: [ctx.set_image(i, descs[i], xys[i], synth.WIDTH, synth.HEIGHT, binary=binary) for i in ids])
    # footprint: the collection registered once more from host memory into a trimmed context
    ctx.clear_images(); ctx.trim(); torch.cuda.synchronize()
    del descs, xys, pd, px
    torch.cuda.empty_cache()
    f0 = free_hbm()
    mem0 = ctx.memory_info()
    ctx.set_images(ids, hd, hx, synth.WIDTH, synth.HEIGHT, binary=binary, wait=True)
    f1 = free_hbm()
    print(f"HBM held by the registered collection: {(f0 - f1) / 1e6:.1f} MB = {(f0 - f1) / raw_bytes:.2f} x the raw rows "
          f"(f32-equivalent: {(f0 - f1) / (a.images * a.feat * hd[0].shape[1] * 4):.2f} x) by hipMemGetInfo; the library's own account "
          f"(r3dm_memory_info: view slabs, ring device, ring host): {[round(x / 1e6, 1) for x in ctx.memory_info()]} MB; "
          f"view 0 holds {ctx.view_info(0)[1] / 1e6:.2f} MB", flush=True)
    print(f"before that registration, after trim: {[round(x / 1e6, 1) for x in mem0]} MB", flush=True)
    pairs = np.array([[0, 1], [0, 2], [1, 2]], np.uint32)
    g = ctx.match_pairs(pairs, 0.8 if binary else 0.6, not binary)
    f2 = free_hbm()
    print(f"... after a first match call on the default path: {(f0 - f2) / 1e6:.1f} MB = {(f0 - f2) / raw_bytes:.2f} x; {g.num_pairs} pairs, {g.num_matches} matches", flush=True)


if __name__ == "__main__":
    main()
