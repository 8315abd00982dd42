"""Generates tests/golden/hnsw_ref_index.npz (run in the authoring container, needs /root/reference).

Pins the HNSW plugin path (matchingAlgorithm 6..8) with the reference-built library: for the three presets of
src/R3DComputeMatches.cpp:533-565 the reference's own hnswlib::HierarchicalNSW (src/thirdparty/hnswlib/hnswlib/hnswalg.h, compiled
where it lies into oracle/_ref/libref_hnsw.so, rows added from ONE thread in row order) is built over view 0 of a seeded scene and
exported as DATA: the level of every row, every link list, the entry point, and searchKnn(ef, 2) of every row of view 1.

Two scenes: integer-valued SIFT bins (128-D; equal distances are common, so the heap tie order of std::priority_queue matters) and
unit-length real-valued rows (144-D, the shape of LIOP).  The descriptors themselves come from regard3d_amd.synth by seed; the
fixture stores a checksum of them so a drifted generator fails loudly instead of silently un-pinning the test.
"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as O
from regard3d_amd import synth

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hnsw_ref_index.npz")
O.build()
assert O.ref_lib() is not None, "oracle/_ref not built: needs /root/reference"

SCENES = {"sift": dict(n=2000, kind="sift", seed=77), "liop": dict(n=1200, kind="liop", seed=78)}
data = {}
for name, s in SCENES.items():
    sc = synth.make_scene(2, s["n"], s["kind"], seed=s["seed"])
    d0, d1 = np.ascontiguousarray(sc.descs[0], np.float32), np.ascontiguousarray(sc.descs[1], np.float32)
    data[f"{name}_crc"] = np.array([zlib.crc32(d0.tobytes()), zlib.crc32(d1.tobytes())], np.uint32)
    data[f"{name}_scene"] = np.array([s["n"], s["seed"]], np.int32)
    for preset, (M, efc, ef) in O.HNSW_PRESETS.items():
        ix, idx, dist = O.ref_hnsw_export(d0, d1, M, efc, ef)
        # the restatement must reproduce the reference-built index before anything is written
        mine = O.hnsw_build(d0, M, efc)
        ex = mine.export()
        for k in ("levels", "links0", "up_off", "up_links"):
            assert np.array_equal(ex[k], ix[k]), (name, preset, k)
        assert ex["enterpoint"] == ix["enterpoint"] and ex["maxlevel"] == ix["maxlevel"]
        mi, md = mine.knn2(d1, ef)
        assert np.array_equal(mi, idx) and np.array_equal(md.view(np.uint32), dist.view(np.uint32)), (name, preset, "search")
        p = f"{name}_{preset}_"
        data[p + "levels"] = ix["levels"].astype(np.int8)
        data[p + "links0"] = ix["links0"].astype(np.int16)          # column 0 = the list length; ids < 32768
        data[p + "up_off"] = ix["up_off"].astype(np.int32)
        data[p + "up_links"] = ix["up_links"].astype(np.int16)
        data[p + "entry"] = np.array([ix["enterpoint"], ix["maxlevel"]], np.int32)
        data[p + "idx"] = idx.astype(np.int16)
        data[p + "dist"] = dist
        print(name, preset, "maxlevel", ix["maxlevel"], "mean degree", ix["links0"][:, 0].mean(), flush=True)
np.savez_compressed(out, **data)
print(out, os.path.getsize(out) / 1e3, "kB")

# ---- hnsw_ref_recall_8k.npz: the recall bar at the size of BASELINE's views (8,192 rows per view).  Only what the bar needs is
# stored: the exact 2-NN (reference-built BruteforceSearch) and the reference-built HierarchicalNSW's searchKnn rows per preset.
out8 = os.path.join(os.path.dirname(out), "hnsw_ref_recall_8k.npz")
data = {}
for name, s in {"sift": dict(kind="sift", seed=81), "liop": dict(kind="liop", seed=82)}.items():
    sc = synth.make_scene(2, 8192, s["kind"], seed=s["seed"])
    d0, d1 = np.ascontiguousarray(sc.descs[0], np.float32), np.ascontiguousarray(sc.descs[1], np.float32)
    data[f"{name}_crc"] = np.array([zlib.crc32(d0.tobytes()), zlib.crc32(d1.tobytes())], np.uint32)
    data[f"{name}_scene"] = np.array([8192, s["seed"]], np.int32)
    ei, ed = O.ref_knn(d0, d1, 2)
    data[f"{name}_exact"] = ei.astype(np.int16)
    for preset, (M, efc, ef) in O.HNSW_PRESETS.items():
        ix, idx, dist = O.ref_hnsw_export(d0, d1, M, efc, ef)
        data[f"{name}_{preset}_idx"] = idx.astype(np.int16)
        print(name, preset, "reference-built recall@1 %.4f @2 %.4f" % ((idx[:, 0] == ei[:, 0]).mean(), (idx[:, 1] == ei[:, 1]).mean()), flush=True)
np.savez_compressed(out8, **data)
print(out8, os.path.getsize(out8) / 1e3, "kB")
