"""Detector probe on synthetic 4000 x 3000 images: r3dm_detect_akaze_batch at B = 1, 2, 4, 8 with the images resident in HBM
(per-image kernel time, fraction of the HBM roof of the pass structure), then the features stage (detect + LIOP + files) over an
image list with K contexts x batches of B -- what the reference runs one image at a time behind a semaphore."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()          # before the library: torch's HIP runtime has to come up first in a process that uses both
from regard3d_amd import api, synth
if any(k.startswith("R3DM_") for k in os.environ):
    api.use_developer_library()          # developer knobs (R3DM_*) only exist in libr3dm_dev.so

h, w = 3000, 4000
NIMG = int(os.environ.get("AK_IMAGES", "16"))
imgs = [synth.make_photo(h, w, seed=100 + k) for k in range(min(NIMG, 8))]
dimgs = [torch.from_numpy(im).cuda() for im in imgs]
torch.cuda.synchronize()
c = api.Context(0)
for thr in (0.001,):
    for B in [int(x) for x in os.environ.get("AK_BATCHES", "1,2,4,8").split(",")]:
        if B > len(dimgs): break
        for rep in range(3):
            t = time.time(); res = c.detect_akaze_batch(dimgs[:B], thr); dt = time.time() - t
        s = c.stats()
        print(json.dumps(dict(image=[h, w], threshold=thr, batch=B, keypoints=[len(r[0]) for r in res], ms_wall_per_image=dt / B * 1e3,
                              ms_kernels_per_image=s.ms_detect_kernels / B, algorithmic_GB_per_image=s.detect_algorithmic_bytes / B / 1e9,
                              hbm_frac=s.detect_algorithmic_bytes / (s.ms_detect_kernels * 1e-3) / 8e12, regrows=int(c.features_totals().n_regrows))), flush=True)
    t = time.time(); kps, resp = c.detect_akaze(imgs[0], thr); dt = time.time() - t
    t = time.time(); desc = c.extract_liop(imgs[0], kps, 8.0); dl = time.time() - t
    print(json.dumps(dict(single_image_from_host=True, keypoints=len(kps), s_detect=dt, s_liop=dl)), flush=True)
if os.environ.get("AK_STAGE", "1") == "0":
    sys.exit(0)
# the features stage over an image list (files to /tmp); every context has seen the image size once before the timed pass
import tempfile, shutil

d = tempfile.mkdtemp()
try:
    lst = [dimgs[k % len(dimgs)] for k in range(NIMG)]
    paths = lambda ext: [f"{d}/i{k}.{ext}" for k in range(NIMG)]
    for conc, batch in ((1, 1), (1, 8), (2, 8), (2, 4), (3, 4), (4, 2)):
        m = api.MultiContext([0] * conc)
        m.extract_features(lst, paths("feat"), paths("desc"), 0.001, batch=batch)          # warm-up: buffers of every context
        for f in os.listdir(d): os.remove(os.path.join(d, f))
        t = time.time(); nf, sk = m.extract_features(lst, paths("feat"), paths("desc"), 0.001, batch=batch); dt = time.time() - t
        for f in os.listdir(d): os.remove(os.path.join(d, f))
        m.close()
        print(json.dumps(dict(features_stage_images=NIMG, contexts=conc, batch=batch, images="device-resident", s_total=dt,
                              ms_per_image=dt / NIMG * 1e3, keypoints=int(nf[0]))), flush=True)
    hl = [imgs[k % len(imgs)] for k in range(NIMG)]
    for conc, batch in ((2, 4),):
        m = api.MultiContext([0] * conc)
        m.extract_features(hl, paths("feat"), paths("desc"), 0.001, batch=batch)
        for f in os.listdir(d): os.remove(os.path.join(d, f))
        t = time.time(); nf, sk = m.extract_features(hl, paths("feat"), paths("desc"), 0.001, batch=batch); dt = time.time() - t
        m.close()
        print(json.dumps(dict(features_stage_images=NIMG, contexts=conc, batch=batch, images="pageable host float32 (48 MB each)", s_total=dt,
                              ms_per_image=dt / NIMG * 1e3)), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
if len(sys.argv) > 1 and sys.argv[1] == "cpu":
    from oracle import pyoracle as o
    t = time.time(); r = o.akaze_detect(imgs[0], 0.001); print("oracle (OpenMP port) %.2fs, %d keypoints" % (time.time() - t, len(r["kps"])))
    print("equal:", np.array_equal(r["kps"], c.detect_akaze(imgs[0], 0.001)[0]))
