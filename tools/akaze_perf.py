"""Detector probe: r3dm_detect_akaze on a synthetic 4000 x 3000 image (the reference's semaphore-serialised stage)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()          # before the library: torch's HIP runtime has to come up first in a process that uses both
from regard3d_amd import api

h, w = 3000, 4000
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
img = 0.5 + 0.1 * np.sin(xx / 17.0) * np.cos(yy / 23.0)
for _ in range(1500):
    cx, cy = rng.uniform(40, w - 40), rng.uniform(40, h - 40); s = rng.uniform(2, 12); a = rng.uniform(0.15, 0.45) * rng.choice([-1, 1])
    x0, x1, y0, y1 = int(max(cx - 5 * s, 0)), int(min(cx + 5 * s, w)), int(max(cy - 5 * s, 0)), int(min(cy + 5 * s, h))
    img[y0:y1, x0:x1] += a * np.exp(-((xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2) / (2 * s * s))
img = np.clip(img + rng.normal(0, 0.01, img.shape), 0, 1).astype(np.float32)
c = api.Context(0)
for thr in (0.001, 0.0001):
    for rep in range(3):
        t = time.time(); kps, resp = c.detect_akaze(img, thr); dt = time.time() - t
    t = time.time(); desc = c.extract_liop(img, kps, 8.0); dl = time.time() - t
    print(json.dumps(dict(image=[h, w], threshold=thr, keypoints=len(kps), s_detect=dt, s_liop=dl, mpix_per_s=h * w / dt / 1e6)), flush=True)
# the features stage over an image list: 16 copies of the image, K images in flight on K contexts of the one GPU (files to /tmp).
# Every context has seen the image size once before the timed pass (work buffers allocated, launch sequence captured).
import tempfile, shutil
print(json.dumps(dict(scale_space_graph_replays=int(c.stats().n_ak_graph_replays))), flush=True)
d = tempfile.mkdtemp()
try:
    imgs = [img] * 16
    paths = lambda ext: [f"{d}/i{k}.{ext}" for k in range(16)]
    for conc in (1, 2, 3, 4):
        m = api.MultiContext([0] * conc)
        m.extract_features(imgs[:conc], paths("feat")[:conc], paths("desc")[:conc], 0.001)          # warm-up: one image per context
        for f in os.listdir(d): os.remove(os.path.join(d, f))
        t = time.time(); nf, sk = m.extract_features(imgs, paths("feat"), paths("desc"), 0.001); dt = time.time() - t
        for f in os.listdir(d): os.remove(os.path.join(d, f))
        m.close()
        print(json.dumps(dict(features_stage_images=16, contexts=conc, s_total=dt, ms_per_image=dt / 16 * 1e3, keypoints=int(nf[0]))), flush=True)
    # the same with the images already on the device (what is left when the 48 MB host-to-device copy per image is not in the way)
    dimgs = [torch.from_numpy(np.ascontiguousarray(img)).cuda() for _ in range(4)] * 4
    torch.cuda.synchronize()
    for conc in (1, 4):
        m = api.MultiContext([0] * conc)
        m.extract_features(dimgs[:conc], paths("feat")[:conc], paths("desc")[:conc], 0.001)
        for f in os.listdir(d): os.remove(os.path.join(d, f))
        t = time.time(); nf, sk = m.extract_features(dimgs, paths("feat"), paths("desc"), 0.001); dt = time.time() - t
        for f in os.listdir(d): os.remove(os.path.join(d, f))
        m.close()
        print(json.dumps(dict(features_stage_images=16, contexts=conc, images="device-resident", s_total=dt, ms_per_image=dt / 16 * 1e3)), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
if len(sys.argv) > 1 and sys.argv[1] == "cpu":
    from oracle import pyoracle as o
    t = time.time(); r = o.akaze_detect(img, 0.001); print("oracle (OpenMP port) %.2fs, %d keypoints" % (time.time() - t, len(r["kps"])))
    print("equal:", np.array_equal(r["kps"], c.detect_akaze(img, 0.001)[0]))
