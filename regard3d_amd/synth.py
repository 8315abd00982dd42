"""Deterministic synthetic multi-view scenes for the compute-matches hot path.

BASELINE.json's configs are synthetic workloads of the reference's operator family (the reference
ships no data).  The generator builds a scene in which the F-matrix filter
has real work to do:

  * a long relief "wall" of world points ordered along X; image i observes the window
    [i*s, i*s + Wd) of world points (Wd = 0.6 n, s = Wd/4), so images 1/2/3 steps apart share
    45 % / 30 % / 15 % of their features and images >= 4 steps apart share none
    (~3 % of the exhaustive pairs of a 200-image set carry true correspondences);
  * the remaining 0.4 n features of every image are distractors (uniform positions, independent
    descriptors);
  * cameras translate along the wall with a small random rotation; keypoints are pinhole
    projections (f = 1.2 W, 4000 x 3000) + N(0, 0.5 px); 15 % of the observations get a uniformly
    random position instead (descriptor matches, geometry does not -> RANSAC outliers);
  * descriptors: "sift" = 128 gamma(0.5) bins, L2-normalised, x512, clipped to 255, observation noise
    N(0, 6), rounded to integers (stored f32 or u8); "liop" = the same without rounding,
    re-normalised to unit length, 144-D f32 (the reference's live descriptor, LIOP,
    /root/reference/src/Regard3DFeatures.h:44,48); "akaze" = 486 random bits, 8 % flips per
    observation, packed LSB-first into 61 bytes and zero-padded to 64
    (/root/reference/src/thirdparty/fast-akaze/AKAZEFeatures.cpp:1069-1072).

Everything is numpy default_rng streams keyed by (seed, image) so any rank can regenerate any image.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np

WIDTH, HEIGHT = 4000, 3000


@dataclass
class Scene:
    kind: str
    dim: int                      # floats (sift/liop) or bytes (akaze, 64 incl. padding)
    descs: List[np.ndarray]       # [n, dim] float32 / uint8
    xys: List[np.ndarray]         # [n, 2] float32 pixel coordinates
    world_ids: List[np.ndarray]   # [n] int64, -1 for distractors
    widths: np.ndarray = field(default=None)
    heights: np.ndarray = field(default=None)

    @property
    def n_images(self) -> int:
        return len(self.descs)

    def exhaustive_pairs(self) -> np.ndarray:
        """openMVG exhaustivePairs(N): all (I, J), I < J, in (I, J) order
        (/root/reference/src/R3DComputeMatches.cpp:2042)."""
        n = self.n_images
        i, j = np.triu_indices(n, k=1)
        return np.stack([i, j], axis=1).astype(np.uint32)


def _sift_from_base(base: np.ndarray, rng, sigma: float, integer: bool) -> np.ndarray:
    d = base + sigma * rng.standard_normal(base.shape, dtype=np.float32)
    np.clip(d, 0.0, 255.0, out=d)
    if integer:
        np.rint(d, out=d)
    return d


def _sift_base(rng, n: int, dim: int) -> np.ndarray:
    g = rng.standard_gamma(0.5, size=(n, dim)).astype(np.float32)
    g /= np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-12)
    g *= 512.0
    np.clip(g, 0.0, 255.0, out=g)
    return g


def intrinsics() -> np.ndarray:
    """pinhole K of every synthetic view: f = 1.2 * width, principal point at the image centre"""
    f = 1.2 * WIDTH
    return np.array([[f, 0.0, 0.5 * WIDTH], [0.0, f, 0.5 * HEIGHT], [0.0, 0.0, 1.0]], np.float64)


def make_scene(n_images: int, n_feat: int, kind: str = "sift", seed: int = 2002,
               dtype: str = "f32", outlier_frac: float = 0.15, dim: int | None = None) -> Scene:
    assert kind in ("sift", "liop", "liopc", "akaze")
    if dim is None:
        dim = {"sift": 128, "liop": 144, "liopc": 144, "akaze": 64}[kind]
    n_shared = int(round(0.6 * n_feat))
    step = max(n_shared // 4, 1)
    n_world = (n_images - 1) * step + n_shared

    rw = np.random.default_rng([seed, 0x57])
    # world geometry: wall along X with relief in Z
    f = 1.2 * WIDTH
    Z0 = 10.0
    win_width = 0.8 * WIDTH * Z0 / f              # world-space width of one window
    delta = win_width / n_shared
    wx = (np.arange(n_world) + rw.uniform(-0.3, 0.3, n_world)) * delta
    wy = rw.uniform(-0.4 * HEIGHT * Z0 / f, 0.4 * HEIGHT * Z0 / f, n_world)
    wz = Z0 * rw.uniform(0.75, 1.25, n_world)
    # world appearance
    if kind == "akaze":
        wbits = rw.integers(0, 2, size=(n_world, 486), dtype=np.uint8)
    else:
        wbase = _sift_base(rw, n_world, dim)

    descs, xys, wids = [], [], []
    for i in range(n_images):
        r = np.random.default_rng([seed, 0x1A6E, i])
        lo = i * step
        ids = np.arange(lo, lo + n_shared)
        n_dis = n_feat - n_shared
        # camera: centre above the window, small yaw/pitch/roll
        cx = (lo + 0.5 * n_shared) * delta
        ang = r.normal(0.0, 0.02, 3)
        ca, sa = np.cos(ang), np.sin(ang)
        Rx = np.array([[1, 0, 0], [0, ca[0], -sa[0]], [0, sa[0], ca[0]]])
        Ry = np.array([[ca[1], 0, sa[1]], [0, 1, 0], [-sa[1], 0, ca[1]]])
        Rz = np.array([[ca[2], -sa[2], 0], [sa[2], ca[2], 0], [0, 0, 1]])
        R = Rz @ Ry @ Rx
        P = np.stack([wx[ids] - cx, wy[ids], wz[ids]], axis=1) @ R.T
        u = f * P[:, 0] / P[:, 2] + 0.5 * WIDTH + r.normal(0.0, 0.5, n_shared)
        v = f * P[:, 1] / P[:, 2] + 0.5 * HEIGHT + r.normal(0.0, 0.5, n_shared)
        bad = r.random(n_shared) < outlier_frac
        u[bad] = r.uniform(0, WIDTH, int(bad.sum()))
        v[bad] = r.uniform(0, HEIGHT, int(bad.sum()))
        xy = np.empty((n_feat, 2), np.float32)
        xy[:n_shared, 0] = u; xy[:n_shared, 1] = v
        xy[n_shared:, 0] = r.uniform(0, WIDTH, n_dis); xy[n_shared:, 1] = r.uniform(0, HEIGHT, n_dis)

        if kind == "akaze":
            bits = np.empty((n_feat, 486), np.uint8)
            flips = (r.random((n_shared, 486)) < 0.08).astype(np.uint8)
            bits[:n_shared] = wbits[ids] ^ flips
            bits[n_shared:] = r.integers(0, 2, size=(n_dis, 486), dtype=np.uint8)
            padded = np.zeros((n_feat, 488), np.uint8); padded[:, :486] = bits
            packed = np.packbits(padded, axis=1, bitorder="little")          # 61 bytes, LSB first
            d = np.zeros((n_feat, dim), np.uint8); d[:, :61] = packed
        else:
            integer = kind in ("sift", "liopc")          # "liopc": integer votes over their norm -- the form vl_liop emits (vl_liop.c:553-575)
            sigma = 6.0
            d = np.empty((n_feat, dim), np.float32)
            d[:n_shared] = _sift_from_base(wbase[ids], r, sigma, integer)
            d[n_shared:] = _sift_from_base(_sift_base(r, n_dis, dim), r, sigma, integer)
            if kind in ("liop", "liopc"):
                d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-12).astype(np.float32)
            if dtype == "u8":
                assert integer
                d = d.astype(np.uint8)
        wid = np.concatenate([ids, -np.ones(n_dis, np.int64)])
        perm = r.permutation(n_feat)
        descs.append(np.ascontiguousarray(d[perm])); xys.append(np.ascontiguousarray(xy[perm])); wids.append(wid[perm])

    return Scene(kind=kind, dim=dim, descs=descs, xys=xys, world_ids=wids,
                 widths=np.full(n_images, WIDTH, np.uint32), heights=np.full(n_images, HEIGHT, np.uint32))


def make_scene_torch(n_images: int, n_feat: int, seed: int = 2002, device="cuda", dim: int | None = None,
                     outlier_frac: float = 0.15, kind: str = "sift"):
    """The scenes of make_scene() ("sift": integer-valued f32 bins; "liop": the same real-valued, unit length, 144-D; "liopc": the same rounded to integer votes BEFORE the normalisation, the form vl_liop.c emits;
    "akaze": 486 random bits with 8 % flips per observation, packed LSB-first into 61 of 64 bytes), sampled with torch's
    device RNG so that bench.py can build the 200 x 8192 workloads (and the 1000-view ones) in seconds, directly in HBM.
    Every rank that calls it with the same seed on the same GPU model gets the same tensors.
    Returns (descs [N, n, dim] f32 -- uint8 for "akaze" --, xys [N, n, 2] f32, world_ids [N, n] int64) on `device`."""
    import torch

    assert kind in ("sift", "liop", "liopc", "akaze")
    if dim is None:
        dim = {"sift": 128, "liop": 144, "liopc": 144, "akaze": 64}[kind]

    g = torch.Generator(device=device); g.manual_seed(seed)
    n_shared = int(round(0.6 * n_feat)); step = max(n_shared // 4, 1)
    n_world = (n_images - 1) * step + n_shared
    n_dis = n_feat - n_shared
    f = 1.2 * WIDTH; Z0 = 10.0
    delta = (0.8 * WIDTH * Z0 / f) / n_shared
    U = lambda *s: torch.rand(*s, generator=g, device=device, dtype=torch.float64)
    Nn = lambda *s: torch.randn(*s, generator=g, device=device, dtype=torch.float64)
    wx = (torch.arange(n_world, device=device, dtype=torch.float64) + (U(n_world) * 0.6 - 0.3)) * delta
    wy = (U(n_world) * 2 - 1) * (0.4 * HEIGHT * Z0 / f)
    wz = Z0 * (0.75 + 0.5 * U(n_world))

    def sift_base(n):
        gam = torch.distributions.Gamma(torch.tensor(0.5, device=device), torch.tensor(1.0, device=device))
        # torch's gamma sampler has no generator argument: derive it from the seeded global state instead
        b = torch._standard_gamma(torch.full((n, dim), 0.5, device=device, dtype=torch.float32))
        b = b / b.norm(dim=1, keepdim=True).clamp_min(1e-12) * 512.0
        return b.clamp_(0.0, 255.0)

    torch.manual_seed(seed); torch.cuda.manual_seed(seed) if str(device).startswith("cuda") else None
    if kind == "akaze":
        wbits = torch.rand((n_world, 488), generator=g, device=device) < 0.5
        wbits[:, 486:] = False
        bitw = (2 ** torch.arange(8, device=device, dtype=torch.int32)).view(1, 1, 8)

        def pack(bits):                                   # [n, 488] bool -> [n, 64] uint8, LSB first, bytes 61..63 zero
            by = (bits.view(bits.shape[0], 61, 8).to(torch.int32) * bitw).sum(dim=2).to(torch.uint8)
            out = torch.zeros((bits.shape[0], dim), device=device, dtype=torch.uint8)
            out[:, :61] = by
            return out
    else:
        wbase = sift_base(n_world)
    descs = torch.empty((n_images, n_feat, dim), device=device, dtype=torch.uint8 if kind == "akaze" else torch.float32)
    xys = torch.empty((n_images, n_feat, 2), device=device, dtype=torch.float32)
    wids = torch.full((n_images, n_feat), -1, device=device, dtype=torch.int64)
    for i in range(n_images):
        lo = i * step
        ids = torch.arange(lo, lo + n_shared, device=device)
        cx = (lo + 0.5 * n_shared) * delta
        ang = (Nn(3) * 0.02).tolist()
        ca, sa = [np.cos(a) for a in ang], [np.sin(a) for a in ang]
        Rx = np.array([[1, 0, 0], [0, ca[0], -sa[0]], [0, sa[0], ca[0]]])
        Ry = np.array([[ca[1], 0, sa[1]], [0, 1, 0], [-sa[1], 0, ca[1]]])
        Rz = np.array([[ca[2], -sa[2], 0], [sa[2], ca[2], 0], [0, 0, 1]])
        R = torch.tensor(Rz @ Ry @ Rx, device=device, dtype=torch.float64)
        Pc = torch.stack([wx[ids] - cx, wy[ids], wz[ids]], dim=1) @ R.T
        u = f * Pc[:, 0] / Pc[:, 2] + 0.5 * WIDTH + 0.5 * Nn(n_shared)
        v = f * Pc[:, 1] / Pc[:, 2] + 0.5 * HEIGHT + 0.5 * Nn(n_shared)
        bad = U(n_shared) < outlier_frac
        u = torch.where(bad, U(n_shared) * WIDTH, u); v = torch.where(bad, U(n_shared) * HEIGHT, v)
        xy = torch.empty((n_feat, 2), device=device, dtype=torch.float64)
        xy[:n_shared, 0] = u; xy[:n_shared, 1] = v
        xy[n_shared:, 0] = U(n_dis) * WIDTH; xy[n_shared:, 1] = U(n_dis) * HEIGHT
        if kind == "akaze":
            bits = torch.empty((n_feat, 488), device=device, dtype=torch.bool)
            flips = torch.rand((n_shared, 488), generator=g, device=device) < 0.08
            bits[:n_shared] = wbits[ids] ^ flips
            bits[n_shared:] = torch.rand((n_dis, 488), generator=g, device=device) < 0.5
            bits[:, 486:] = False
            d = pack(bits)
        else:
            d = torch.empty((n_feat, dim), device=device, dtype=torch.float32)
            d[:n_shared] = wbase[ids]; d[n_shared:] = sift_base(n_dis)
            d += 6.0 * torch.randn((n_feat, dim), generator=g, device=device, dtype=torch.float32)
            d.clamp_(0.0, 255.0)
            if kind in ("sift", "liopc"):
                d.round_()
            if kind in ("liop", "liopc"):                # "liopc": integer votes over their norm, as vl_liop emits them
                d /= d.norm(dim=1, keepdim=True).clamp_min(1e-12)
        perm = torch.randperm(n_feat, generator=g, device=device)
        descs[i] = d[perm]; xys[i] = xy[perm].float()
        w = torch.full((n_feat,), -1, device=device, dtype=torch.int64); w[:n_shared] = ids
        wids[i] = w[perm]
    return descs, xys, wids


# ------------------------------------------------------------------------------------------------
# synthetic PHOTOGRAPHS (the features stage and the whole-stage bench: pixels -> matches.*.txt)
# ------------------------------------------------------------------------------------------------
def _texture(hh: int, ww: int, seed: int, device):
    """A [hh, ww] float32 texture in [0, 1] with structure at every scale from 3 px to 200 px: band-limited noise octaves
    (what a determinant-of-Hessian detector fires on) -- seeded, generated on `device` with torch."""
    import torch
    import torch.nn.functional as F

    g = torch.Generator(device=device); g.manual_seed(seed)
    img = torch.full((1, 1, hh, ww), 0.5, device=device)
    for cell, amp in ((192, 0.11), (96, 0.10), (48, 0.09), (24, 0.075), (12, 0.055), (6, 0.035), (3, 0.012)):     # ~22 k A-KAZE keypoints per 12 Mpx at threshold 0.001
        gh, gw = hh // cell + 3, ww // cell + 3
        n = torch.randn((1, 1, gh, gw), generator=g, device=device)
        up = F.interpolate(n, size=(gh * cell, gw * cell), mode="bicubic", align_corners=False)
        img = img + amp * up[:, :, cell:cell + hh, cell:cell + ww]
    return img.clamp_(0.0, 1.0)[0, 0]


def make_photo(h: int = HEIGHT, w: int = WIDTH, seed: int = 0, device="cpu") -> np.ndarray:
    """one [h, w] float32 gray image in [0, 1], 8-bit quantised (value / 255 as the reference's loader produces)"""
    import torch
    t = _texture(h, w, seed, device)
    return (torch.round(t * 255.0) * np.float32(1.0 / 255.0)).cpu().numpy().astype(np.float32)


def make_photo_set(n_images: int, h: int = HEIGHT, w: int = WIDTH, seed: int = 7, device="cpu", overlap_step: float = 0.22,
                   bgr: bool = False, noise: float = 0.004):
    """n_images photographs of ONE textured plane taken by a camera that moves along it: view i sees the window of the world
    texture starting at i * overlap_step * w (so neighbours share 78 %, views 4 steps apart 12 %, 5 apart nothing) through its
    own small homography (rotation, scale, perspective) with its own sensor noise, quantised to 8 bits -- the F / E / H filters
    have true inliers on the overlapping pairs and nothing on the others.
    Returns (images, K): images = list of torch tensors on `device`, [h, w] float32 (gray / 255) or, with bgr=True,
    [h, w, 3] uint8 (a colour cast per channel, what cv::imread would decode); K = the 3x3 pinhole matrix used (f = 1.2 w)."""
    import torch
    import torch.nn.functional as F

    step = overlap_step * w
    ww = int(w + step * (n_images - 1) + 0.2 * w) + 8
    hh = int(1.2 * h) + 8
    world = _texture(hh, ww, seed, device)[None, None]
    rng = np.random.default_rng(seed)
    f = 1.2 * w
    K = np.array([[f, 0, 0.5 * w], [0, f, 0.5 * h], [0, 0, 1.0]])
    ys, xs = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32), torch.arange(w, device=device, dtype=torch.float32), indexing="ij")
    out = []
    for i in range(n_images):
        a = rng.normal(0, 0.02); s = 1.0 + rng.normal(0, 0.03)
        px, py = rng.normal(0, 2e-6, 2)                                        # perspective terms
        tx = 0.1 * w + i * step + rng.normal(0, 0.01 * w); ty = 0.1 * h + rng.normal(0, 0.01 * h)
        ca, sa = np.cos(a) * s, np.sin(a) * s
        xc, yc = xs - 0.5 * w, ys - 0.5 * h
        den = 1.0 + px * xc + py * yc
        wxp = (ca * xc - sa * yc) / den + 0.5 * w + tx
        wyp = (sa * xc + ca * yc) / den + 0.5 * h + ty
        grid = torch.stack([wxp / (ww - 1) * 2 - 1, wyp / (hh - 1) * 2 - 1], dim=-1)[None]
        v = F.grid_sample(world, grid, mode="bilinear", padding_mode="border", align_corners=True)[0, 0]
        g = torch.Generator(device=device); g.manual_seed(seed * 1000003 + i)
        v = (v + noise * torch.randn((h, w), generator=g, device=device)).clamp_(0.0, 1.0)
        if bgr:
            cast = torch.tensor([0.96, 1.0, 0.93], device=device).view(1, 1, 3)
            out.append(torch.round(v[..., None] * cast * 255.0).to(torch.uint8).contiguous())
        else:
            out.append((torch.round(v * 255.0) * np.float32(1.0 / 255.0)).contiguous())
    return out, K
