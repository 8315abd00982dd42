"""LIOP descriptor: the one part of the path whose reference implementation compiles stand-alone
(/root/reference/src/thirdparty/liop/vl_liop.c -> oracle/_ref/libref_liop.so), so here parity IS pinned:
restatement == reference bit for bit (live, when the reference build is present) and == the committed
golden descriptors the reference produced (always)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_liop_golden_from_reference(oracle):
    z = np.load(os.path.join(GOLDEN, "liop_patches.npz"))
    d = oracle.liop_describe(z["patches"])
    assert d.shape == (24, 144)
    assert np.array_equal(d, z["ref_desc"])                    # bit-exact, ties included
    # known answers: 4! * 6 = 144 dims, non-negative, unit norm; constant patch -> zeros
    assert (d >= 0).all()
    nrm = np.linalg.norm(d.astype(np.float64), axis=1)
    assert np.allclose(np.delete(nrm, 20), 1.0, atol=1e-6) and nrm[20] == 0.0


def test_liop_live_against_reference_build(oracle):
    if oracle.ref_liop_lib() is None:
        pytest.skip("oracle/_ref/libref_liop.so not built (reference tree absent)")
    rng = np.random.default_rng(3)
    P = rng.random((300, 41, 41)).astype(np.float32)
    P[:100] = np.round(P[:100] * 32) / 32                       # many equal intensities: the sort's tie order matters
    P[100:150, 10:30, 10:30] = 0.125
    assert np.array_equal(oracle.liop_describe(P), oracle.ref_liop(P))


def test_liop_rotation_invariance_property(oracle):
    # LIOP is invariant to 90-degree rotations of the patch (sample circle starts at atan2(y, x))
    rng = np.random.default_rng(5)
    from scipy.ndimage import gaussian_filter
    P = np.stack([gaussian_filter(rng.random((41, 41)), 1.5).astype(np.float32) for _ in range(8)])
    d0 = oracle.liop_describe(P)
    d1 = oracle.liop_describe(np.ascontiguousarray(np.rot90(P, 1, axes=(1, 2))))
    assert np.allclose(d0, d1, atol=2e-2)


def test_patch_extraction_restatement_properties(oracle):
    """OpenCV is not available, so the warp+blur restatement is checked by properties: identity-like warp
    reproduces a Gaussian-blurred crop; out-of-image taps read 0; the Gaussian kernel sums to 1."""
    from scipy.ndimage import correlate1d
    rng = np.random.default_rng(2)
    img = rng.random((200, 300)).astype(np.float32)
    # size/41*factor = 1 and kp.angle = -90 -> angle 0: alpha = 1, beta = 0 -> pure translation by (x-20, y-20)
    kps = np.array([[150.0, 100.0, 41.0 / 8.0, -90.0]], np.float32)
    p = oracle.liop_extract_patches(img, kps, 8.0)[0]
    crop = img[80:121, 130:171]
    x = np.arange(11) - 5.0
    k = np.exp(-0.5 / 1.44 * x * x).astype(np.float32); k = (k * (1.0 / k.astype(np.float64).sum())).astype(np.float32)
    ref = correlate1d(correlate1d(crop.astype(np.float64), k.astype(np.float64), axis=1, mode="mirror"), k.astype(np.float64), axis=0, mode="mirror")
    assert np.allclose(p, ref, atol=2e-6)
    assert abs(float(k.sum()) - 1.0) < 1e-6
    far = oracle.liop_extract_patches(img, np.array([[-500.0, -500.0, 5.0, 0.0]], np.float32), 8.0)[0]
    assert np.all(far == 0)
