"""The REAL-valued matching path pinned by reference-built code (VERDICT r2 "weak" 1): tests/golden/liop_match_ref.npz holds two
views of 8,192 LIOP-144 descriptors computed by the reference's own vl_liop.c and the 3-NN of every row of view 1 among view 0
computed by the reference's own hnswlib::BruteforceSearch (generator: tools/make_golden_liop_match.py; data only).

hnswlib's AVX L2Sqr sums the 144 squared differences lane-wise, OpenMVG's L2 (what the oracle restates and the
GPU reproduces bit for bit) four at a time from left to right -- two float sums of the same 144 terms.  Each differs from the real
sum d by at most (n + 2) u d (n = 144 additions + the two roundings of a term, u = 2^-24), so

    TOL(d) = 2 (144 + 2) 2^-24 d          (1.74e-5 d; the differences observed are far smaller and are printed)

bounds the difference between the two, and the comparison rules are those of SURVEY.md section 8(a)-note:
  * distances: |d_k(ours) - d_k(reference)| <= TOL(d_k) on EVERY row, k = 1, 2;
  * indices: equal on every row whose reference d1 / d2 / d3 are farther apart than TOL (rows inside it are counted and printed);
  * ratio verdict d1 < 0.36 d2: equal on every row with |d1 - 0.36 d2| > TOL(d1) + 0.36 TOL(d2).
The same three checks run against the CPU restatement (here, `-m "not gpu"`), the GPU default path (f32 MFMA tiles + f32 re-score +
certification) and the opt-in split-f16 nominator (`-m gpu`)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
U = 2.0 ** -24


def _load():
    z = np.load(os.path.join(G, "liop_match_ref.npz"))
    A = (z["hist0"].astype(np.float32) / z["norm0"][:, None]).astype(np.float32)       # desc[i] /= norm, vl_liop.c:572-574
    B = (z["hist1"].astype(np.float32) / z["norm1"][:, None]).astype(np.float32)
    return A, B, z["ref_idx"], z["ref_dist"].astype(np.float64)


def _tol(d):
    return 2.0 * (144 + 2) * U * d


def _check(idx, dist, ref_idx, ref_dist, who):
    dist = dist.astype(np.float64)
    err = np.abs(dist - ref_dist[:, :2])
    assert (err <= _tol(ref_dist[:, :2])).all(), f"{who}: distance beyond the accumulation bound, max {err.max():.3e}"
    ulp = np.spacing(ref_dist[:, :2].astype(np.float32)).astype(np.float64)
    gap12 = ref_dist[:, 1] - ref_dist[:, 0]; gap23 = ref_dist[:, 2] - ref_dist[:, 1]
    clear1 = gap12 > _tol(ref_dist[:, 0]) + _tol(ref_dist[:, 1])
    clear2 = clear1 & (gap23 > _tol(ref_dist[:, 1]) + _tol(ref_dist[:, 2]))
    assert np.array_equal(idx[clear1, 0], ref_idx[clear1, 0]), f"{who}: nearest row differs from the reference-built search"
    assert np.array_equal(idx[clear2, 1], ref_idx[clear2, 1]), f"{who}: second row differs from the reference-built search"
    R = np.float64(np.float32(0.36))
    margin = np.abs(ref_dist[:, 0] - R * ref_dist[:, 1])
    decided = margin > _tol(ref_dist[:, 0]) + R * _tol(ref_dist[:, 1])
    ours = dist[:, 0].astype(np.float32) < np.float32(0.36) * dist[:, 1].astype(np.float32)
    ref = ref_dist[:, 0].astype(np.float32) < np.float32(0.36) * ref_dist[:, 1].astype(np.float32)
    assert np.array_equal(ours[decided], ref[decided]), f"{who}: ratio verdict differs from the reference-built search"
    print(f"{who}: rows {len(idx)}, excluded from the index check {int((~clear1).sum())} (1st) / {int((~clear2).sum())} (2nd), "
          f"from the verdict check {int((~decided).sum())}; index-equal anyway on {int((idx[:, 0] == ref_idx[:, 0]).sum())} / "
          f"{int((idx[:, 1] == ref_idx[:, 1]).sum())} rows; max |d - d_ref| = {float((err / ulp).max()):.1f} ulp; "
          f"matches {int(ours.sum())} (reference {int(ref.sum())})")
    return ours


def test_fixture_is_what_the_reference_routine_emits():
    """unit-norm, non-negative LIOP rows rebuilt from (histogram, norm); the reference's 3-NN ascending"""
    A, B, ri, rd = _load()
    assert A.shape == B.shape == (8192, 144) and ri.shape == rd.shape == (8192, 3)
    for D in (A, B):
        assert (D >= 0).all() and np.allclose(np.linalg.norm(D.astype(np.float64), axis=1), 1.0, atol=1e-6)
    assert (np.diff(rd, axis=1) >= 0).all() and ri.min() >= 0 and ri.max() < 8192


def test_cpu_restatement_against_the_reference_built_search(oracle):
    A, B, ri, rd = _load()
    idx, dist = oracle.knn2(A, B)
    ours = _check(idx, dist, ri, rd, "oracle")
    # the restatement's matcher (ratio test + de-duplication) reports exactly the rows its own 2-NN passes
    m = oracle.match_distance_ratio(A, B, 0.6, True)
    assert np.array_equal(np.sort(m[:, 1]), np.flatnonzero(ours))
    if oracle.ref_lib() is not None:        # authoring container: the live reference build agrees with the committed arrays
        li, ld = oracle.ref_knn(A, B, 3)
        assert np.array_equal(li, ri) and np.array_equal(ld.astype(np.float64), rd)


@pytest.mark.gpu
@pytest.mark.parametrize("split", [False, True], ids=["f32_tiles", "split_f16"])
def test_gpu_paths_against_the_reference_built_search(ctx, oracle, split):
    A, B, ri, rd = _load()
    ctx.set_split_mfma(split)
    try:
        idx, dist = ctx.knn2(A, B)
        s = ctx.stats()
        assert s.n_split_mfma == int(split) and s.n_integer_mfma == 0
        ours = _check(idx, dist, ri, rd, "gpu split-f16" if split else "gpu f32 tiles")
        # ... and bit for bit what the CPU restatement returns (the parity bar of the path itself)
        oi, od = oracle.knn2(A, B)
        assert np.array_equal(dist, od) and np.array_equal(idx, oi)
        # the matcher proper: two registered views, ratio 0.6 squared -> the rows the reference-built distances pass
        ctx.clear_images()
        xy = np.zeros((8192, 2), np.float32); xy[:, 0] = np.arange(8192)
        ctx.set_image(0, A, xy, 4000, 3000); ctx.set_image(1, B, xy, 4000, 3000)
        g = ctx.match_pairs(np.array([[0, 1]], np.uint32), 0.6, True)
        assert np.array_equal(np.sort(g.matches[:, 1]), np.flatnonzero(ours))
        assert np.array_equal(g.matches[np.argsort(g.matches[:, 1]), 0], idx[ours, 0])
    finally:
        ctx.set_split_mfma(False)
        ctx.clear_images()
