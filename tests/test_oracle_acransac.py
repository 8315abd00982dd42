"""Pins the CPU restatement of the AC-RANSAC F-matrix filter (oracle/acransac.c) -- no GPU needed.

Known answers from SURVEY.md A.8 (6)-(8) plus numpy second opinions for the pieces whose exact
OpenMVG source is not in /root/reference: 7-point solver vs an SVD null space + numpy.roots,
cubic solver vs numpy.roots, log-combinatorial tables vs scipy, sample stream vs golden values.
"""
import os

import numpy as np
import pytest
from scipy.special import gammaln

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
W, H = 4000, 3000


def two_view(n_in, n_out, sigma, seed):
    """pinhole pair with known F: x2^T F x1 = 0"""
    rng = np.random.default_rng(seed)
    f = 1.2 * W
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]])
    ang = 0.08
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([1.0, 0.1, 0.05])
    X = np.stack([rng.uniform(-4, 4, n_in), rng.uniform(-3, 3, n_in), rng.uniform(8, 14, n_in)], 1)
    p1 = (K @ X.T).T; p1 = p1[:, :2] / p1[:, 2:]
    X2 = (R @ X.T).T + t
    p2 = (K @ X2.T).T; p2 = p2[:, :2] / p2[:, 2:]
    p1 += rng.normal(0, sigma, p1.shape); p2 += rng.normal(0, sigma, p2.shape)
    o1 = np.stack([rng.uniform(0, W, n_out), rng.uniform(0, H, n_out)], 1)
    o2 = np.stack([rng.uniform(0, W, n_out), rng.uniform(0, H, n_out)], 1)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
    return np.concatenate([p1, o1]), np.concatenate([p2, o2]), F


def test_rng_stream_golden(oracle):
    z = np.load(os.path.join(GOLDEN, "rng_stream.npz"))
    args = [(0, 1, 0, 0), (0, 1, 0, 1), (0, 1, 1, 0), (3, 5, 2, 0), (199, 7, 2047, 6), (4294967295, 0, 0, 0)]
    got = np.array([oracle.lib().orc_rng_u64(5489, *a) for a in args], np.uint64)
    assert np.array_equal(got, z["u64"])
    smp = np.stack([oracle.sample7(5489, 3, 5, it, np.arange(100, dtype=np.uint32)) for it in range(4)])
    assert np.array_equal(smp, z["sample7_pool100"])


def test_rng_python_restatement(oracle):
    # independent restatement of the integer recipe (splitmix64 finaliser, two rounds)
    M = (1 << 64) - 1
    def mix(z):
        z ^= z >> 30; z = (z * 0xbf58476d1ce4e5b9) & M
        z ^= z >> 27; z = (z * 0x94d049bb133111eb) & M
        return z ^ (z >> 31)
    G = 0x9E3779B97F4A7C15
    for (seed, I, J, it, at) in [(5489, 0, 1, 0, 0), (1, 17, 4, 999, 3), (2**63 + 5, 2**32 - 1, 2**32 - 1, 2**32 - 1, 7)]:
        a = mix((seed + G * (1 + ((I << 32) | J))) & M)
        exp = mix((a + G * (1 + ((it << 32) | at))) & M)
        assert oracle.lib().orc_rng_u64(seed, I, J, it, at) == exp


def test_sample7_distinct_and_from_pool(oracle):
    pool = np.array([5, 9, 11, 20, 21, 22, 23, 40], np.uint32)       # barely more than 7: many rejections
    for it in range(50):
        s = oracle.sample7(7, 1, 2, it, pool)
        assert len(set(s.tolist())) == 7 and set(s.tolist()) <= set(pool.tolist())


def test_cubic_against_numpy_roots(oracle):
    rng = np.random.default_rng(0)
    for _ in range(300):
        c = rng.normal(size=4)
        r = oracle.solve_cubic(c)
        ref = np.roots(c[::-1]); ref = np.sort(ref[np.abs(ref.imag) < 1e-10].real)
        assert len(r) == len(ref)
        assert np.allclose(np.sort(r), ref, rtol=1e-8, atol=1e-10)
    assert oracle.solve_cubic([0, 0, 0, 1]).tolist() == [0.0, 0.0, 0.0]       # triple root
    assert len(oracle.solve_cubic([1, 2, 3, 0])) == 0                          # not a cubic: no model


def test_seven_point_against_svd_nullspace(oracle):
    rng = np.random.default_rng(1)
    x1, x2, _ = two_view(400, 0, 0.0, 5)
    s = 1 / np.sqrt(W * H)
    n1 = x1 * s - 0.5 * np.array([W, H]) * s; n2 = x2 * s - 0.5 * np.array([W, H]) * s
    for _ in range(200):
        idx = rng.choice(400, 7, replace=False)
        Fo = oracle.seven_point(n1[idx], n2[idx])
        a, b = n1[idx], n2[idx]
        A = np.stack([b[:, 0] * a[:, 0], b[:, 0] * a[:, 1], b[:, 0], b[:, 1] * a[:, 0], b[:, 1] * a[:, 1], b[:, 1],
                      a[:, 0], a[:, 1], np.ones(7)], 1)
        Vt = np.linalg.svd(A)[2]
        F1, F2 = Vt[-1].reshape(3, 3), Vt[-2].reshape(3, 3)
        ls = np.array([-1.0, 0.0, 1.0, 2.0])
        c = np.polyfit(ls, [np.linalg.det(F1 + l * F2) for l in ls], 3)
        rr = np.roots(c); rr = rr[np.abs(rr.imag) < 1e-9].real
        Fn = [F1 + l * F2 for l in rr]
        assert len(Fo) == len(Fn) and len(Fo) in (1, 3)
        for F in Fo:
            assert abs(np.linalg.det(F)) < 1e-9 * np.linalg.norm(F) ** 3 + 1e-14
            for k in range(7):                                                # epipolar constraint of the sample
                assert abs(np.append(b[k], 1) @ F @ np.append(a[k], 1)) < 1e-9 * np.linalg.norm(F)
            f = F.ravel() / np.linalg.norm(F)
            best = min(min(np.linalg.norm(f - g.ravel() / np.linalg.norm(g)), np.linalg.norm(f + g.ravel() / np.linalg.norm(g))) for g in Fn)
            assert best < 1e-6


def test_symmetric_epipolar_error_formula(oracle):
    rng = np.random.default_rng(2)
    F = rng.normal(size=(3, 3)); x = rng.normal(size=2); y = rng.normal(size=2)
    Fx = F @ np.append(x, 1); Fty = F.T @ np.append(y, 1)
    exp = (np.append(y, 1) @ Fx) ** 2 * (1 / (Fx[0] ** 2 + Fx[1] ** 2) + 1 / (Fty[0] ** 2 + Fty[1] ** 2)) / 4
    got = oracle.lib().orc_sym_epipolar_err(np.ascontiguousarray(F).ctypes.data, x[0], x[1], y[0], y[1])
    assert got == pytest.approx(exp, rel=1e-14)


def test_logcombi_tables(oracle):
    n = 500
    a, b = oracle.logcombi_tables(n, 7)
    k = np.arange(n + 1)
    ref_n = (gammaln(n + 1) - gammaln(k + 1) - gammaln(n - k + 1)) / np.log(10)
    ref_n[0] = 0; ref_n[n] = 0
    assert np.allclose(a, ref_n, rtol=2e-5, atol=2e-4)
    nn = np.arange(n + 1).astype(float)
    ref_k = np.where(nn > 7, (gammaln(nn + 1) - gammaln(8) - gammaln(np.maximum(nn - 7, 0) + 1)) / np.log(10), 0.0)
    assert np.allclose(b, ref_k, rtol=2e-5, atol=2e-4)
    # the float loop of makelogcombi_n restated in numpy float32
    l10 = np.log10(np.arange(n + 1, dtype=np.float32), where=np.arange(n + 1) > 0, out=np.full(n + 1, -np.inf, np.float32)).astype(np.float32)
    for kk in (1, 2, 7, 100, 250, 251, 400, 499):
        r = np.float32(0); kq = min(kk, n - kk)
        for i in range(1, kq + 1):
            r = np.float32(r + np.float32(l10[n - i + 1] - l10[i]))
        assert abs(float(a[kk]) - float(r)) <= 4e-6 * max(1.0, abs(float(r)))   # numpy's log10f may differ from glibc's by an ulp


def test_acransac_known_scene(oracle):
    x1, x2, Ft = two_view(200, 200, 0.3, 11)
    inl, fr = oracle.acransac_F(x1, x2, W, H, W, H, 4.0, 2048, seed=5489, I=0, J=1)
    assert fr.accepted == 1
    inl = set(inl.tolist())
    assert len(inl & set(range(200))) >= 0.95 * 200          # >= 95 % of the true inliers
    assert len(inl - set(range(200))) <= 0.02 * 200 + 2       # <= ~2 % outliers slip in
    assert 0 < fr.threshold < 4.0
    F = np.array(fr.F).reshape(3, 3); F /= np.linalg.norm(F); Ft = Ft / np.linalg.norm(Ft)
    # AC-RANSAC returns the best MINIMAL-sample model (no refit): close to the truth in Frobenius terms
    # and as good as the true F on the true inliers (median symmetric epipolar distance, pixels)
    assert min(np.linalg.norm(F - Ft), np.linalg.norm(F + Ft)) < 0.1

    def med_err(Fm):
        a = np.c_[x1[:200], np.ones(200)]; b = np.c_[x2[:200], np.ones(200)]
        Fa = a @ Fm.T; Fb = b @ Fm
        num = np.sum(b * Fa, 1) ** 2
        return np.median(np.sqrt(num * (1 / (Fa[:, 0] ** 2 + Fa[:, 1] ** 2) + 1 / (Fb[:, 0] ** 2 + Fb[:, 1] ** 2)) / 4))
    assert med_err(F) < 2.0 * med_err(Ft) + 0.1
    assert fr.nfa < 0 and fr.n_iter <= 2048 + 204


def test_acransac_rejects_pure_outliers_and_small_sets(oracle):
    rng = np.random.default_rng(3)
    x1 = np.stack([rng.uniform(0, W, 60), rng.uniform(0, H, 60)], 1); x2 = np.stack([rng.uniform(0, W, 60), rng.uniform(0, H, 60)], 1)
    inl, fr = oracle.acransac_F(x1, x2, W, H, W, H)
    assert fr.accepted == 0 and (len(inl) == 0 or len(inl) <= 17)
    assert fr.n_iter >= 2048 - 204                                 # the reserve is only released after a success
    inl, fr = oracle.acransac_F(x1[:7], x2[:7], W, H, W, H)       # n <= 7 -> returns at once
    assert len(inl) == 0 and fr.accepted == 0 and fr.n_iter == 0


def test_acransac_is_deterministic_and_keyed_by_pair(oracle):
    x1, x2, _ = two_view(120, 80, 0.4, 4)
    a, fa = oracle.acransac_F(x1, x2, W, H, W, H, I=3, J=9)
    b, fb = oracle.acransac_F(x1, x2, W, H, W, H, I=3, J=9)
    c, fc = oracle.acransac_F(x1, x2, W, H, W, H, I=4, J=9)
    assert np.array_equal(a, b) and fa.nfa == fb.nfa
    assert (fa.n_models != fc.n_models) or (fa.nfa != fc.nfa) or not np.array_equal(a, c)   # a different stream
    assert set(a.tolist()) & set(range(120))


def test_filter_collection_acceptance_rule(oracle):
    # 18 inliers pass (> 17.5), a pair with only 12 matches cannot
    x1, x2, _ = two_view(40, 0, 0.2, 6)
    xy = [x1.astype(np.float32), x2.astype(np.float32)]
    pairs = np.array([[0, 1]], np.uint32)
    m = np.stack([np.arange(40), np.arange(40)], 1).astype(np.uint32)
    oc, om = oracle.filter_F_collection(xy, [W, W], [H, H], pairs, [40], m)
    assert oc[0] > 17 and set(map(tuple, om.tolist())) <= set(map(tuple, m.tolist()))
    oc, om = oracle.filter_F_collection(xy, [W, W], [H, H], pairs, [12], m[:12])
    assert oc[0] == 0 and len(om) == 0


# ---------------------------------------------------------------- homography variant

def _h_scene(n_in, n_out, sigma, seed):
    rng = np.random.default_rng(seed)
    Ht = np.array([[1.02, 0.03, 40.0], [-0.02, 0.98, -25.0], [1e-5, -2e-5, 1.0]])
    x1 = np.stack([rng.uniform(0, W, n_in + n_out), rng.uniform(0, H, n_in + n_out)], 1)
    x2h = (Ht @ np.c_[x1, np.ones(len(x1))].T).T
    x2 = x2h[:, :2] / x2h[:, 2:]
    x2[:n_in] += rng.normal(0, sigma, (n_in, 2))
    x2[n_in:] = np.stack([rng.uniform(0, W, n_out), rng.uniform(0, H, n_out)], 1)
    return x1, x2, Ht


def test_four_point_dlt_exact(oracle):
    x1, x2, Ht = _h_scene(4, 0, 0.0, 1)
    Hh = oracle.four_point_h(x1, x2)
    # raw pixel coordinates (no pre-conditioning in this direct call): DLT is exact up to its conditioning
    assert np.allclose(Hh / Hh[2, 2], Ht, rtol=1e-5, atol=1e-6)
    x2h = (Hh @ np.c_[x1, np.ones(4)].T).T
    assert np.allclose(x2h[:, :2] / x2h[:, 2:], x2, atol=1e-6)


def test_acransac_H_known_scene(oracle):
    x1, x2, Ht = _h_scene(200, 100, 0.4, 2)
    inl, fr = oracle.acransac_H(x1, x2, W, H, W, H)
    assert fr.accepted == 1 and fr.nfa < 0 and 0 < fr.threshold < 4.0
    s = set(inl.tolist())
    assert len(s & set(range(200))) >= 190 and len(s - set(range(200))) <= 3
    Hh = np.array(fr.F).reshape(3, 3); Hh /= Hh[2, 2]
    assert np.allclose(Hh[:2, :2], Ht[:2, :2], atol=5e-3) and np.allclose(Hh[:2, 2], Ht[:2, 2], atol=3.0)
    # n <= 4 returns at once; pure outliers are rejected; acceptance needs > 2.5 * 4 inliers
    assert len(oracle.acransac_H(x1[:4], x2[:4], W, H, W, H)[0]) == 0
    _, fo = oracle.acransac_H(x1[200:260], x2[200:260], W, H, W, H)
    assert fo.accepted == 0
