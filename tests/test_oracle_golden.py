"""
CPU tests of the oracle (oracle/tda_oracle.c via oracle/port.py, oracle/brute.py):
  * against every golden vector captured from the reference's own functions
    (tests/golden/reference_golden.npz, made by tests/golden/make_golden.py);
  * the Rips restatement against an independent brute-force boundary-matrix reduction and
    closed-form known answers (PARITY UNPINNED against the real ripser wheel: absent);
  * the Wasserstein restatement against scipy's linear_sum_assignment on persim's block matrix
    and against exhaustive search.
"""
import json

import numpy as np
import pytest

from oracle import brute, port

BANDS = ["delta", "theta", "alpha", "beta", "gamma"]


def test_corr_dist_matches_reference(golden):
    corr, dist = port.corr_dist_batch(golden["cd_windows"])
    assert np.abs(corr - golden["cd_corr"]).max() <= 1e-12       # BLAS summation order only
    off = ~np.eye(47, dtype=bool)
    far = golden["cd_dist"][:, off] > 1e-3
    assert np.abs(dist[:, off] - golden["cd_dist"][:, off])[far].max() <= 1e-12
    assert np.array_equal(dist.astype(np.float32), golden["cd_dist"].astype(np.float32))
    # zero-variance channel -> NaN -> 0 -> sqrt(2); duplicated channel -> ~0; anti-correlated -> 2
    assert np.all(dist[2, 5, np.arange(47) != 5] == np.sqrt(2.0))
    assert dist[3, 11, 7] < 1e-7 and abs(dist[3, 20, 30] - 2.0) < 1e-12
    assert np.all(np.diagonal(dist, axis1=1, axis2=2) == 0)


def test_windows_tau_takens_match_reference(golden):
    sig = golden["cw_signal"]
    n = port.lib().orc_count_windows(len(sig), 250, 62)
    assert n == golden["cw_windows"].shape[0] == 71
    assert port.lib().orc_count_windows(100, 250, 62) == golden["cw_empty_shape"][0] == 0
    for s, t in zip(golden["tau_signals"], golden["tau_values"]):
        assert port.compute_tau(s, 125) == t
    assert port.compute_tau(np.full(250, 1.5), 125) == golden["tau_const"][0]
    ramp = np.arange(250.0)
    assert port.compute_tau(ramp, 125) == golden["tau_ramp"][0]
    assert port.compute_tau(ramp) == golden["tau_ramp"][1]
    for name, s, t in zip(BANDS, golden["tau_signals"], golden["tau_values"]):
        assert np.array_equal(port.takens(s, 3, int(t), 2), golden["tk_pc_" + name])
    assert port.takens(golden["tau_signals"][0], 3, 125, 2).shape == tuple(golden["tk_empty_shape"])
    assert np.array_equal(port.takens(golden["tau_signals"][1], 3, 5, 1), golden["tk_nosub"])


def test_preprocessing_handed_to_ripser_matches_reference(golden):
    """What the recording stand-ins captured at utils.py:131 and :140."""
    kw = json.loads(str(golden["ap_kwargs"]))
    assert kw == {"maxdim": 1.0, "thresh": 2.0}
    kw = json.loads(str(golden["ep_kwargs"]))
    assert kw == {"maxdim": 1.0, "thresh": 2.0, "distance_matrix": True}
    for name in BANDS:
        pn = port.minmax_normalise(golden["tk_pc_" + name])
        assert np.array_equal(pn, golden["ap_pcnorm_" + name])
        dm = port.cloud_dm(pn)
        ref = golden["pd_" + name]                               # sklearn.pairwise_distances
        assert np.abs(dm - ref).max() < 1e-13
        assert np.array_equal(dm.astype(np.float32), ref.astype(np.float32))
    assert np.array_equal(port.minmax_normalise(golden["ap_flat_in"]), golden["ap_flat_norm"])
    assert np.array_equal(golden["ap_small_h0"], [[0, 0]]) and np.array_equal(golden["ap_small_h1"], [[0, 0]])
    f = brute.eeg_prepare(golden["ep_in"])
    assert np.array_equal(f, golden["ep_dm"].astype(np.float32))
    import ctypes as C
    g = np.empty((47, 47), np.float32)
    port.lib().orc_eeg_prepare(port._p(port._d(golden["ep_in"]), port.c_dp), 47, 1, port._p(g, port.c_fp))
    assert np.array_equal(g, f)


def test_features_bit_exact_vs_reference(golden):
    assert json.loads(str(golden["ef_keys"])) == port.FEATURE_KEYS
    for k in ["mixed", "single", "empty_finite", "zero_pers", "f32vals", "big"]:
        assert np.array_equal(port.features(golden["ef_in_" + k]), golden["ef_out_" + k]), k


def test_clean_and_exception_semantics(golden):
    assert np.array_equal(brute.clean(np.array([[0.0, 0.5], [0.0, np.inf], [0.3, 0.3]])), [[0, 0.5], [0.3, 0.3]])
    for i in range(3):
        assert golden[f"sw_a{i}"].shape[1] == 2 and golden[f"sw_b{i}"].shape[1] == 2
    assert np.array_equal(golden["sw_b0"], [[0, 0]])           # all-infinite diagram -> [[0,0]]
    assert np.array_equal(golden["sw_a1"], [[0, 0]])           # empty -> [[0,0]]
    assert np.array_equal(golden["sw_a2"], [[0, 0]])           # 1-D -> [[0,0]]
    assert np.isnan(golden["sw_exc"][0])                       # any exception -> NaN


def _dm(p):
    return np.sqrt(((p[:, None] - p[None]) ** 2).sum(-1))


def test_rips_known_answers():
    sq = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], float)
    h0, h1 = port.rips_dm(_dm(sq))
    assert np.array_equal(h0, [[0, 1], [0, 1], [0, 1], [0, np.inf]])
    assert np.array_equal(h1, np.array([[1.0, np.float32(np.sqrt(2.0))]]))
    hexa = np.array([[np.cos(k * np.pi / 3), np.sin(k * np.pi / 3)] for k in range(6)])
    h0, h1 = port.rips_dm(_dm(hexa))
    assert len(h1) == 1 and h1[0, 1] == np.float32(np.sqrt(3.0))
    h0, h1 = port.rips_dm(_dm(np.array([[0, 0], [1, 0], [0.3, 0.8]])))
    assert len(h1) == 0                                       # zero-persistence pair dropped
    h0, h1 = port.rips_dm(_dm(sq), thresh=1.2)
    assert np.array_equal(h1, [[1.0, np.inf]])                # essential H1 under a truncating threshold
    two = np.array([[0, 0], [0.1, 0], [5, 5], [5.1, 5]], float)
    h0, h1 = port.rips_dm(_dm(two), thresh=1.0)
    assert np.isinf(h0[:, 1]).sum() == 2
    dup = np.array([[0, 0], [0, 0], [1, 0], [1, 0], [0, 1]], float)
    h0, h1 = port.rips_dm(_dm(dup))
    assert len(h0) == 5 - 1 - 2 + 1                           # zero-length merges dropped
    # order conventions: H0 ascending death then inf; H1 descending birth
    rng = np.random.default_rng(0)
    d = rng.random((30, 30)); d = (d + d.T) / 2; np.fill_diagonal(d, 0)
    h0, h1 = port.rips_dm(d)
    assert np.isinf(h0[-1, 1]) and np.all(np.diff(h0[:-1, 1]) >= 0) and np.all(np.diff(h1[:, 0]) <= 0)


def test_rips_port_equals_brute_force():
    rng = np.random.default_rng(1)
    for trial in range(45):
        n = int(rng.integers(1, 28))
        if trial % 3 == 0:
            X = rng.standard_normal((n, 3)); d = _dm(X)
        elif trial % 3 == 1:
            d = rng.random((n, n)); d = (d + d.T) / 2; np.fill_diagonal(d, 0)
        else:
            d = np.round(rng.random((n, n)) * 4) / 4; d = (d + d.T) / 2; np.fill_diagonal(d, 0)   # ties
        th = 2.0 if trial % 2 == 0 else float(np.quantile(d, 0.6))
        a = port.rips_dm(d, thresh=th)
        b = brute.rips_brute(brute.eeg_prepare(d), th)
        assert np.array_equal(brute.sort_rows(a[0]), b[0]) and np.array_equal(brute.sort_rows(a[1]), b[1]), trial


def test_config1_single_window_plumbing():
    """BASELINE configs[0]: one synthetic 47x47 window on the CPU path; also the reference's
    commented-out smoke test input (v2:255-258)."""
    X = np.random.default_rng(42).standard_normal((47, 250))
    _, d = port.corr_dist(X)
    h0, h1 = port.rips_dm(d)
    assert len(h0) == 47 and np.isinf(h0[:, 1]).sum() == 1 and np.isfinite(h1).all()
    vals = set(d.astype(np.float32).astype(np.float64).ravel())
    assert all(v in vals for v in h0[:-1, 1]) and all(v in vals for v in h1.ravel())
    b0, b1 = brute.rips_brute(brute.eeg_prepare(d), 2.0)
    assert np.array_equal(brute.sort_rows(h0), b0) and np.array_equal(brute.sort_rows(h1), b1)
    t = np.random.default_rng(42).random((47, 47)); t = (t + t.T) / 2; np.fill_diagonal(t, 0)
    h0, h1 = port.rips_dm(t)
    b0, b1 = brute.rips_brute(brute.eeg_prepare(t), 2.0)
    assert np.array_equal(brute.sort_rows(h0), b0) and np.array_equal(brute.sort_rows(h1), b1)


def test_audio_persistence_port_vs_brute():
    from tda_eeg_audio_amd import synth
    wins = synth.audio_windows(3, "delta", seed=5)
    for w in wins:
        tau = port.compute_tau(w, 125)
        (h0, h1), P = port.audio_persistence(w, tau)
        pn = port.minmax_normalise(port.takens(w, 3, tau, 2))
        dm = port.cloud_dm(pn).astype(np.float32)
        if P <= 60:
            b0, b1 = brute.rips_brute(dm, 2.0)
            assert np.array_equal(brute.sort_rows(h0), b0) and np.array_equal(brute.sort_rows(h1), b1)
    (h0, h1), P = port.audio_persistence(wins[0], 124)
    assert P == 1 and np.array_equal(h0, [[0, 0]]) and np.array_equal(h1, [[0, 0]])


def test_wasserstein_port_vs_persim_restatement():
    rng = np.random.default_rng(2)
    assert abs(port.wasserstein([[0, 1]], [[0, 2]]) - 1.0) < 1e-12
    assert port.wasserstein([[0, 1]], [[0, 1]]) == 0.0
    for _ in range(40):
        A = np.sort(rng.random((int(rng.integers(1, 50)), 2)), axis=1)
        B = np.sort(rng.random((int(rng.integers(1, 50)), 2)), axis=1)
        w = port.wasserstein(A, B)
        assert abs(w - brute.wasserstein_persim(A, B)) < 1e-10
        assert abs(w - port.wasserstein(B, A)) < 1e-10
    A = np.sort(rng.random((7, 2)), axis=1)
    assert abs(port.wasserstein(A, np.zeros((1, 2))) - ((A[:, 1] - A[:, 0]) / np.sqrt(2)).sum()) < 1e-12
    for _ in range(10):
        A = np.sort(rng.random((int(rng.integers(1, 5)), 2)), axis=1)
        B = np.sort(rng.random((int(rng.integers(1, 5)), 2)), axis=1)
        assert abs(port.wasserstein(A, B) - brute.wasserstein_bruteforce(A, B)) < 1e-9


def _single_linkage_deaths(dm64):
    """H0 deaths by a library that is neither ours nor ripser: scipy's single-linkage merge heights on the float32
    distances ripser would see (heights are selected distances, never computed: exact)."""
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import squareform
    d = dm64.astype(np.float32).astype(np.float64)
    d = (d + d.T) / 2.0 if not np.array_equal(d, d.T) else d
    np.fill_diagonal(d, 0.0)
    return np.sort(linkage(squareform(d, checks=False), method="single")[:, 2])


def test_h0_equals_scipy_single_linkage():
    """H0 of a Vietoris-Rips filtration is single-linkage clustering: finite deaths = merge heights, zero heights
    (coincident points) dropped, one essential class.  Pins the oracle's H0 -- float32 cast, threshold, zero rows,
    row order -- against scipy on EEG-like matrices, white noise (near-tied lengths), and Takens clouds of all bands."""
    from tda_eeg_audio_amd import synth
    for kind in ("latent", "white"):
        W = synth.eeg_windows(12, seed=31, kind=kind) if kind == "white" else synth.eeg_windows(12, seed=31)
        for w in W:
            _, d = port.corr_dist(w)
            h0 = port.rips_dm(d)[0]
            ref = _single_linkage_deaths((d + d.T) / 2.0)
            ref = ref[ref > 0]
            assert np.array_equal(h0[:-1, 1], ref) and np.all(h0[:, 0] == 0) and np.isinf(h0[-1, 1])
    wins, _ = synth.audio_windows_all_bands(6, seed=5)
    for w in wins:
        tau = port.compute_tau(w, 125)
        (h0, _), P = port.audio_persistence(w, tau)
        pc = port.minmax_normalise(port.takens(w, 3, tau, 2))
        ref = _single_linkage_deaths(port.cloud_dm(pc))
        ref = ref[ref > 0]
        assert len(pc) == P and np.array_equal(h0[:-1, 1], ref) and np.isinf(h0[-1, 1])
    # coincident points: zero-height merges give no row
    pts = np.array([[0.0, 0.0], [0.0, 0.0], [1.0, 0.0], [1.0, 0.0], [0.0, 2.0]])
    dm = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    h0 = port.rips_dm(dm, thresh=10.0)[0]
    assert np.array_equal(h0[:-1, 1], [1.0, 2.0]) and np.isinf(h0[-1, 1])


def _polygon(n):
    ang = 2.0 * np.pi * np.arange(n) / n
    return np.stack([np.cos(ang), np.sin(ang)], axis=1)


def polygon_h1(n):
    """Vietoris-Rips of n evenly spaced points on the unit circle (Adamaszek & Adams, "The Vietoris-Rips complexes of
    a circle", 2017): the complex with all chords of up to k steps is homotopy equivalent to a circle while k < n/3
    and has no H1 from k >= n/3 on -- ONE class, born at the side 2 sin(pi/n), dying at the chord of ceil(n/3) steps."""
    import math
    return 2.0 * math.sin(math.pi / n), 2.0 * math.sin(math.pi * math.ceil(n / 3) / n)


def test_h1_of_regular_polygons_matches_the_theorem():
    """A known answer that is not ours for H1, with as many tied lengths as a metric can have (n chords per length):
    every n from 4 to 60 gives exactly one H1 row at the theorem's values (float32 of the float64 chords), and
    n - 1 finite H0 rows at the side length."""
    for n in range(4, 61):
        P = _polygon(n)
        dm = np.sqrt(((P[:, None] - P[None]) ** 2).sum(-1))
        h0, h1 = port.rips_dm(dm, thresh=10.0)
        b, d = polygon_h1(n)
        assert h1.shape == (1, 2), n
        assert abs(h1[0, 0] - b) < 2e-7 * b and abs(h1[0, 1] - d) < 2e-7 * d, (n, h1, b, d)
        assert len(h0) == n and np.all(np.abs(h0[:-1, 1] - b) < 2e-7 * b) and np.isinf(h0[-1, 1])
        # under a threshold between birth and death the class is essential
        h0t, h1t = port.rips_dm(dm, thresh=(b + d) / 2.0)
        assert h1t.shape == (1, 2) and abs(h1t[0, 0] - b) < 2e-7 * b and np.isinf(h1t[0, 1]), n


def lattice(k, dim=2):
    import itertools
    return np.array(list(itertools.product(range(k), repeat=dim)), dtype=np.float64)


def test_h1_of_lattices_cube_and_cross_polytope():
    """More answers that follow from the geometry alone: a k x k unit lattice has (k-1)^2 independent unit squares,
    each filled when its diagonals arrive -- (k-1)^2 rows (1, sqrt 2), all alive at once (up to 100 classes); the
    unit cube has 5 independent faces; the cross-polytope's edges all arrive at sqrt 2 together with the triangles
    that fill them: no H1 row."""
    def dm(P):
        return np.sqrt(((P[:, None] - P[None]) ** 2).sum(-1))
    r2 = np.float64(np.float32(np.sqrt(2.0)))
    for k in range(2, 12):
        h0, h1 = port.rips_dm(dm(lattice(k)), thresh=100.0)
        assert h1.shape == ((k - 1) ** 2, 2) and np.all(h1[:, 0] == 1.0) and np.all(h1[:, 1] == r2), k
        assert len(h0) == k * k and np.all(h0[:-1, 1] == 1.0)
    h0, h1 = port.rips_dm(dm(lattice(2, 3)), thresh=100.0)
    assert h1.shape == (5, 2) and np.all(h1[:, 0] == 1.0) and np.all(h1[:, 1] == r2)
    for d in (3, 4, 5):
        cross = np.concatenate([np.eye(d), -np.eye(d)])
        assert len(port.rips_dm(dm(cross), thresh=100.0)[1]) == 0


def far_polygons():
    """Three regular polygons of different size far from each other (closer than the threshold, farther than any of
    their deaths): the filtration below the gap is a disjoint union, so H1 is the three classes of the parts."""
    parts = [(7, 1.0, (0.0, 0.0)), (12, 1.5, (10.0, 0.0)), (20, 0.8, (0.0, 10.0))]
    pts = np.concatenate([r * _polygon(n) + np.array(c) for n, r, c in parts])
    exp = sorted([(r * polygon_h1(n)[0], r * polygon_h1(n)[1]) for n, r, _ in parts], reverse=True)
    return pts, np.array(exp), parts


def test_h1_of_a_disjoint_union_is_the_union():
    pts, exp, parts = far_polygons()
    dm = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    h0, h1 = port.rips_dm(dm, thresh=100.0)
    assert h1.shape == (3, 2) and np.all(np.abs(h1 - exp) < 3e-7 * exp), (h1, exp)     # rows come in descending birth
    n_tot = len(pts)
    sides = np.sort(np.concatenate([np.full(n - 1, np.float32(np.float32(r) * 0 + 2.0 * r * np.sin(np.pi / n))) for n, r, _ in parts]))
    assert len(h0) == n_tot and np.isinf(h0[-1, 1])
    assert np.all(np.abs(np.sort(h0[:n_tot - 3, 1]) - sides) < 3e-7 * sides)          # the sides of the three polygons
    assert np.all(h0[n_tot - 3:n_tot - 1, 1] > 6.0)                                    # and the two gaps between them
