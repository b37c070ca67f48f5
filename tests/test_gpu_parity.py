"""
GPU parity tests: the HIP path, called through the C ABI (libtdaeeg.so), against the CPU
oracle (oracle/tda_oracle.c + oracle/brute.py) on the same seeded inputs and against the
golden vectors captured from the reference (tests/golden/reference_golden.npz).

Bars: bit-exact for H0/H1 (birth, death) pairs (compared as sorted multisets of float32-exact
values) and for every float64 stage whose operation order the oracle fixes (corr/dist, tau);
Wasserstein within 1e-6 absolute (north_star), observed ~1e-12; features within 1e-12 relative.
"""
import numpy as np
import pytest

from oracle import brute, port
from tda_eeg_audio_amd import engine, synth

pytestmark = pytest.mark.gpu


def _same_multiset(a, b):
    return np.array_equal(brute.sort_rows(a), brute.sort_rows(b))


# ------------------------------------------------------------------ corr -> dist
def test_corr_dist_bit_exact_vs_oracle(ctx):
    W = synth.eeg_windows(37, seed=7)
    W[3, 5] = 1.25                       # zero-variance channel
    W[4, 11] = W[4, 7]                   # duplicated channel
    corr, dist = engine.corr_dist_batch(W, ctx=ctx)
    oc, od = port.corr_dist_batch(W)
    assert np.array_equal(dist, od)
    assert np.array_equal(corr, oc)
    assert np.all(dist[3, 5, np.arange(47) != 5] == np.sqrt(2.0))


def test_corr_dist_vs_reference_golden(ctx, golden):
    corr, dist = engine.corr_dist_batch(golden["cd_windows"], ctx=ctx)
    assert np.abs(corr - golden["cd_corr"]).max() <= 1e-12
    assert np.abs(dist - golden["cd_dist"]).max() <= 1e-7      # sqrt near r=1 amplifies 1e-16 -> 1e-8
    off = ~np.eye(47, dtype=bool)
    far = golden["cd_dist"][:, off] > 1e-3
    assert np.abs(dist[:, off] - golden["cd_dist"][:, off])[far].max() <= 1e-12
    # float32 cast (what ripser consumes) identical
    assert np.array_equal(dist.astype(np.float32), golden["cd_dist"].astype(np.float32))


def test_corr_dist_sliding_equals_stacked_windows(ctx):
    """Sliding windows read in place from the recording == create_sliding_windows + per-window loop."""
    rng = np.random.default_rng(4)
    for n_s in (4606, 2663, 250, 249, 311):
        sig = rng.standard_normal((47, n_s)) + 0.3 * rng.standard_normal((1, n_s))
        corr, dist = engine.corr_dist_sliding(sig, 250, 62, want_corr=True, ctx=ctx)
        n_win = (n_s - 250) // 62 + 1 if n_s >= 250 else 0
        assert dist.shape == (n_win, 47, 47)
        if n_win:
            stack = np.stack([sig[:, i * 62:i * 62 + 250] for i in range(n_win)])
            oc, od = port.corr_dist_batch(stack)
            assert np.array_equal(dist, od) and np.array_equal(corr, oc)
    assert (4606 - 250) // 62 + 1 == 71          # results/preprocessing_metadata.csv:2


def test_corr_dist_other_shapes(ctx):
    rng = np.random.default_rng(3)
    for n_ch, n_t in [(2, 5), (8, 64), (47, 500), (64, 100), (33, 51)]:
        W = rng.standard_normal((3, n_ch, n_t))
        corr, dist = engine.corr_dist_batch(W, ctx=ctx)
        oc, od = port.corr_dist_batch(W)
        assert np.array_equal(dist, od) and np.array_equal(corr, oc)


# ------------------------------------------------------------------ Rips from distance matrices
def test_h1_of_regular_polygons_matches_the_theorem(ctx):
    """Known answer from the literature (Adamaszek & Adams 2017, see tests/test_oracle_golden.py::polygon_h1): n evenly
    spaced points on a circle have ONE H1 class, born at the side, dying at the chord of ceil(n/3) steps -- through the
    distance-matrix kernel (n = 4..128: both vertex-word widths), the point-cloud kernel (P = 4..128, coordinates as
    given) and, under a threshold between birth and death, as an essential class.  n chords share every length: the
    heaviest ties a metric offers."""
    from test_oracle_golden import _polygon, polygon_h1
    for n in list(range(4, 49)) + [63, 64, 65, 96, 127, 128]:
        P = _polygon(n)
        dm = np.sqrt(((P[:, None] - P[None]) ** 2).sum(-1))
        b, d = polygon_h1(n)
        h0, h1, st = engine.rips_dm_batch(np.stack([dm, dm]), thresh=10.0, ctx=ctx)
        assert not st.any(), (n, st)
        for w in range(2):
            assert h1[w].shape == (1, 2), (n, h1[w])
            assert abs(h1[w][0, 0] - b) < 2e-7 * b and abs(h1[w][0, 1] - d) < 2e-7 * d, (n, h1[w])
            assert len(h0[w]) == n and np.all(np.abs(h0[w][:-1, 1] - b) < 2e-7 * b) and np.isinf(h0[w][-1, 1])
        h0, h1, st = engine.rips_dm_batch(dm[None], thresh=(b + d) / 2.0, ctx=ctx)
        assert not st.any() and h1[0].shape == (1, 2) and abs(h1[0][0, 0] - b) < 2e-7 * b and np.isinf(h1[0][0, 1]), n
        c0, c1, st = engine.cloud_rips_batch(P[None], normalise=False, thresh=10.0, ctx=ctx)
        assert not st.any() and c1[0].shape == (1, 2), (n, c1[0])
        assert abs(c1[0][0, 0] - b) < 2e-7 * b and abs(c1[0][0, 1] - d) < 2e-7 * d, (n, c1[0])


def test_h1_of_a_disjoint_union_is_the_union(ctx):
    """Three regular polygons of different size far apart (39 points): three classes with three different births and
    deaths alive together, each at its part's theorem values, through both kernels."""
    from test_oracle_golden import far_polygons
    pts, exp, _ = far_polygons()
    dm = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    h0, h1, st = engine.rips_dm_batch(dm[None], thresh=100.0, ctx=ctx)
    assert st[0] == 0 and h1[0].shape == (3, 2) and np.all(np.abs(h1[0] - exp) < 3e-7 * exp), h1[0]
    c0, c1, st = engine.cloud_rips_batch(pts[None], normalise=False, thresh=100.0, ctx=ctx)
    assert st[0] == 0 and c1[0].shape == (3, 2) and np.all(np.abs(c1[0] - exp) < 3e-7 * exp), c1[0]
    assert np.array_equal(np.sort(c0[0][:, 1]), np.sort(h0[0][:, 1])) or np.allclose(np.sort(c0[0][:-1, 1]), np.sort(h0[0][:-1, 1]), rtol=3e-7)


def test_scaling_by_a_power_of_two_scales_the_diagram_exactly(ctx):
    """Properties that need no oracle: doubling or halving every distance (and the threshold) is exact in float32 and
    float64, so every birth and death doubles or halves bit for bit and the row counts stay; a far-away extra point
    adds one H0 merge at its distance and changes nothing else.  710 EEG-like matrices, 200 clouds."""
    W = synth.eeg_windows(710, seed=77)
    dist = engine.corr_dist_batch(W, want_corr=False, ctx=ctx)
    h0, c0, h1, c1, st = engine.rips_dm_batch(dist, thresh=2.0, ctx=ctx, raw=True)
    for f in (2.0, 0.5, 1024.0):
        g0, d0, g1, d1, st2 = engine.rips_dm_batch(dist * f, thresh=2.0 * f, ctx=ctx, raw=True)
        assert np.array_equal(d0, c0) and np.array_equal(d1, c1) and not st2.any()
        for w in range(0, 710, 7):
            assert np.array_equal(g0[w, :c0[w]], h0[w, :c0[w]] * f) and np.array_equal(g1[w, :c1[w]], h1[w, :c1[w]] * f)
    rng = np.random.default_rng(5)
    P = rng.random((200, 60, 3))
    a0, a1, st = engine.cloud_rips_batch(P, normalise=False, thresh=100.0, ctx=ctx)
    b0, b1, st2 = engine.cloud_rips_batch(P * 4.0, normalise=False, thresh=400.0, ctx=ctx)
    assert not st.any() and not st2.any()
    far = np.concatenate([P, np.full((200, 1, 3), 50.0)], axis=1)
    f0, f1, st3 = engine.cloud_rips_batch(far, normalise=False, thresh=100.0, ctx=ctx)
    assert not st3.any()
    for w in range(200):
        assert np.array_equal(b0[w], a0[w] * 4.0) and np.array_equal(b1[w], a1[w] * 4.0)
        assert np.array_equal(f1[w], a1[w]) and len(f0[w]) == len(a0[w]) + 1
        assert np.array_equal(f0[w][:len(a0[w]) - 1], a0[w][:-1]) and f0[w][-2, 1] > 40.0 and np.isinf(f0[w][-1, 1])


def test_h1_of_lattices_cube_and_cross_polytope(ctx):
    """(k-1)^2 rows (1, sqrt 2) for the k x k unit lattice -- up to 100 classes alive at once, i.e. through the widening
    passes and, for the 11 x 11 lattice (121 points: the 128-class table does not fit LDS at that size; more than 64
    classes on a cloud), through the last rung with its class vectors in HBM --, 5 for the unit cube, none for
    cross-polytopes (tests/test_oracle_golden.py has the argument); both kernels, both first-pass class widths."""
    from test_oracle_golden import lattice
    r2 = np.float64(np.float32(np.sqrt(2.0)))

    def dm(P):
        return np.sqrt(((P[:, None] - P[None]) ** 2).sum(-1))
    for words in ((2, 1), (1, 1)):
        ctx.set_class_words(*words)
        try:
            for k in range(2, 12):
                P = lattice(k)
                h0, h1, st = engine.rips_dm_batch(dm(P)[None], thresh=100.0, ctx=ctx)
                assert st[0] == 0, (k, st)
                assert h1[0].shape == ((k - 1) ** 2, 2) and np.all(h1[0][:, 0] == 1.0) and np.all(h1[0][:, 1] == r2), k
                assert len(h0[0]) == k * k and np.all(h0[0][:-1, 1] == 1.0)
                c0, c1, st = engine.cloud_rips_batch(P[None], normalise=False, thresh=100.0, ctx=ctx)
                assert st[0] == 0, (k, st)
                assert c1[0].shape == ((k - 1) ** 2, 2) and np.all(c1[0][:, 0] == 1.0) and np.all(c1[0][:, 1] == r2), k
                assert len(c0[0]) == k * k and np.all(c0[0][:-1, 1] == 1.0)
            cube = lattice(2, 3)
            for h1 in (engine.rips_dm_batch(dm(cube)[None], thresh=100.0, ctx=ctx)[1][0],
                       engine.cloud_rips_batch(cube[None], normalise=False, thresh=100.0, ctx=ctx)[1][0]):
                assert h1.shape == (5, 2) and np.all(h1[:, 0] == 1.0) and np.all(h1[:, 1] == r2)
            for d in (3, 4):
                cross = np.concatenate([np.eye(d), -np.eye(d)])
                assert len(engine.rips_dm_batch(dm(cross)[None], thresh=100.0, ctx=ctx)[1][0]) == 0
                if d <= 4:
                    assert len(engine.cloud_rips_batch(cross[None], normalise=False, thresh=100.0, ctx=ctx)[1][0]) == 0
        finally:
            ctx.set_class_words(2, 1)


def test_rips_dm_known_answers(ctx):
    sq = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], float)
    hexa = np.array([[np.cos(k * np.pi / 3), np.sin(k * np.pi / 3)] for k in range(6)])
    tri = np.array([[0, 0], [1, 0], [0.3, 0.8]])

    def dm(p):
        return np.sqrt(((p[:, None] - p[None]) ** 2).sum(-1))

    h0, h1, st = engine.rips_dm_batch(np.stack([dm(sq)]), ctx=ctx)
    assert st[0] == 0
    assert np.array_equal(h0[0], [[0, 1], [0, 1], [0, 1], [0, np.inf]])
    assert np.array_equal(h1[0], np.array([[1.0, np.float32(np.sqrt(2.0))]], dtype=np.float64))
    h0, h1, st = engine.rips_dm_batch(np.stack([dm(hexa)]), ctx=ctx)
    assert len(h1[0]) == 1 and h1[0][0, 0] == np.float32(dm(hexa)[0, 1]) and h1[0][0, 1] == np.float32(np.sqrt(3.0))
    h0, h1, st = engine.rips_dm_batch(np.stack([dm(tri)]), ctx=ctx)
    assert len(h1[0]) == 0 and len(h0[0]) == 3
    # 4-cycle with thresh below the diagonal: one essential H1 class; two clusters: 2 essential H0
    h0, h1, st = engine.rips_dm_batch(np.stack([dm(sq)]), thresh=1.2, ctx=ctx)
    assert np.array_equal(h1[0], [[1.0, np.inf]])
    two = np.array([[0, 0], [0.1, 0], [5, 5], [5.1, 5]], float)
    h0, h1, st = engine.rips_dm_batch(np.stack([dm(two)]), thresh=1.0, ctx=ctx)
    assert np.isinf(h0[0][:, 1]).sum() == 2 and len(h0[0]) == 4
    # duplicated points: zero-length merges are dropped
    dup = np.array([[0, 0], [0, 0], [1, 0], [1, 0], [0, 1]], float)
    h0, h1, st = engine.rips_dm_batch(np.stack([dm(dup)]), ctx=ctx)
    assert len(h0[0]) == 5 - 1 - 2 + 1


def test_rips_dm_config2_710_windows_bit_exact(ctx):
    """BASELINE config 2: 710 EEG windows, 47x47, thresh 2.0."""
    W = synth.eeg_windows(710, seed=42)
    dist = engine.corr_dist_batch(W, want_corr=False, ctx=ctx)
    h0, h1, st = engine.rips_dm_batch(dist, ctx=ctx)
    assert not st.any()
    stc, oh0, ok0, oh1, ok1 = port.rips_dm_batch(dist)
    assert stc == 0
    for w in range(710):
        assert len(h0[w]) == ok0[w] and len(h1[w]) == ok1[w], w
        assert _same_multiset(h0[w], oh0[w, :ok0[w]]), w
        assert _same_multiset(h1[w], oh1[w, :ok1[w]]), w
        assert len(h0[w]) == 47 and np.isinf(h0[w][-1, 1]) and np.all(np.diff(h0[w][:-1, 1]) >= 0)
        assert np.all(np.diff(h1[w][:, 0]) <= 0)          # ripser order: descending birth


def test_rips_dm_white_noise_and_random_metrics(ctx):
    rng = np.random.default_rng(5)
    mats = []
    W = synth.eeg_windows(24, seed=9, kind="white")
    mats += list(engine.corr_dist_batch(W, want_corr=False, ctx=ctx))
    for _ in range(24):
        d = rng.random((47, 47)); d = (d + d.T) / 2; np.fill_diagonal(d, 0); mats.append(d)
    for _ in range(16):    # heavy ties
        d = np.round(rng.random((47, 47)) * 6) / 6; d = (d + d.T) / 2; np.fill_diagonal(d, 0); mats.append(d)
    mats = np.stack(mats)
    for thresh in (2.0, 0.9, 0.4):
        h0, h1, st = engine.rips_dm_batch(mats, thresh=thresh, h1_cap=512, ctx=ctx)
        assert not st.any()
        for w in range(len(mats)):
            o = port.rips_dm(mats[w], thresh=thresh)
            assert _same_multiset(h0[w], o[0]) and _same_multiset(h1[w], o[1]), (thresh, w)


def test_rips_dm_against_brute_force_small(ctx):
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 5, 13, 30, 64):
        mats = []
        for _ in range(6):
            d = rng.random((n, n)); d = (d + d.T) / 2; np.fill_diagonal(d, 0); mats.append(d)
        mats = np.stack(mats)
        for thresh in (2.0, 0.55):
            h0, h1, st = engine.rips_dm_batch(mats, thresh=thresh, h1_cap=1024, ctx=ctx)
            assert not st.any()
            for w in range(len(mats)):
                b0, b1 = brute.rips_brute(brute.eeg_prepare(mats[w]), thresh)
                assert _same_multiset(h0[w], b0) and _same_multiset(h1[w], b1), (n, thresh, w)


def test_rips_dm_larger_n_two_word_path(ctx):
    rng = np.random.default_rng(12)
    for n in (65, 100, 128):
        X = rng.standard_normal((n, 3))
        d = np.sqrt(((X[:, None] - X[None]) ** 2).sum(-1))
        h0, h1, st = engine.rips_dm_batch(d[None], thresh=10.0, h1_cap=1024, ctx=ctx)
        o = port.rips_dm(d, thresh=10.0)
        assert st[0] == 0 and _same_multiset(h0[0], o[0]) and _same_multiset(h1[0], o[1]), n


def test_rips_dm_symmetrise_semantics(ctx, golden):
    D = golden["ep_in"]                                   # asymmetric, non-zero diagonal, one negative
    keep = D.copy()
    h0, h1, st = engine.rips_dm_batch(D[None], ctx=ctx)
    assert np.array_equal(D, keep)                        # input untouched (utils.py:137)
    o = port.rips_f32(golden["ep_dm"].astype(np.float32)) # the matrix the reference hands to ripser
    assert _same_multiset(h0[0], o[0]) and _same_multiset(h1[0], o[1])


def test_rips_class_capacity_widens_automatically(ctx):
    """64 class bits are too few for white-noise windows (up to ~70 classes alive at once): the
    widening retry launches (128, 256, 512 bits) must repair exactly the flagged windows."""
    W = synth.eeg_windows(16, seed=3, kind="white")
    dist = engine.corr_dist_batch(W, want_corr=False, ctx=ctx)
    for words in (1, 2, 4):
        ctx.set_class_words(words, 1)
        try:
            h0, h1, st = engine.rips_dm_batch(dist, ctx=ctx)
            assert not st.any(), (words, st)
            for w in range(16):
                o = port.rips_dm(dist[w])
                assert _same_multiset(h0[w], o[0]) and _same_multiset(h1[w], o[1]), (words, w)
        finally:
            ctx.set_class_words(2, 1)


def test_rips_worst_case_class_count_n47(ctx):
    """Complete bipartite K(23,24) at distance 1, everything else at distance 2 - with thresh 1.5
    there are 23*24 - 46 = 506 essential H1 classes alive at once (the maximum for 47 points)."""
    n = 47
    side = np.arange(n) < 23
    d = np.where(side[:, None] != side[None, :], 1.0, 2.0)
    d = d + np.random.default_rng(0).random((n, n)) * 1e-3          # break ties
    d = (d + d.T) / 2
    np.fill_diagonal(d, 0)
    h0, h1, st = engine.rips_dm_batch(d[None], thresh=1.5, h1_cap=1024, ctx=ctx)
    assert st[0] == 0 and len(h1[0]) == 506 and np.isinf(h1[0][:, 1]).all()
    o = port.rips_dm(d, thresh=1.5)
    assert _same_multiset(h0[0], o[0]) and _same_multiset(h1[0], o[1])
    # and with the full threshold every one of them dies
    h0, h1, st = engine.rips_dm_batch(d[None], thresh=3.0, h1_cap=1024, ctx=ctx)
    o = port.rips_dm(d, thresh=3.0)
    assert st[0] == 0 and _same_multiset(h1[0], o[1])


def test_cloud_with_more_classes_than_lds_holds(ctx):
    """Point clouds above 112 points have one class word only in LDS; a cloud with more than 64 classes alive at once
    goes to the last rung of the ladder (class vectors in HBM) and comes back exact -- ripser never refuses a cloud."""
    k = 60
    ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
    a = np.stack([np.cos(ang), np.sin(ang), np.zeros(k)], 1)
    b = np.stack([np.cos(ang + np.pi / k), np.sin(ang + np.pi / k), np.full(k, 0.9)], 1)
    pc = np.concatenate([a, b])                      # two rings of 60: a "cylinder" graph, many squares
    h0, h1, st = engine.cloud_rips_batch(pc[None], normalise=False, thresh=0.95, h1_cap=1024, ctx=ctx)
    o = port.rips_f32(port.cloud_dm(pc).astype(np.float32), thresh=0.95)
    assert st[0] == 0
    assert _same_multiset(h0[0], o[0]) and _same_multiset(h1[0], o[1])


def test_last_rung_alone_on_ordinary_windows(ctx):
    """The pass with the class vectors in HBM is a complete sweep of its own (no chunks, no link argument, no class
    capacity): forced onto ordinary inputs (TDA_RETRY_LAST_RUNG on windows flagged by hand) it must give what the oracle
    gives -- audio windows of three bands, EEG-like and white-noise matrices, a tie-heavy metric."""
    import torch
    dev = torch.device("cuda", 0)
    ctx.set_retry_policy(ctx.RETRY_LAST_RUNG)
    try:
        for band in ("gamma", "alpha", "delta"):
            aw = synth.audio_windows(6, band, seed=31)
            tau = int(engine.tau_batch(aw[:1], 125, ctx=ctx)[0])
            out = engine.DeviceDiagrams(6, 128, 1024, dev)
            out.status.fill_(2)
            engine.takens_rips_dev(torch.from_numpy(aw).to(dev), torch.full((6,), tau, dtype=torch.int32, device=dev), out, ctx=ctx)
            torch.cuda.synchronize()
            a0, a1 = out.to_lists()
            assert int((out.status & ~4).max()) == 0
            for w in range(6):
                (o, _) = port.audio_persistence(aw[w], tau)
                assert _same_multiset(a0[w], o[0]) and _same_multiset(a1[w], o[1]), (band, w)
        rng = np.random.default_rng(5)
        mats = [engine.corr_dist_batch(synth.eeg_windows(3, seed=9, kind=k), want_corr=False, ctx=ctx) for k in ("latent", "white")]
        q = np.round(rng.random((2, 40, 40)) * 6) / 6                      # heavy ties
        q = (q + q.transpose(0, 2, 1)) / 2
        for i in range(2):
            np.fill_diagonal(q[i], 0.0)
        for d in mats + [q]:
            n_win, n = d.shape[0], d.shape[1]
            out = engine.DeviceDiagrams(n_win, n, 1100, dev)
            out.status.fill_(2)
            engine.rips_dm_dev(torch.from_numpy(np.ascontiguousarray(d)).to(dev), out, h1_cap=1100, ctx=ctx)
            torch.cuda.synchronize()
            a0, a1 = out.to_lists()
            assert int(out.status.max()) == 0
            for w in range(n_win):
                o = port.rips_dm(d[w])
                assert _same_multiset(a0[w], o[0]) and _same_multiset(a1[w], o[1]), w
    finally:
        ctx.set_retry_policy(ctx.RETRY_AUTO)


def test_widening_passes_redo_exactly_the_flagged_windows(ctx):
    """The widening passes work off a list of the flagged windows (rips.hip: retry_collect_kernel).  Hand-made flags on a
    batch larger than any retry grid -- every window, every third one, none -- redone by the widening kernels alone
    (TDA_RETRY_ONLY): the flagged windows get the diagrams of an ordinary run bit for bit, the others are not touched
    (their rows stay the poison written beforehand), and a second pass over the emptied list changes nothing."""
    import torch
    dev = torch.device("cuda", 0)
    n = 1500
    wins, _ = synth.audio_windows_all_bands(n // 5, seed=11)
    wins = wins[:n]
    wt = torch.from_numpy(np.ascontiguousarray(wins)).to(dev)
    tau = engine.tau_dev(wt, max_lag=125, ctx=ctx)
    ref = engine.takens_rips_dev(wt, tau, ctx=ctx)
    torch.cuda.synchronize()
    assert not ref.status.cpu().numpy().any()
    r0, r1 = ref.to_lists()
    for every in (1, 3, 0):
        out = engine.DeviceDiagrams(n, 128, engine.DEFAULT_H1_CAP, dev)
        out.status.zero_(); out.c0.fill_(-5); out.c1.fill_(-5)
        flagged = np.zeros(n, bool)
        if every:
            flagged[::every] = True
            out.status[torch.from_numpy(np.nonzero(flagged)[0]).to(dev)] = 2
        ctx.set_retry_policy(ctx.RETRY_ONLY)
        try:
            for _ in range(2):
                engine.takens_rips_dev(wt, tau, out, ctx=ctx)
        finally:
            ctx.set_retry_policy(ctx.RETRY_AUTO)
        torch.cuda.synchronize()
        c0, c1, st = out.c0.cpu().numpy(), out.c1.cpu().numpy(), out.status.cpu().numpy()
        assert not st.any()
        assert (c0[~flagged] == -5).all() and (c1[~flagged] == -5).all()
        h0 = out.h0.cpu().numpy(); h1 = out.h1.cpu().numpy()
        for w in np.nonzero(flagged)[0][:: max(1, int(flagged.sum()) // 200)]:
            assert _same_multiset(h0[w, :c0[w]], r0[w]) and _same_multiset(h1[w, :c1[w]], r1[w]), (every, int(w))


def test_rips_h1_truncation_flag(ctx):
    W = synth.eeg_windows(4, seed=3, kind="white")
    dist = engine.corr_dist_batch(W, want_corr=False, ctx=ctx)
    h0, c0, h1, c1, st = engine.rips_dm_batch(dist, h1_cap=8, ctx=ctx, raw=True)
    assert np.all(st & 1) and np.all(c1 > 8)
    for w in range(4):
        assert c1[w] == len(port.rips_dm(dist[w])[1])


# ------------------------------------------------------------------ tau, Takens + Rips (audio)
def test_tau_bit_exact(ctx, golden):
    wins, _ = synth.audio_windows_all_bands(40, seed=5)
    tau = engine.tau_batch(wins, max_lag=125, ctx=ctx)
    ref = np.array([port.compute_tau(w, 125) for w in wins])
    assert np.array_equal(tau, ref)
    assert np.array_equal(engine.tau_batch(golden["tau_signals"], max_lag=125, ctx=ctx), golden["tau_values"])
    assert engine.tau_batch(np.full((1, 250), 1.5), max_lag=125, ctx=ctx)[0] == golden["tau_const"][0]
    ramp = np.arange(250.0)[None]
    assert engine.tau_batch(ramp, max_lag=125, ctx=ctx)[0] == golden["tau_ramp"][0]
    assert engine.tau_batch(ramp, max_lag=None, ctx=ctx)[0] == golden["tau_ramp"][1]


def test_takens_rips_config4_bit_exact(ctx):
    """BASELINE config 4 shape: band-limited windows -> tau -> Takens(3, tau, 2) -> Rips."""
    wins, band = synth.audio_windows_all_bands(30, seed=42)
    tau = engine.tau_batch(wins, max_lag=125, ctx=ctx)
    h0, h1, npts, st = engine.takens_rips_batch(wins, tau, ctx=ctx)
    assert not st.any()
    for w in range(len(wins)):
        (o0, o1), P = port.audio_persistence(wins[w], int(tau[w]))
        assert npts[w] == P
        assert _same_multiset(h0[w], o0) and _same_multiset(h1[w], o1), (w, int(tau[w]))
        assert len(h0[w]) <= P


def test_takens_rips_edge_cases(ctx):
    wins = synth.audio_windows(6, "delta", seed=1)
    tau = np.array([124, 125, 200, 1, 100, 62], np.int32)     # P = 1, 0, 0, 124, 25, 63
    h0, h1, npts, st = engine.takens_rips_batch(wins, tau, ctx=ctx)
    assert list(npts) == [1, 0, 0, 124, 25, 63]
    for w in (0, 1, 2):                                        # utils.py:125-126
        assert st[w] == 4 and np.array_equal(h0[w], [[0, 0]]) and np.array_equal(h1[w], [[0, 0]])
    for w in (3, 4, 5):
        (o0, o1), P = port.audio_persistence(wins[w], int(tau[w]))
        assert st[w] == 0 and _same_multiset(h0[w], o0) and _same_multiset(h1[w], o1)
    # constant window: zero range -> 1 (utils.py:129); all points coincide
    const = np.full((1, 250), 0.75)
    h0, h1, npts, st = engine.takens_rips_batch(const, np.array([2], np.int32), ctx=ctx)
    assert np.array_equal(h0[0], [[0, np.inf]]) and len(h1[0]) == 0


def test_cloud_rips_matches_reference_preprocessing(ctx, golden):
    """compute_audio_persistence(point_cloud): normalisation pinned by the recording stub."""
    for name in ("delta", "theta", "alpha", "beta", "gamma"):
        pc = golden["tk_pc_" + name]
        h0, h1, st = engine.cloud_rips_batch(pc[None], ctx=ctx)
        # oracle fed with the reference's own pc_norm and sklearn's own distance matrix
        o = port.rips_f32(golden["pd_" + name].astype(np.float32))
        assert st[0] == 0 and _same_multiset(h0[0], o[0]) and _same_multiset(h1[0], o[1]), name


def test_small_clouds_short_last_chunk(ctx):
    """Clouds whose edge list ends inside a sweep chunk (fewer edges than one chunk, or a short last chunk),
    every class-vector width.  tests/golden/regress_cloud_p31.npy is a 31-point cloud on which the table
    rewrite once walked past the end of the edge list (found by tools/stress_parity.py)."""
    import os
    pcs = [np.load(os.path.join(os.path.dirname(__file__), "golden", "regress_cloud_p31.npy"))]
    rng = np.random.default_rng(77)
    for P in (3, 4, 7, 12, 20, 31, 32, 33, 40, 46):
        for rep in range(4):
            pc = rng.random((P, 3))
            if rep == 3:
                pc[P // 2:] = pc[:P - P // 2]                  # duplicate points: zero-length edges
            pcs.append(pc)
    try:
        for words in ((2, 1), (1, 1), (2, 2)):
            ctx.set_class_words(*words)
            for i, pc in enumerate(pcs):
                h0, h1, st = engine.cloud_rips_batch(pc[None], normalise=True, thresh=2.0, h1_cap=4096, ctx=ctx)
                dm = port.cloud_dm(port.minmax_normalise(pc)).astype(np.float32)
                o = port.rips_f32(dm, thresh=2.0)
                assert (st[0] & ~4) == 0
                assert _same_multiset(h0[0], o[0]) and _same_multiset(h1[0], o[1]), (words, i, len(pc))
                if len(pc) <= 12:
                    b = brute.rips_brute(dm.astype(np.float64), 2.0)
                    assert _same_multiset(h1[0], b[1]), (words, i)
    finally:
        ctx.set_class_words(2, 1)


# ------------------------------------------------------------------ features
def test_features_vs_reference_golden_and_oracle(ctx, golden):
    names = ["mixed", "single", "empty_finite", "zero_pers", "f32vals", "big"]
    rows, cnt = engine.pack_diagrams([golden["ef_in_" + k] for k in names])
    feat = engine.features_batch(rows, cnt, ctx=ctx)
    for i, k in enumerate(names):
        ref = golden["ef_out_" + k]
        assert np.allclose(feat[i], ref, rtol=1e-12, atol=1e-15), k
        assert np.array_equal(feat[i, :10], ref[:10]), k          # everything but the entropy (log) bit-exact
    rows, cnt = engine.pack_diagrams([np.zeros((0, 2))])
    assert np.array_equal(engine.features_batch(rows, cnt, ctx=ctx)[0], np.zeros(11))


def test_features_and_aggregate_on_real_diagrams(ctx):
    W = synth.eeg_windows(78, seed=21)
    dist = engine.corr_dist_batch(W, want_corr=False, ctx=ctx)
    h0, c0, h1, c1, st = engine.rips_dm_batch(dist, ctx=ctx, raw=True)
    f0 = engine.features_batch(h0, c0, ctx=ctx)
    f1 = engine.features_batch(h1, c1, ctx=ctx)
    for w in range(78):
        assert np.allclose(f0[w], port.features(h0[w, :c0[w]]), rtol=1e-12, atol=0)
        assert np.allclose(f1[w], port.features(h1[w, :c1[w]]), rtol=1e-12, atol=0)
    seg = np.array([0, 39, 78], np.int32)
    agg = engine.aggregate_batch(f0, f1, seg, ctx=ctx)
    for s in range(2):
        a, b = seg[s], seg[s + 1]
        for f in range(11):
            exp = [np.mean(f0[a:b, f]), np.std(f0[a:b, f]), np.mean(f1[a:b, f]), np.std(f1[a:b, f])]
            assert np.array_equal(agg[s, 4 * f:4 * f + 4], exp), (s, f)


def test_pairwise_sums_beyond_128_terms(ctx):
    """numpy sums more than 128 terms by recursive halving (left half a multiple of 8); the kernels walk the same tree
    with a stack kept across the lanes of the wave: diagrams of 129..2000 rows and groups of 129..1111 windows, the
    sums bit-equal to np.mean / np.std / np.sum."""
    rng = np.random.default_rng(12)
    dg = []
    for k in (129, 130, 136, 255, 256, 257, 700, 1025, 2000):
        b = rng.random(k)
        d = np.stack([b, b + rng.random(k)], axis=1).astype(np.float32).astype(np.float64)
        d[rng.integers(0, k, 3), 1] = np.inf
        dg.append(d)
    rows, cnt = engine.pack_diagrams(dg)
    feat = engine.features_batch(rows, cnt, ctx=ctx)
    for i, d in enumerate(dg):
        ref = port.features(d)
        assert np.array_equal(feat[i, :10], ref[:10]), len(d)
        assert np.allclose(feat[i, 10], ref[10], rtol=1e-12, atol=0)
    f0 = rng.standard_normal((129 + 300 + 1111 + 5, 11)); f1 = rng.standard_normal(f0.shape)
    seg = np.array([0, 129, 429, 1540, 1545], np.int32)
    agg = engine.aggregate_batch(f0, f1, seg, ctx=ctx)
    for s_ in range(4):
        a, b = seg[s_], seg[s_ + 1]
        for f in range(11):
            exp = [np.mean(f0[a:b, f]), np.std(f0[a:b, f]), np.mean(f1[a:b, f]), np.std(f1[a:b, f])]
            assert np.array_equal(agg[s_, 4 * f:4 * f + 4], exp), (s_, f)


# ------------------------------------------------------------------ Wasserstein
def test_wasserstein_known_answers(ctx):
    A = np.array([[0.0, 1.0]]); B = np.array([[0.0, 2.0]])
    E = np.zeros((0, 2))
    pts = np.array([[0.1, 0.9], [0.2, 0.5], [0.0, 0.3]])
    ra, ca = engine.pack_diagrams([A, A, pts, E, pts])
    rb, cb = engine.pack_diagrams([B, A, E, E, pts[::-1]])
    out, st = engine.wasserstein_batch(ra, ca, rb, cb, ctx=ctx, want_status=True)
    assert not st.any()
    assert abs(out[0] - 1.0) < 1e-12
    assert out[1] == 0.0
    assert abs(out[2] - ((pts[:, 1] - pts[:, 0]) / np.sqrt(2)).sum()) < 1e-12
    assert out[3] == 0.0
    assert abs(out[4]) < 1e-12


def test_wasserstein_vs_persim_restatement_and_bruteforce(ctx):
    rng = np.random.default_rng(8)
    As, Bs = [], []
    for _ in range(200):
        As.append(np.sort(rng.random((int(rng.integers(0, 60)), 2)), axis=1))
        Bs.append(np.sort(rng.random((int(rng.integers(0, 130)), 2)), axis=1))
    ra, ca = engine.pack_diagrams(As, cap=64); rb, cb = engine.pack_diagrams(Bs, cap=130)
    out, st = engine.wasserstein_batch(ra, ca, rb, cb, ctx=ctx, want_status=True)
    assert not st.any()
    ref = np.array([brute.safe_wasserstein_oracle(a, b) for a, b in zip(As, Bs)])
    assert np.abs(out - ref).max() < 1e-6            # north_star tolerance
    assert np.abs(out - ref).max() < 1e-10           # what is actually achieved
    # symmetry
    out2 = engine.wasserstein_batch(rb, cb, ra, ca, ctx=ctx)
    assert np.abs(out - out2).max() < 1e-10
    # buffers of 256 rows (the H1 capacity of the pipeline): the small first launch takes the pairs of up to 64 x 64
    # points, the launch sized by the capacities the rest -- sizes on both sides of the limit, in one batch
    sizes = [(0, 0), (1, 64), (64, 64), (65, 3), (64, 65), (100, 100), (3, 250), (200, 130), (40, 41), (63, 64)]
    As = [np.sort(rng.random((m, 2)), axis=1) for m, _ in sizes]
    Bs = [np.sort(rng.random((n, 2)), axis=1) for _, n in sizes]
    ra, ca = engine.pack_diagrams(As, cap=256); rb, cb = engine.pack_diagrams(Bs, cap=256)
    out, st = engine.wasserstein_batch(ra, ca, rb, cb, ctx=ctx, want_status=True)
    assert not st.any()
    ref = np.array([brute.safe_wasserstein_oracle(a, b) for a, b in zip(As, Bs)])
    assert np.abs(out - ref).max() < 1e-10, np.abs(out - ref)
    # exhaustive optimum for tiny diagrams
    As = [np.sort(rng.random((int(rng.integers(1, 5)), 2)), axis=1) for _ in range(20)]
    Bs = [np.sort(rng.random((int(rng.integers(1, 5)), 2)), axis=1) for _ in range(20)]
    ra, ca = engine.pack_diagrams(As); rb, cb = engine.pack_diagrams(Bs)
    out = engine.wasserstein_batch(ra, ca, rb, cb, ctx=ctx)
    for i in range(20):
        assert abs(out[i] - brute.wasserstein_bruteforce(As[i], Bs[i])) < 1e-9


def test_wasserstein_equal_birth_fast_path(ctx):
    """Two diagrams whose points all share one birth (H0 vs H0) take the 1-D wavefront path: compare
    with the full assignment (persim restatement on scipy's solver), incl. coincident and duplicated
    deaths, and check that unsorted input (general solver) gives the same value."""
    rng = np.random.default_rng(21)
    As, Bs = [], []
    for k in range(120):
        m, n = int(rng.integers(1, 60)), int(rng.integers(1, 125))
        a = np.sort(rng.random(m).astype(np.float32).astype(np.float64)) * (1.5 if k % 3 else 0.3)
        b = np.sort(rng.random(n).astype(np.float32).astype(np.float64)) * (0.4 if k % 2 else 1.5)
        if k % 5 == 0:
            b[: min(m, n) // 2] = a[: min(m, n) // 2][: len(b[: min(m, n) // 2])]     # exact coincidences
            b = np.sort(b)
        if k % 7 == 0:
            a[1:] = np.where(rng.random(m - 1) < 0.3, a[:-1], a[1:]); a = np.sort(a)  # duplicates
        birth = 0.0 if k % 4 else 0.125
        As.append(np.stack([np.full(m, birth), birth + a], 1)); Bs.append(np.stack([np.full(n, birth), birth + b], 1))
    ra, ca = engine.pack_diagrams(As, cap=64); rb, cb = engine.pack_diagrams(Bs, cap=128)
    out, st = engine.wasserstein_batch(ra, ca, rb, cb, ctx=ctx, want_status=True)
    ref = np.array([brute.wasserstein_persim(a, b) for a, b in zip(As, Bs)])
    assert not st.any() and np.abs(out - ref).max() < 1e-9, np.abs(out - ref).max()
    # shuffled rows -> not sorted -> general solver; same optimum
    As2 = [a[rng.permutation(len(a))] for a in As]
    Bs2 = [b[rng.permutation(len(b))] for b in Bs]
    ra2, ca2 = engine.pack_diagrams(As2, cap=64); rb2, cb2 = engine.pack_diagrams(Bs2, cap=128)
    out2 = engine.wasserstein_batch(ra2, ca2, rb2, cb2, ctx=ctx)
    assert np.abs(out2 - ref).max() < 1e-9 and np.abs(out2 - out).max() < 1e-9


def test_wasserstein_infinite_rows_ignored_and_index_pairs(ctx):
    A = np.array([[0.0, 0.5], [0.0, np.inf], [0.1, 0.7]])
    B = np.array([[0.0, np.inf]])
    C_ = np.array([[0.2, 0.9], [0.3, 0.35]])
    ra, ca = engine.pack_diagrams([A, B, C_])
    idx_a = np.array([0, 0, 1, 2], np.int32); idx_b = np.array([1, 2, 2, 0], np.int32)
    out = engine.wasserstein_batch(ra, ca, ra, ca, idx_a, idx_b, ctx=ctx)
    dg = [A, B, C_]
    for k in range(4):
        assert abs(out[k] - brute.safe_wasserstein_oracle(dg[idx_a[k]], dg[idx_b[k]])) < 1e-10


def test_end_to_end_h0_h1_wasserstein_on_pipeline_diagrams(ctx):
    """cmp:88-96 on synthetic data: EEG diagrams vs audio diagrams, H0 and H1."""
    n = 60
    W = synth.eeg_windows(n, seed=77)
    aw = synth.audio_windows(n, "beta", seed=78)
    dist = engine.corr_dist_batch(W, want_corr=False, ctx=ctx)
    e0, ec0, e1, ec1, est = engine.rips_dm_batch(dist, ctx=ctx, raw=True)
    tau = engine.tau_batch(aw[:1], max_lag=125, ctx=ctx)[0]
    a0, ac0, a1, ac1, npts, ast = engine.takens_rips_batch(aw, tau, ctx=ctx, raw=True)
    w0 = engine.wasserstein_batch(e0, ec0, a0, ac0, ctx=ctx)
    w1 = engine.wasserstein_batch(e1, ec1, a1, ac1, ctx=ctx)
    for w in range(n):
        r0 = brute.safe_wasserstein_oracle(e0[w, :ec0[w]], a0[w, :ac0[w]])
        r1 = brute.safe_wasserstein_oracle(e1[w, :ec1[w]], a1[w, :ac1[w]])
        assert abs(w0[w] - r0) < 1e-6 and abs(w1[w] - r1) < 1e-6, (w, w0[w], r0, w1[w], r1)


# ------------------------------------------------------------------ whole step, batches in flight
def test_pipeline_step_vs_oracle_and_lanes(ctx):
    """pipeline.run_step (cmp:77-122 batched) against the CPU restatement of the same unit, and
    pipeline.Lanes: five different batches through three lanes give bit-identical rows to the same
    batches run one after the other (no buffer of a lane is shared or reused too early)."""
    import torch
    from oracle import pipeline_ref
    from tda_eeg_audio_amd import pipeline
    dev = torch.device("cuda", 0)
    n_win, wpr = 45, 15
    seg_off = np.array([0, 15, 30, 45], np.int32)
    batches = []
    for k in range(5):
        eeg = synth.eeg_windows(n_win, seed=500 + k, windows_per_recording=wpr)
        aud = synth.audio_windows(n_win, ["beta", "alpha", "delta", "theta", "gamma"][k], seed=600 + k)
        batches.append((eeg, aud, torch.from_numpy(eeg).to(dev), torch.from_numpy(aud).to(dev)))
    ws = pipeline.Workspace(n_win, seg_off, dev)
    serial = []
    for eeg, aud, eeg_t, aud_t in batches:
        res = pipeline.run_step(eeg_t, aud_t, ws, ctx=ctx)
        torch.cuda.synchronize()
        assert int(ws.eeg.status.max()) == 0 and int((ws.aud.status & ~4).max()) == 0
        serial.append(res.cpu().numpy().copy())
    ref = pipeline_ref.reference_step_cpu(batches[0][0], batches[0][1], seg_off)
    assert np.abs(serial[0][:, :2] - ref[:, :2]).max() < 1e-6          # Wasserstein means (north_star bar)
    assert np.array_equal(serial[0][:, 2:4], ref[:, 2:4])              # tau, window counts
    assert np.allclose(serial[0][:, 4:], ref[:, 4:], rtol=1e-9, atol=1e-12)
    lanes = pipeline.Lanes(3, n_win, seg_off, dev)
    got = [lanes.submit(eeg_t, aud_t, ctx=ctx, post=lambda r: r.clone()) for _, _, eeg_t, aud_t in batches]
    lanes.drain()
    torch.cuda.synchronize()
    for k in range(5):
        assert np.array_equal(got[k].result().cpu().numpy(), serial[k], equal_nan=True), k
    # (a batch whose first pass leaves a flag -- a window whose classes outlive the narrow layout of the audio kernel, or
    # more classes alive than bits -- is re-run with the full ladder before it is published: equal to the serial result above)
    assert lanes.repairs <= len(batches)
    # the same through captured HIP graphs (one per lane and input buffer pair), replayed twice
    glanes = pipeline.Lanes(2, n_win, seg_off, dev, graph=True)
    for rnd in range(2):
        got = [glanes.submit(eeg_t, aud_t, ctx=ctx, post=lambda r: r.clone()) for _, _, eeg_t, aud_t in batches[:4]]
        glanes.drain()
        torch.cuda.synchronize()
        for k in range(4):
            assert np.array_equal(got[k].result().cpu().numpy(), serial[k], equal_nan=True), (rnd, k)
    assert len(glanes.graphs) == 4


def test_lanes_verify_then_publish_repairs_overflowing_batches(ctx):
    """Deferred retries: with 64 class bits white-noise EEG windows overflow the first pass; the lanes must notice
    (flags in pinned memory), re-run those batches with the widening passes and publish the exact rows."""
    import torch
    from tda_eeg_audio_amd import pipeline
    dev = torch.device("cuda", 0)
    n_win, wpr = 30, 15
    seg_off = np.array([0, 15, 30], np.int32)
    batches = []
    for k in range(4):
        kind = "white" if k % 2 == 0 else "latent"
        eeg = synth.eeg_windows(n_win, seed=900 + k, windows_per_recording=wpr, kind=kind)
        aud = synth.audio_windows(n_win, "beta", seed=950 + k)
        batches.append((torch.from_numpy(eeg).to(dev), torch.from_numpy(aud).to(dev)))
    ctx.set_class_words(1, 1)
    try:
        ws = pipeline.Workspace(n_win, seg_off, dev)
        serial = []
        for eeg_t, aud_t in batches:
            serial.append(pipeline.run_step(eeg_t, aud_t, ws, ctx=ctx, retry="auto").cpu().numpy().copy())
            assert int(ws.eeg.status.max()) == 0
        # the first pass alone does flag windows of the white-noise batches
        pipeline.run_step(batches[0][0], batches[0][1], ws, ctx=ctx, retry="first")
        torch.cuda.synchronize()
        assert int((ws.eeg.status & 2).max()) == 2 and bool(ws.flags_host.any())
        for graph in (False, True):
            lanes = pipeline.Lanes(2, n_win, seg_off, dev, graph=graph, defer_retries=True)
            got = [lanes.submit(e, a, ctx=ctx, post=lambda r: r.clone()) for e, a in batches]
            lanes.drain()
            torch.cuda.synchronize()
            assert lanes.repairs >= 2, lanes.repairs
            for k in range(4):
                assert got[k].repaired == (k % 2 == 0) or got[k].repaired, k
                assert np.array_equal(got[k].result().cpu().numpy(), serial[k], equal_nan=True), (graph, k)
    finally:
        ctx.set_class_words(2, 1)


def test_fused_row_kernels_equal_their_parts(ctx):
    """tda_tau_segments_dev == tda_tau_batch_dev on the first window of each group (broadcast per window);
    tda_recording_rows_dev == tda_segment_nanmean x 2 + tda_aggregate_batch + tau / count columns, bit for bit."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    seg = np.array([0, 15, 16, 40, 40, 53], np.int32)              # includes a single-window and an empty group
    n = int(seg[-1])
    aud = torch.from_numpy(synth.audio_windows(n, "theta", seed=9)).to(dev)
    seg_t = torch.from_numpy(seg).to(dev)
    tau_w = torch.full((n,), -7, dtype=torch.int32, device=dev)
    tau_s = engine.tau_segments_dev(aud, seg_t, 125, None, tau_w, ctx=ctx).cpu().numpy()
    first = [int(s) for s, e in zip(seg[:-1], seg[1:]) if e > s]
    ref = engine.tau_dev(aud[first].contiguous(), 125, ctx=ctx).cpu().numpy()
    nonempty = np.diff(seg) > 0
    assert np.array_equal(tau_s[nonempty], ref) and np.all(tau_s[~nonempty] == 0)
    assert np.array_equal(tau_w.cpu().numpy(), np.repeat(tau_s, np.diff(seg)))
    w0 = rng.random(n); w1 = rng.random(n); w1[3] = np.nan; w1[15] = np.nan        # group 1 is all-NaN in w1
    f0 = rng.random((n, 11)); f1 = rng.random((n, 11))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rows = engine.recording_rows_dev(t(w0), t(w1), torch.from_numpy(tau_s).to(dev), t(f0), t(f1), seg_t, ctx=ctx).cpu().numpy()
    assert np.array_equal(rows[:, 0], engine.segment_nanmean_dev(t(w0), seg_t, ctx=ctx).cpu().numpy(), equal_nan=True)
    assert np.array_equal(rows[:, 1], engine.segment_nanmean_dev(t(w1), seg_t, ctx=ctx).cpu().numpy(), equal_nan=True)
    assert np.isnan(rows[1, 1]) and np.isnan(rows[3, 0])
    assert np.array_equal(rows[:, 2], tau_s.astype(np.float64)) and np.array_equal(rows[:, 3], np.diff(seg).astype(np.float64))
    assert np.array_equal(rows[:, 4:], engine.aggregate_dev(t(f0), t(f1), seg_t, ctx=ctx).cpu().numpy())


# ------------------------------------------------------------------ fused EEG window kernel
def test_fused_eeg_window_equals_two_kernels(ctx):
    """tda_eeg_window_batch_dev (samples -> corr -> dist -> Rips in one launch, nb2:198-207 + utils.py:135-141) against
    tda_corr_dist_batch_dev + tda_rips_dm_batch_dev on BASELINE.json configs[1] (710 windows): rows, counts and
    status words bit for bit, and the optional matrices equal to the corr_dist kernel's."""
    import torch
    dev = torch.device("cuda", ctx.device)
    W = synth.eeg_windows(710, seed=11, windows_per_recording=15)
    W[5, 7] = 0.5                        # zero-variance channel
    W[6, 12] = W[6, 3]                   # duplicated channel: zero-length edge
    wt = torch.from_numpy(W).to(dev)
    dist = torch.empty((710, 47, 47), dtype=torch.float64, device=dev)
    corr = torch.empty_like(dist)
    engine.corr_dist_dev(wt, dist, corr, ctx=ctx)
    two = engine.rips_dm_dev(dist, ctx=ctx)
    for words in (0, 1, 2):              # 32 (fused kernel only), 64, 128 class bits in the first pass
        ctx.set_class_words(words, 1)
        try:
            d2 = torch.full_like(dist, -1.0)
            c2 = torch.full_like(dist, -1.0)
            one = engine.eeg_window_dev(wt, dist_t=d2, corr_t=c2, ctx=ctx)
            bare = engine.eeg_window_dev(wt, ctx=ctx)               # without the matrices
        finally:
            ctx.set_class_words(2, 1)
        torch.cuda.synchronize()
        assert torch.equal(d2, dist) and torch.equal(c2, corr)
        for got in (one, bare):
            assert torch.equal(got.c0, two.c0) and torch.equal(got.c1, two.c1) and torch.equal(got.status, two.status)
            assert int(got.status.max()) == 0
            k0 = int(two.c0.max()); k1 = int(two.c1.max())
            m0 = torch.arange(47, device=dev)[None, :] < two.c0[:, None]
            m1 = torch.arange(got.h1.shape[1], device=dev)[None, :] < two.c1[:, None]
            assert torch.equal(got.h0[m0], two.h0[m0]) and torch.equal(got.h1[m1], two.h1[m1]) and k0 <= 47 and k1 > 0
    # and against the oracle on a sample
    e0, e1 = one.to_lists()
    for w in (0, 5, 6, 349, 709):
        _, od = port.corr_dist(W[w])
        o = port.rips_dm(od)
        assert _same_multiset(e0[w], o[0]) and _same_multiset(e1[w], o[1])


def test_fused_eeg_window_widening_passes(ctx):
    """White-noise windows need more than 64 classes at once: the first pass flags them and the widening passes of
    the fused kernel (which recompute the matrix from the samples) deliver the same diagrams as the two-kernel path."""
    import torch
    dev = torch.device("cuda", ctx.device)
    W = synth.eeg_windows(96, seed=5, kind="white")
    wt = torch.from_numpy(W).to(dev)
    dist = engine.corr_dist_dev(wt, ctx=ctx)
    two = engine.rips_dm_dev(dist, ctx=ctx)
    b0, b1 = two.to_lists()
    for words in (1, 0):                 # ladders 64 -> 128 -> 512 and 32 -> 64 -> 128 -> 512
        ctx.set_class_words(words, 1)
        try:
            ctx.set_retry_policy(ctx.RETRY_FIRST_PASS)
            first = engine.eeg_window_dev(wt, ctx=ctx)
            torch.cuda.synchronize()
            flagged = int((first.status & 2).ne(0).sum())
            ctx.set_retry_policy(ctx.RETRY_AUTO)
            one = engine.eeg_window_dev(wt, ctx=ctx)
        finally:
            ctx.set_retry_policy(ctx.RETRY_AUTO)
            ctx.set_class_words(2, 1)
        torch.cuda.synchronize()
        assert flagged > 0, "the test needs windows that overflow the first pass"
        assert int(one.status.max()) == 0 and torch.equal(one.c0, two.c0) and torch.equal(one.c1, two.c1)
        a0, a1 = one.to_lists()
        for w in range(96):
            assert np.array_equal(a0[w], b0[w]) and np.array_equal(a1[w], b1[w])


def _bipartite_windows(n_rep, seed):
    """47-channel windows whose correlation graph is K(23,24): every cross-group correlation 0.23, every
    within-group one 0.20 (a positive-definite Gram matrix), so the 552 cross edges (d = 1.241) enter before any
    within-group edge (d = 1.265) and (23-1)(24-1) = 506 H1 classes are alive at once -- the theoretical maximum
    for 47 points, which only the 512-bit widening pass can hold."""
    rng = np.random.default_rng(seed)
    m, n, c, w = 23, 24, 0.23, 0.20
    G = np.full((47, 47), w)
    G[:m, m:] = c; G[m:, :m] = c
    np.fill_diagonal(G, 1.0)
    ev, U = np.linalg.eigh(G)
    assert ev.min() > 0
    L = (U * np.sqrt(ev)) @ U.T
    out = np.empty((n_rep, 47, 250))
    for k in range(n_rep):
        Z = rng.standard_normal((250, 47))
        Z -= Z.mean(axis=0)
        Q, _ = np.linalg.qr(Z)
        out[k] = (L @ Q.T) * rng.uniform(0.5, 2.0, size=(47, 1))      # per-channel scale: correlation unchanged
    return out


def test_fused_eeg_window_512_classes(ctx):
    """The widest widening pass of the fused kernel (512 class bits, 256 VGPRs) on windows that need 506 classes
    alive at once, against the two-kernel path and the oracle."""
    import torch
    dev = torch.device("cuda", ctx.device)
    W = _bipartite_windows(5, seed=3)
    W = np.concatenate([W, synth.eeg_windows(7, seed=2)])             # ordinary windows in the same batch
    wt = torch.from_numpy(W).to(dev)
    dist = engine.corr_dist_dev(wt, ctx=ctx)
    two = engine.rips_dm_dev(dist, h1_cap=1024, ctx=ctx)
    ctx.set_retry_policy(ctx.RETRY_FIRST_PASS)
    try:
        first = engine.eeg_window_dev(wt, h1_cap=1024, ctx=ctx)
        torch.cuda.synchronize()
        assert (first.status[:5] & 2).ne(0).all() and (first.status[5:] == 0).all()
    finally:
        ctx.set_retry_policy(ctx.RETRY_AUTO)
    one = engine.eeg_window_dev(wt, h1_cap=1024, ctx=ctx)
    torch.cuda.synchronize()
    assert int(one.status.max()) == 0 and int(two.status.max()) == 0
    assert torch.equal(one.c0, two.c0) and torch.equal(one.c1, two.c1)
    a0, a1 = one.to_lists(); b0, b1 = two.to_lists()
    d = dist.cpu().numpy()
    for w in range(len(W)):
        assert np.array_equal(a0[w], b0[w]) and np.array_equal(a1[w], b1[w])
        o = port.rips_dm(d[w])
        assert _same_multiset(a0[w], o[0]) and _same_multiset(a1[w], o[1])
    assert min(len(a1[w]) for w in range(5)) >= 400              # hundreds of classes born before the first death


def test_recording_rows_skip_degenerate_audio_windows(ctx):
    """cmp:90-91: windows whose Takens cloud has fewer than 3 points (status bit 4) or none (16) take no part in the
    distances; np.nanmean runs over the survivors in their order (cmp:117-118), NaN when none is left (the reference
    drops the band, cmp:101-102).  recording_rows_kernel against numpy on crafted status words."""
    import torch
    dev = torch.device("cuda", ctx.device)
    rng = np.random.default_rng(3)
    seg = np.array([0, 15, 30, 45, 52], np.int32)
    n = 52
    w0 = rng.random(n) * 20; w1 = rng.random(n)
    w1[3] = np.nan; w0[20] = np.nan                      # failed solves stay NaN and are ignored by nanmean
    stb = np.zeros(n, np.int32)
    stb[[1, 4, 16, 17, 18]] = 4                           # a few short clouds
    stb[30:45] = 4                                        # a whole group: tau = 124
    stb[47] = 16
    f0 = rng.random((n, 11)); f1 = rng.random((n, 11))
    tau = np.array([3, 84, 124, 7], np.int32)
    t = lambda a: torch.from_numpy(a).to(dev)
    out = engine.recording_rows_dev(t(w0), t(w1), t(tau), t(f0), t(f1), t(seg), status_a=t(np.zeros(n, np.int32)),
                                    status_b=t(stb), seg_flags=torch.zeros(4, dtype=torch.int32, device=dev), ctx=ctx)
    got = out.cpu().numpy()
    for s in range(4):
        a, b = seg[s], seg[s + 1]
        keep = (stb[a:b] & (4 | 16)) == 0
        for col, w in ((0, w0), (1, w1)):
            x = w[a:b][keep]
            exp = np.nanmean(x) if np.isfinite(x).any() else np.nan
            assert (np.isnan(exp) and np.isnan(got[s, col])) or got[s, col] == exp, (s, col, got[s, col], exp)
        assert got[s, 2] == tau[s] and got[s, 3] == b - a
        for f in range(11):                     # v2:429-436: np.mean / np.std of the list of per-window values
            assert got[s, 4 + 4 * f] == np.mean(list(f0[a:b, f])) and got[s, 7 + 4 * f] == np.std(list(f1[a:b, f]))
    assert np.isnan(got[2, :2]).all()


def test_ranking_degenerate_key_distributions(ctx):
    """The bucket ranking of the Rips kernels on the distributions that defeat a linear bucket map: every edge the same
    length (one bucket holds all E edges: the quadratic fallback), all lengths zero (duplicate points: the map's scale
    is infinite), two values only, one outlier that stretches the range, and a NaN entry (must not hang; the edge is
    dropped like one beyond the threshold)."""
    rng = np.random.default_rng(12)
    cases = []
    for n in (47, 128):
        d = np.full((n, n), 1.0); np.fill_diagonal(d, 0.0); cases.append(("equal", d))
        cases.append(("zero", np.zeros((n, n))))
        d = np.where(rng.random((n, n)) < 0.5, 0.5, 1.5); d = np.minimum(d, d.T); np.fill_diagonal(d, 0.0); cases.append(("two", d))
        if n == 47:                      # (a random metric on 128 points needs more classes than the n > 64 ladder holds)
            d = rng.random((n, n)) * 1e-3 + 1e-3; d = np.minimum(d, d.T); d[0, 1] = d[1, 0] = 1.9; np.fill_diagonal(d, 0.0)
            cases.append(("outlier", d))
    for name, d in cases:
        h0, h1, st = engine.rips_dm_batch(d[None], thresh=2.0, h1_cap=4096, ctx=ctx)
        o = port.rips_dm(d, thresh=2.0)
        assert st[0] == 0 and _same_multiset(h0[0], o[0]) and _same_multiset(h1[0], o[1]), (name, d.shape)
    # point clouds: all points identical; points on a line with equal spacing (massive ties)
    for pc in (np.ones((60, 3)), np.stack([np.arange(90.0), np.zeros(90), np.zeros(90)], 1)):
        h0, h1, st = engine.cloud_rips_batch(pc[None], normalise=True, thresh=2.0, h1_cap=4096, ctx=ctx)
        o = port.rips_f32(port.cloud_dm(port.minmax_normalise(pc)).astype(np.float32), thresh=2.0)
        assert (st[0] & ~4) == 0 and _same_multiset(h0[0], o[0]) and _same_multiset(h1[0], o[1])
    # NaN: no hang, the other windows of the batch are untouched
    W = port.corr_dist_batch(synth.eeg_windows(3, seed=4))[1]
    W[1, 5, 9] = W[1, 9, 5] = np.nan
    h0, h1, st = engine.rips_dm_batch(W, ctx=ctx)
    for w in (0, 2):
        o = port.rips_dm(W[w])
        assert st[w] == 0 and _same_multiset(h0[w], o[0]) and _same_multiset(h1[w], o[1])
    assert len(h0[1]) >= 1


@pytest.mark.parametrize("merge", [False, True], ids=["per_band", "merged"])
def test_corpus_pass_rows_vs_oracle(ctx, merge):
    """pipeline.CorpusPass -- the loops of run_analysis / process_recording turned inside out (cmp:131-138, 63-122): a
    small corpus (7 recordings x 5 bands x 15 windows, bench.py's generators) through band batches, lanes, HIP graphs
    and the per-pass row block, against the CPU restatement of every (recording, band) group; two passes give
    identical rows, and the widening passes leave no status word set.  merged: all five bands as ONE batch per pass
    (band-major groups; the EEG bands are views of one allocation, the audio bands are concatenated)."""
    import torch
    from oracle import port
    from tda_eeg_audio_amd import pipeline
    dev = torch.device("cuda", ctx.device)
    n_rec, wpr, bands = 7, 15, synth.BANDS
    recs = np.arange(n_rec)
    eeg = synth.corpus_eeg_dev(recs, wpr, len(bands), dev, seed=9)
    aud_all = synth.corpus_audio(n_rec, wpr, bands, seed=11)
    aud = [torch.from_numpy(np.ascontiguousarray(aud_all[b].reshape(-1, 250))).to(dev) for b in bands]
    ctx.set_class_words(1, 1)
    try:
        runner = pipeline.CorpusPass(eeg, aud, wpr, dev, ctx, depth=3, graph=True, merge_bands=merge)
        assert len(runner.batches) == (1 if merge else len(bands))
        runner.step()
        first = runner.finish().clone()
        runner.step(); runner.step()
        rows = runner.finish()
        torch.cuda.synchronize()
        if merge:                                  # bands that are separate allocations are concatenated: same rows
            again = pipeline.CorpusPass([e.clone() for e in eeg], aud, wpr, dev, ctx, depth=2, graph=False,
                                        merge_bands=True)
            assert again.eeg[0].data_ptr() != eeg[0].data_ptr() and runner.eeg[0].data_ptr() == eeg[0].data_ptr()
            again.step()
            assert torch.equal(again.finish(), rows)
    finally:
        ctx.set_class_words(2, 1)
    assert rows.shape == (n_rec, len(bands) * 48) and torch.equal(rows, first)
    for ws in runner.lanes.ws:
        assert int(ws.eeg.status.max()) == 0 and int((ws.aud.status & ~4).max()) == 0
    got = rows.cpu().numpy().reshape(n_rec, len(bands), 48)
    for bi, b in enumerate(bands):
        e = eeg[bi].cpu().numpy(); a = aud[bi].cpu().numpy()
        for r in range(n_rec):
            ref = port.segment_step(e[r * wpr:(r + 1) * wpr], a[r * wpr:(r + 1) * wpr])
            assert np.abs(got[r, bi, :2] - ref[:2]).max() < 1e-6, (b, r)           # Wasserstein means (north_star bar)
            assert np.array_equal(got[r, bi, 2:4], ref[2:4])                        # tau, window count
            assert np.allclose(got[r, bi, 4:], ref[4:], rtol=1e-9, atol=1e-12)


def test_api_edge_cases_empty_truncated_oversized_and_bad_arguments(ctx):
    """The error conventions of SURVEY.md section 8b at the C ABI: empty batches are no-ops, more H1 rows than h1_cap
    is reported per window with the TRUE count (status bit 1) and the rows that fit, a tau that would give more than
    TDA_MAX_POINTS points (or tau < 1) is flagged (bit 16) without touching the other windows, < 3 points gives
    [[0,0]],[[0,0]] (bit 4, utils.py:125-126), and bad arguments come back as error codes with a message, never a
    crash."""
    from tda_eeg_audio_amd import _lib
    # empty batches
    h0, h1, st = engine.rips_dm_batch(np.zeros((0, 47, 47)), ctx=ctx)
    assert h0 == [] and h1 == [] and len(st) == 0
    assert engine.corr_dist_batch(np.zeros((0, 47, 250)), want_corr=False, ctx=ctx).shape == (0, 47, 47)
    assert engine.tau_batch(np.zeros((0, 250)), 125, ctx=ctx).shape == (0,)
    assert engine.wasserstein_batch(*engine.pack_diagrams([]), *engine.pack_diagrams([]), ctx=ctx).shape == (0,)
    # H1 capacity: true count reported, first rows kept
    W = synth.eeg_windows(4, seed=8, kind="white")
    dist = engine.corr_dist_batch(W, want_corr=False, ctx=ctx)
    full0, full1, st_full = engine.rips_dm_batch(dist, h1_cap=1024, ctx=ctx)
    a0, c0, a1, c1, st = engine.rips_dm_batch(dist, h1_cap=8, ctx=ctx, raw=True)
    assert (st_full == 0).all() and (st & 1).all() and not (st & 2).any()
    assert np.array_equal(c1, [len(d) for d in full1]) and all(len(d) > 8 for d in full1)
    assert all(_same_multiset(x, y) for x, y in zip([a0[i, :c0[i]] for i in range(4)], full0))
    # clouds: tau = 0 (more points than the kernel takes / not a delay), tau = 124 (one point), ordinary
    aw = synth.audio_windows(3, "alpha", seed=2)
    h0, h1, npts, st = engine.takens_rips_batch(aw, np.array([0, 124, 7], np.int32), ctx=ctx)
    assert st[0] & 16 and len(h0[0]) == 0 and len(h1[0]) == 0
    assert st[1] == 4 and npts[1] == 1 and np.array_equal(h0[1], [[0.0, 0.0]]) and np.array_equal(h1[1], [[0.0, 0.0]])
    o = port.audio_persistence(aw[2], 7)[0]
    assert st[2] == 0 and _same_multiset(h0[2], o[0]) and _same_multiset(h1[2], o[1])
    # bad arguments: error code + message, the context stays usable
    with pytest.raises(_lib.TdaError, match="n must be in"):
        engine.rips_dm_batch(np.zeros((1, 200, 200)), ctx=ctx)
    with pytest.raises(_lib.TdaError, match="class words"):
        ctx.set_class_words(3, 1)
    with pytest.raises(_lib.TdaError, match="33..48 channels"):
        import torch
        engine.eeg_window_dev(torch.zeros((2, 20, 250), dtype=torch.float64, device=torch.device("cuda", ctx.device)), ctx=ctx)
    with pytest.raises(ValueError, match="Unknown method"):
        engine.corr_to_dist_batch(np.zeros((1, 4, 4)), method="cosine", ctx=ctx)
    h0, h1, st = engine.rips_dm_batch(dist[:1], ctx=ctx)
    assert st[0] == 0
