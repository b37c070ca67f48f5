"""GPU tests of the reference-named mirrors (tda_eeg_audio_amd.utils / graphs / drivers): same
call shapes as scripts/utils.py and nb2, results checked against the oracle and the golden vectors."""
import os

import numpy as np
import pytest

from oracle import brute, port
from tda_eeg_audio_amd import drivers, graphs, synth, utils

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(brute.sort_rows(a), brute.sort_rows(b))


def test_graphs_functions(ctx, golden, tmp_path):
    W = golden["cd_windows"]
    c = graphs.compute_correlation_matrix(W[1])
    d = graphs.correlation_to_distance(c, method="euclidean")
    assert np.abs(c - golden["cd_corr"][1]).max() < 1e-12
    assert np.array_equal(d.astype(np.float32), golden["cd_dist"][1].astype(np.float32))
    r = golden["cd_corr"][0]
    assert np.allclose(graphs.correlation_to_distance(r, "abs"), np.where(np.eye(47, dtype=bool), 0, 1 - np.abs(r)), atol=1e-15)
    assert np.allclose(graphs.correlation_to_distance(r, "standard"), np.where(np.eye(47, dtype=bool), 0, 1 - r), atol=1e-15)
    assert np.allclose(graphs.correlation_to_distance(r, "sqrt"),
                       np.where(np.eye(47, dtype=bool), 0, np.sqrt(np.maximum(1 - r ** 2, 0))), atol=1e-12)
    with pytest.raises(ValueError):
        graphs.correlation_to_distance(r, "nope")
    # on-disk format round trip (preprocessed/<rec>/<band>.npy -> graphs/<rec>/<band>_*.npy)
    rec = tmp_path / "preprocessed" / "bb01_ut01"
    rec.mkdir(parents=True)
    wins = synth.eeg_windows(7, seed=3)
    np.save(rec / "alpha.npy", wins)
    meta, failed = graphs.batch_process_graphs(tmp_path / "preprocessed", tmp_path / "graphs", ["alpha", "beta"])
    assert failed == [] and meta[0]["bands"]["alpha"]["n_windows"] == 7 and "beta" not in meta[0]["bands"]
    dist = np.load(tmp_path / "graphs" / "bb01_ut01" / "alpha_distances.npy")
    corr = np.load(tmp_path / "graphs" / "bb01_ut01" / "alpha_correlations.npy")
    oc, od = port.corr_dist_batch(wins)
    assert dist.shape == (7, 47, 47) and np.array_equal(dist, od) and np.array_equal(corr, oc)


def test_utils_single_call_mirrors(ctx, golden):
    D = golden["ep_in"]
    keep = D.copy()
    dg = utils.compute_eeg_persistence(D)
    assert np.array_equal(D, keep) and len(dg) == 2 and dg[0].dtype == np.float64
    o = port.rips_f32(golden["ep_dm"].astype(np.float32))
    assert _same(dg[0], o[0]) and _same(dg[1], o[1])
    with pytest.raises(ValueError):
        utils.compute_eeg_persistence(np.zeros((3, 4)))
    s = golden["tau_signals"][3]
    tau = utils.compute_tau(s, max_lag=125)
    assert tau == golden["tau_values"][3]
    pc = utils.takens_embedding(s, utils.TAKENS_DIM, tau, utils.TAKENS_SUBSAMPLE)
    a = utils.compute_audio_persistence(pc)
    o = port.rips_f32(golden["pd_beta"].astype(np.float32))
    assert _same(a[0], o[0]) and _same(a[1], o[1])
    small = utils.compute_audio_persistence(pc[:2])
    assert np.array_equal(small[0], [[0, 0]]) and np.array_equal(small[1], [[0, 0]])
    f = utils.extract_features(golden["ef_in_mixed"])
    assert list(f) == utils.FEATURE_KEYS and isinstance(f["n_features"], int)
    assert np.allclose([f[k] for k in utils.FEATURE_KEYS], golden["ef_out_mixed"], rtol=1e-12)
    w = utils.safe_wasserstein(dg[1], a[1])
    assert abs(w - brute.safe_wasserstein_oracle(dg[1], a[1])) < 1e-6
    assert utils.safe_wasserstein(np.zeros((0, 2)), np.array([[0.25, 1.0]])) == \
        pytest.approx(brute.safe_wasserstein_oracle(np.zeros((0, 2)), np.array([[0.25, 1.0]])), abs=1e-12)
    assert utils.safe_wasserstein(np.array([1.0, 2.0]), np.array([[0.25, 1.0]])) == \
        pytest.approx(0.75 / np.sqrt(2), abs=1e-12)            # 1-D input -> [[0,0]] (utils.py:183-184)


def test_drivers_process_file_features_and_cross_wasserstein(ctx, tmp_path):
    rec = tmp_path / "graphs" / "slow" / "bb01_ut01"
    rec.mkdir(parents=True)
    dists = {}
    for bi, band in enumerate(["delta", "gamma"]):
        W = synth.eeg_windows(50, seed=10 + bi)
        dists[band] = port.corr_dist_batch(W)[1]
        np.save(rec / f"{band}_distances.npy", dists[band])
    feats, meta = drivers.process_file_features(rec, ["delta", "theta", "gamma"], max_windows_per_band=39)
    assert meta["n_windows"] == {"delta": 50, "theta": 0, "gamma": 50} and meta["n_windows_used"]["delta"] == 39
    keys = list(feats)
    assert keys[:4] == ["delta_h0_n_features_mean", "delta_h0_n_features_std", "delta_h1_n_features_mean",
                        "delta_h1_n_features_std"] and len(keys) == 88
    use = drivers.select_windows_md5("bb01_ut01", "gamma", 50, 39)
    f1 = np.array([port.features(port.rips_dm(dists["gamma"][i])[1]) for i in use])
    assert feats["gamma_h1_total_persistence_mean"] == pytest.approx(f1[:, 9].mean(), rel=1e-12)
    assert feats["gamma_h1_total_persistence_std"] == pytest.approx(f1[:, 9].std(), rel=1e-12)
    # matched-vs-mismatched style cross Wasserstein (mvm:87-95)
    eeg = drivers.get_eeg_diagrams(rec, ["delta"])
    assert len(eeg["delta"]) == 15
    aud = drivers.get_audio_diagrams_from_windows(synth.audio_windows(40, "delta", seed=2))
    w = drivers.compute_cross_wasserstein(eeg["delta"], aud)
    ref = np.nanmean([brute.safe_wasserstein_oracle(e[1], a[1]) for e, a in zip(eeg["delta"], aud)])
    assert abs(w - ref) < 1e-6
    assert np.isnan(drivers.compute_cross_wasserstein([], aud))
    # comparison driver on arrays (cmp:63-122)
    out = drivers.process_recording_arrays({"delta": synth.audio_windows(40, "delta", seed=2)}, {"delta": dists["delta"]})
    assert out["delta"]["n_windows"] == 15 and out["delta"]["tau"] >= 1
    assert np.isfinite(out["delta"]["wasserstein_h0"]) and out["delta"]["eeg_h1_features"].shape == (15, 11)


def test_create_dataset_on_graphs_tree(ctx, tmp_path):
    """v2:499-606 on a small synthetic graphs/ tree: row order, 220 columns, window equalisation by
    the global per-band minimum with md5-seeded sampling; values against the oracle."""
    bands = ["delta", "theta", "alpha", "beta", "gamma"]
    recs = {"slow": [("bb01_ut01", 44), ("bb02_ut03", 41)], "fast": [("bb01_ut05", 40), ("bb03_ut02", 43)]}
    store = {}
    for cond, lst in recs.items():
        for name, nwin in lst:
            d = tmp_path / "graphs" / cond / name
            d.mkdir(parents=True)
            for bi, band in enumerate(bands):
                W = synth.eeg_windows(nwin + bi, seed=hash((name, band)) % 10000)
                dm = port.corr_dist_batch(W)[1]
                np.save(d / f"{band}_distances.npy", dm)
                store[(name, band)] = dm
    X, y, subjects, names, filenames, meta = drivers.create_dataset(tmp_path / "graphs" / "slow", tmp_path / "graphs" / "fast")
    assert X.shape == (4, 220) and list(y) == [0, 0, 1, 1]
    assert filenames == ["bb01_ut01", "bb02_ut03", "bb01_ut05", "bb03_ut02"] and list(subjects) == ["bb01", "bb02", "bb01", "bb03"]
    assert names == drivers.feature_names() and np.isfinite(X).all()
    mins = drivers.compute_min_windows_per_band([tmp_path / "graphs" / "slow", tmp_path / "graphs" / "fast"])
    assert mins == {b: 40 + i for i, b in enumerate(bands)}
    assert meta[0]["n_windows_used"] == mins and meta[0]["subject"] == "bb01"
    # one recording-band against the oracle
    name, band, bi = "bb03_ut02", "alpha", 2
    use = drivers.select_windows_md5(name, band, store[(name, band)].shape[0], mins[band])
    f0 = np.array([port.features(port.rips_dm(store[(name, band)][i])[0]) for i in use])
    f1 = np.array([port.features(port.rips_dm(store[(name, band)][i])[1]) for i in use])
    row = X[3, 44 * bi:44 * (bi + 1)]
    exp = np.stack([f0.mean(0), f0.std(0), f1.mean(0), f1.std(0)], 1).ravel()
    assert np.allclose(row, exp, rtol=1e-12, atol=1e-15)
    # batch slicing (BATCH_START/BATCH_END semantics, v2:537-541)
    Xb, yb, _, _, fb, _ = drivers.create_dataset(tmp_path / "graphs" / "slow", tmp_path / "graphs" / "fast",
                                                 batch_start=1, batch_end=3)
    assert fb == filenames[1:3] and np.array_equal(Xb, X[1:3])
    drivers.save_dataset(tmp_path / "features", X, y, subjects, names, filenames)
    assert np.array_equal(np.load(tmp_path / "features" / "X.npy"), X)
    assert (tmp_path / "features" / "feature_names.txt").read_text().split() == names


def test_spearman_matches_scipy(ctx):
    from scipy.stats import spearmanr
    from tda_eeg_audio_amd import engine
    rng = np.random.default_rng(5)
    n_seg, per = 12, 15
    fa = rng.random((n_seg * per, 11)); fb = rng.random((n_seg * per, 11))
    fa[:per, 0] = 3.0                                    # constant series -> r = 0, p = 1 (cmp:113-114)
    fa[per:2 * per, 0] = np.round(fa[per:2 * per, 0] * 4)   # ties -> average ranks
    fb[2 * per:3 * per, 6] = fa[2 * per:3 * per, 6]      # identical -> r = 1
    seg = np.arange(0, n_seg * per + 1, per).astype(np.int32)
    seg = np.concatenate([seg[:-1], [seg[-1] - 12, seg[-1]]]).astype(np.int32)     # a 3- and a 12-window group
    r, p = engine.spearman_batch(fa, fb, seg, ctx=ctx)
    for s in range(len(seg) - 1):
        a, b = seg[s], seg[s + 1]
        for k, c in enumerate(engine.SPEARMAN_COLS):
            x, y = fa[a:b, c], fb[a:b, c]
            if len(x) >= 5 and np.std(x) > 1e-10 and np.std(y) > 1e-10:
                rr, pp = spearmanr(x, y)
            else:
                rr, pp = 0.0, 1.0
            assert abs(r[s, k] - rr) < 1e-12 and abs(p[s, k] - pp) < 1e-10, (s, k, r[s, k], rr, p[s, k], pp)


def test_process_recording_file_and_detailed_csv(ctx, tmp_path):
    """cmp:45-157 on the reference's layout: data/<cond>/<rec>.mat (audio track `y`) + graphs/<cond>/<rec>/ ->
    the rows of eeg_audio_tda_detailed.csv; values against process_recording_arrays on the same inputs."""
    import scipy.io as sio
    from tda_eeg_audio_amd import preprocess
    rng = np.random.default_rng(11)
    recs = [("slow", "bb01_ut01"), ("fast", "bb02_ut04")]
    for cond, name in recs:
        (tmp_path / "data" / cond).mkdir(parents=True, exist_ok=True)
        n = int(44100 * 4.2)
        t = np.arange(n) / 44100.0
        y = (np.sin(2 * np.pi * 3.1 * t) + 0.5 * rng.standard_normal(n)) * (1 + 0.5 * np.sin(2 * np.pi * 0.7 * t))
        sio.savemat(str(tmp_path / "data" / cond / f"{name}.mat"), {"y": np.stack([y, y * 0.5], axis=1)})   # stereo
        d = tmp_path / "graphs" / cond / name
        d.mkdir(parents=True)
        for bi, band in enumerate(drivers.BANDS[:3]):                  # two bands have no graph file (cmp:67-69)
            np.save(d / f"{band}_distances.npy", port.corr_dist_batch(synth.eeg_windows(14 + bi, seed=bi + len(name)))[1])
    (tmp_path / "data" / "slow" / "orphan.mat").write_bytes(b"")          # no graph dir: skipped (cmp:48-49)
    results, df = drivers.run_analysis(tmp_path / "data", tmp_path / "graphs", out_csv=tmp_path / "results" / "detailed.csv")
    assert [r["filename"] for r in results] == ["bb01_ut01.mat", "bb02_ut04.mat"]
    assert list(df.columns) == drivers.DETAILED_COLUMNS and len(df) == 6
    assert list(df["band"][:3]) == drivers.BANDS[:3] and set(df["n_windows"]) <= {13, 14, 15}
    assert (tmp_path / "results" / "detailed.csv").read_text().splitlines()[0] == ",".join(drivers.DETAILED_COLUMNS)
    # same numbers as the array-level driver on the same audio
    y = sio.loadmat(str(tmp_path / "data" / "slow" / "bb01_ut01.mat"))["y"].mean(axis=1)
    aw = preprocess.audio_to_band_windows(y.astype(np.float64), 44100)
    dists = {b: np.load(tmp_path / "graphs" / "slow" / "bb01_ut01" / f"{b}_distances.npy") for b in drivers.BANDS[:3]}
    direct = drivers.process_recording_arrays(aw, dists)
    for b in drivers.BANDS[:3]:
        row = df[(df.filename == "bb01_ut01.mat") & (df.band == b)].iloc[0]
        assert row.wasserstein_h0 == direct[b]["wasserstein_h0"] and row.tau == direct[b]["tau"]
        assert np.isfinite(row.wasserstein_h1) and 0.0 <= row.corr_n_features_p <= 1.0


def test_validation_issues_reach_the_metadata_files(ctx, tmp_path):
    """v2:380-382 + v2:684-688: the first window of a band is validated, the outcome lands in the recording's
    metadata, and save_dataset writes metadata.csv / metadata.json next to X.npy."""
    import json
    for cond, name in (("slow", "bb01_ut01"), ("fast", "bb01_ut02")):
        d = tmp_path / "graphs" / cond / name
        d.mkdir(parents=True)
        for band in drivers.BANDS:
            dm = port.corr_dist_batch(synth.eeg_windows(8, seed=len(band)))[1]
            if (cond, band) == ("fast", "theta"):
                dm[0, 3, 3] = 0.25                         # non-zero diagonal in the first window
                dm[0, 2, 5] += 1e-3                        # and an asymmetry
            np.save(d / f"{band}_distances.npy", dm)
    feats, md = drivers.process_file_features(tmp_path / "graphs" / "fast" / "bb01_ut02")
    assert md["validation_issues"] == ["theta: No simétrica: asimetría máxima=0.001000", "theta: Diagonal no cero: max=0.250000"]
    X, y, subjects, names, filenames, meta = drivers.create_dataset(tmp_path / "graphs" / "slow", tmp_path / "graphs" / "fast")
    assert meta[0]["validation_issues"] == [] and meta[1]["validation_issues"] == md["validation_issues"]
    drivers.save_dataset(tmp_path / "features", X, y, subjects, names, filenames, meta)
    saved = json.load(open(tmp_path / "features" / "metadata.json"))
    assert saved[1]["validation_issues"] == md["validation_issues"] and saved[0]["filename"] == "bb01_ut01"
    head = (tmp_path / "features" / "metadata.csv").read_text().splitlines()[0].split(",")
    assert head[:3] == ["n_windows", "n_windows_used", "validation_issues"] and "filename" in head and "label" in head


_DS_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from tda_eeg_audio_amd import drivers
from tda_eeg_audio_amd import dist as tdist
rank, world, local = tdist.init_from_env(backend="gloo")       # two ranks share the one GPU of the test box
torch.cuda.set_device(0)
root = sys.argv[2]
X, y, subjects, names, filenames, meta = drivers.create_dataset(root + "/graphs/slow", root + "/graphs/fast",
                                                                rank=rank, world_size=world)
ref = np.load(root + "/X_single.npy")
assert np.array_equal(X, ref), np.abs(X - ref).max()
assert [m["filename"] for m in meta] == filenames and len(meta) == len(filenames) == 5
if rank == 0: print("DATASET_OK")
dist.destroy_process_group()
"""


def test_create_dataset_two_ranks_equals_one(ctx, tmp_path):
    """The multi-GPU replacement of the BATCH_START/BATCH_END partial files (v2:55-60, 608-638): recordings dealt
    over two ranks, rows all-gathered from device memory, metadata of every rank collected -- identical to the
    single-process result.  (gloo here: both ranks use the one GPU of the test box; RCCL is the driver's 8-GPU run.)"""
    import subprocess, sys
    for cond, names_ in (("slow", ["bb01_ut01", "bb02_ut01", "bb03_ut01"]), ("fast", ["bb01_ut02", "bb02_ut02"])):
        for k, name in enumerate(names_):
            d = tmp_path / "graphs" / cond / name
            d.mkdir(parents=True)
            for bi, band in enumerate(drivers.BANDS):
                np.save(d / f"{band}_distances.npy", port.corr_dist_batch(synth.eeg_windows(9 + k + bi, seed=17 * k + bi))[1])
    X = drivers.create_dataset(tmp_path / "graphs" / "slow", tmp_path / "graphs" / "fast")[0]
    np.save(tmp_path / "X_single.npy", X)
    script = tmp_path / "worker.py"
    script.write_text(_DS_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 200), str(script), root, str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0 and "DATASET_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_utils_batch_runs_the_reference_loop_unchanged(ctx):
    """The per-window loop of process_recording (cmp:88-99), character for character, inside ``with utils.batch():`` --
    one launch per stage on leaving the block instead of six host round trips per window -- gives the values of the
    immediate calls (lists of floats / dicts afterwards, NaN and [[0, 0]] semantics included) and is several times faster."""
    import time
    from tda_eeg_audio_amd import engine, synth, utils
    from tda_eeg_audio_amd.utils import (TAKENS_DIM, TAKENS_SUBSAMPLE, compute_audio_persistence, compute_eeg_persistence,
                                         compute_tau, extract_features, safe_wasserstein, takens_embedding)
    audio_wins = synth.audio_windows(16, "alpha", seed=77)
    eeg_dists = engine.corr_dist_batch(synth.eeg_windows(16, seed=78), want_corr=False, ctx=ctx)
    idx = np.linspace(0, 15, 15, dtype=int)
    tau = compute_tau(audio_wins[idx[0]], max_lag=125)

    def loop():
        wass_h0, wass_h1, audio_feat_ts, eeg_feat_ts = [], [], [], []
        for w in idx:
            pc = takens_embedding(audio_wins[w], TAKENS_DIM, tau, TAKENS_SUBSAMPLE)
            if len(pc) < 3:
                continue
            a_dgms = compute_audio_persistence(pc)
            e_dgms = compute_eeg_persistence(eeg_dists[w])
            wass_h0.append(safe_wasserstein(e_dgms[0], a_dgms[0]))
            wass_h1.append(safe_wasserstein(e_dgms[1], a_dgms[1]))
            audio_feat_ts.append(extract_features(a_dgms[1]))
            eeg_feat_ts.append(extract_features(e_dgms[1]))
        return wass_h0, wass_h1, audio_feat_ts, eeg_feat_ts
    loop()                                                # warm
    t0 = time.perf_counter(); ref = loop(); t_ref = time.perf_counter() - t0
    with utils.batch():
        loop()
    t0 = time.perf_counter()
    with utils.batch():
        got = loop()
    t_got = time.perf_counter() - t0
    assert len(got[0]) == len(ref[0]) == 15
    assert float(np.nanmean(got[0])) == float(np.nanmean(ref[0])) and float(np.nanmean(got[1])) == float(np.nanmean(ref[1]))
    for k in range(15):
        assert float(got[0][k]) == ref[0][k] and float(got[1][k]) == ref[1][k]
        for feat in ("mean_persistence", "total_persistence", "persistence_entropy", "max_persistence", "n_features"):
            assert got[2][k][feat] == ref[2][k][feat] and got[3][k][feat] == ref[3][k][feat]
        assert dict(got[3][k].items()) == ref[3][k]
    a_ts = [f["n_features"] for f in got[2]]
    assert all(isinstance(x, int) for x in a_ts)
    print(f"per-call loop {t_ref * 1e3:.1f} ms, batched {t_got * 1e3:.1f} ms: x{t_ref / t_got:.1f}")
    assert t_ref / t_got > 4.0
    # values are usable inside the block too (the first use flushes what is queued), and errors keep their types
    with utils.batch():
        d = compute_eeg_persistence(eeg_dists[0])
        assert np.asarray(d[0]).shape == (47, 2)
        assert np.array_equal(np.asarray(d[1]), np.asarray(compute_eeg_persistence(eeg_dists[0])[1]))
        with pytest.raises(ValueError):
            compute_eeg_persistence(np.zeros((3, 4)))
