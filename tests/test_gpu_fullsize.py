"""
Full-size GPU tests (BASELINE.json configs 3-5) through size-independent properties, plus random
samples against the oracle.  Inputs are generated on the GPU (there is no corpus: data/, graphs/
are git-ignored in the reference), results stay in HBM; only samples and reductions come back.

  config 3: 1,416 recordings x 5 bands x 39 windows (v2's min-equalised count, v2:519-527)
            = 276,120 EEG windows -> corr/dist -> Rips H0/H1 -> features -> (1416, 220)
  config 4: 1,416 x 5 = 7,080 audio windows -> tau -> Takens -> Rips
  config 5: 7,080 matched + 7,080 mismatched EEG/audio diagram pairs, H0 and H1
"""
import numpy as np
import pytest

from oracle import brute, port
from tda_eeg_audio_amd import engine, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

N_REC, N_BANDS, WPR = 1416, 5, 39


def _same(a, b):
    return np.array_equal(brute.sort_rows(a), brute.sort_rows(b))


def _single_linkage_deaths(dm64):
    """H0 deaths from scipy (neither our code nor ripser): single-linkage merge heights on the float32 distances,
    zero heights dropped -- selected values, exact."""
    from scipy.cluster.hierarchy import linkage
    from scipy.spatial.distance import squareform
    d = dm64.astype(np.float32).astype(np.float64)
    np.fill_diagonal(d, 0.0)
    h = np.sort(linkage(squareform(d, checks=False), method="single")[:, 2])
    return h[h > 0]


def _eeg_band_on_gpu(band, dev):
    g = torch.Generator(device=dev); g.manual_seed(1000 + band)
    A = torch.randn((N_REC, 1, 47, 8), generator=g, device=dev, dtype=torch.float64)
    S = torch.randn((N_REC, WPR, 8, 250), generator=g, device=dev, dtype=torch.float64)
    E = torch.randn((N_REC, WPR, 47, 250), generator=g, device=dev, dtype=torch.float64)
    return (torch.matmul(A, S) + 0.5 * E).reshape(N_REC * WPR, 47, 250).contiguous()


def test_config3_full_corpus_features(ctx):
    dev = torch.device("cuda", 0)
    feats = torch.empty((N_REC, N_BANDS * 44), dtype=torch.float64, device=dev)
    seg = torch.arange(0, N_REC * WPR + 1, WPR, dtype=torch.int32, device=dev)
    rng = np.random.default_rng(0)
    for band in range(N_BANDS):
        X = _eeg_band_on_gpu(band, dev)
        n = X.shape[0]
        dist = engine.corr_dist_dev(X, ctx=ctx)
        dg = engine.rips_dm_dev(dist, ctx=ctx)
        f0 = engine.features_dev(dg.h0, dg.c0, ctx=ctx)
        f1 = engine.features_dev(dg.h1, dg.c1, ctx=ctx)
        agg = engine.aggregate_dev(f0, f1, seg, ctx=ctx)
        feats[:, band * 44:(band + 1) * 44] = agg
        torch.cuda.synchronize()
        # ---- properties that hold for every window ----
        assert int(dg.status.abs().max()) == 0
        assert bool((dg.c0 == 47).all())
        assert bool(torch.isinf(dg.h0[:, 46, 1]).all()) and bool((dg.h0[:, :, 0] == 0).all())
        assert bool((dg.h0[:, 1:46, 1] >= dg.h0[:, :45, 1]).all())           # ascending deaths
        k1 = dg.c1.long()
        col = torch.arange(dg.h1_cap, device=dev)[None, :]
        valid = col < k1[:, None]
        b, d = dg.h1[:, :, 0], dg.h1[:, :, 1]
        assert bool(((d > b) | ~valid).all()) and bool((torch.isfinite(d) | ~valid).all())
        assert bool(((b[:, 1:] <= b[:, :-1]) | ~valid[:, 1:]).all())          # descending births
        assert bool((f0[:, 0] == 46).all()) and bool((f0[:, 1] == 1).all())   # n_features, n_essential
        assert bool((f1[:, 0] == k1.double()).all()) and bool((f1[:, 1] == 0).all())
        # (c/s_i)/s_j vs (c/s_j)/s_i: symmetric to the last bits only, exactly like np.corrcoef
        assert float((dist - dist.transpose(1, 2)).abs().max()) < 1e-14
        assert bool((torch.diagonal(dist, dim1=1, dim2=2) == 0).all())
        # every birth/death is one of the window's float32 distances (values are selected, never computed)
        pick = torch.from_numpy(rng.choice(n, 256, replace=False)).to(dev)
        d32 = dist[pick].float().double().reshape(256, -1)
        for rows, cnt in ((dg.h0[pick, :46, 1], None), (dg.h1[pick], dg.c1[pick])):
            vals = rows.reshape(256, -1)
            ok = (vals[:, :, None] == d32[:, None, :]).any(-1)
            if cnt is not None:
                m = (torch.arange(dg.h1_cap, device=dev)[None, :] < cnt[:, None].long()).repeat_interleave(2, dim=1)
                ok = ok | ~m
            assert bool(ok.all())
        # idempotence: a second run is bit-identical
        dg2 = engine.rips_dm_dev(dist, ctx=ctx)
        torch.cuda.synchronize()
        assert torch.equal(dg2.c1, dg.c1) and torch.equal(dg2.h0, dg.h0)
        assert bool(((dg2.h1 == dg.h1) | ~valid[:, :, None]).all())
        # relabelling the electrodes does not change the diagrams (as multisets)
        perm = torch.from_numpy(rng.permutation(47)).to(dev)
        sub = dist[pick[:64]]
        dgp = engine.rips_dm_dev(sub[:, perm][:, :, perm].contiguous(), ctx=ctx)
        torch.cuda.synchronize()
        assert torch.equal(dgp.c1, dg.c1[pick[:64]])
        assert torch.equal(torch.sort(dgp.h0[:, :, 1], dim=1)[0], torch.sort(dg.h0[pick[:64], :, 1], dim=1)[0])
        # ---- a random sample against the oracle, bit-exact ----
        samp = rng.choice(n, 60, replace=False)
        Xs = X[torch.from_numpy(samp).to(dev)].cpu().numpy()
        h0s, h1s = dg.h0[torch.from_numpy(samp).to(dev)].cpu().numpy(), dg.h1[torch.from_numpy(samp).to(dev)].cpu().numpy()
        c1s = dg.c1[torch.from_numpy(samp).to(dev)].cpu().numpy()
        ds = dist[torch.from_numpy(samp).to(dev)].cpu().numpy()
        for i in range(len(samp)):
            oc, od = port.corr_dist(Xs[i])
            assert np.array_equal(od, ds[i])
            o = port.rips_dm(od)
            assert _same(h0s[i], o[0]) and _same(h1s[i, :c1s[i]], o[1])
        # ---- and a larger sample of H0 against scipy's single linkage (an implementation that is not ours) ----
        samp2 = torch.from_numpy(rng.choice(n, 400, replace=False)).to(dev)
        d2 = dist[samp2].cpu().numpy(); h2 = dg.h0[samp2].cpu().numpy()
        for i in range(len(d2)):
            assert np.array_equal(h2[i, :46, 1], _single_linkage_deaths((d2[i] + d2[i].T) / 2.0))
        del X, dist, dg, dg2
    feats = feats.cpu().numpy()
    assert feats.shape == (1416, 220) and np.isfinite(feats).all()
    assert np.all(feats[:, 0::44] == 46.0) and np.all(feats[:, 1::44] == 0.0)     # <band>_h0_n_features mean / std


def test_config4_audio_and_config5_pairs(ctx):
    dev = torch.device("cuda", 0)
    n_per_band = N_REC
    wins, band_id = synth.audio_windows_all_bands(n_per_band, seed=99)             # (7080, 250)
    W = torch.from_numpy(wins).to(dev)
    tau = engine.tau_dev(W, 125, ctx=ctx)
    aud = engine.takens_rips_dev(W, tau, ctx=ctx)
    torch.cuda.synchronize()
    tau_h = tau.cpu().numpy()
    assert np.array_equal(tau_h[:200], [port.compute_tau(w, 125) for w in wins[:200]])
    P = aud.n_points.cpu().numpy()
    assert np.array_equal(P, (250 - 2 * tau_h + 1) // 2)
    assert int(aud.status.abs().max()) == 0
    c0 = aud.c0.cpu().numpy()
    assert np.all(c0 <= P) and np.all(c0 >= P - 2)           # P-1 merges (+1 essential), coincident points dropped
    h0 = aud.h0.cpu().numpy(); h1 = aud.h1.cpu().numpy(); c1 = aud.c1.cpu().numpy()
    # the normalised cloud lies in the unit cube: no finite H0 death beyond its diagonal (essential rows are +inf)
    fin = (np.arange(h0.shape[1])[None, :] < c0[:, None]) & np.isfinite(h0[:, :, 1])
    assert np.all(h0[:, :, 1][fin] <= float(np.float32(np.sqrt(3.0))))
    rng = np.random.default_rng(1)
    for w in rng.choice(len(wins), 80, replace=False):
        (o0, o1), Pw = port.audio_persistence(wins[w], int(tau_h[w]))
        assert Pw == P[w] and _same(h0[w, :c0[w]], o0) and _same(h1[w, :c1[w]], o1), w
    for w in rng.choice(len(wins), 300, replace=False):     # H0 against scipy's single linkage
        pc = port.minmax_normalise(port.takens(wins[w], 3, int(tau_h[w]), 2))
        ref = _single_linkage_deaths(port.cloud_dm(pc))
        assert np.array_equal(h0[w, :c0[w] - 1, 1], ref) and np.isinf(h0[w, c0[w] - 1, 1]), w
    # ---- config 5: matched and mismatched pairs against 7,080 EEG diagrams ----
    g = torch.Generator(device=dev); g.manual_seed(7)
    n = len(wins)
    A = torch.randn((n // 15, 1, 47, 8), generator=g, device=dev, dtype=torch.float64)
    S = torch.randn((n // 15, 15, 8, 250), generator=g, device=dev, dtype=torch.float64)
    E = torch.randn((n // 15, 15, 47, 250), generator=g, device=dev, dtype=torch.float64)
    X = (torch.matmul(A, S) + 0.5 * E).reshape(-1, 47, 250).contiguous()
    eeg = engine.rips_dm_dev(engine.corr_dist_dev(X, ctx=ctx), ctx=ctx)
    idx = torch.arange(n, dtype=torch.int32, device=dev)
    mism = torch.roll(idx, 15)                                 # window w of another recording
    out = {}
    for name, (ea, ec, aa, ac) in {"h0": (eeg.h0, eeg.c0, aud.h0, aud.c0), "h1": (eeg.h1, eeg.c1, aud.h1, aud.c1)}.items():
        wm, sm = engine.wasserstein_dev(ea, ec, aa, ac, idx, idx, ctx=ctx)
        wx, sx = engine.wasserstein_dev(ea, ec, aa, ac, idx, mism, ctx=ctx)
        wr, sr = engine.wasserstein_dev(aa, ac, ea, ec, idx, idx, ctx=ctx)          # symmetry
        ws, ss = engine.wasserstein_dev(ea, ec, ea, ec, idx, idx, ctx=ctx)          # identity
        torch.cuda.synchronize()
        assert int(sm.abs().max()) == 0 and int(sx.abs().max()) == 0
        assert float((wm - wr).abs().max()) < 1e-9
        assert float(ws.abs().max()) < 1e-9
        assert bool((wm >= 0).all()) and bool(torch.isfinite(wm).all()) and bool(torch.isfinite(wx).all())
        out[name] = (wm.cpu().numpy(), wx.cpu().numpy())
    e0, e1 = eeg.to_lists()
    a0, a1 = aud.to_lists()
    mism_h = mism.cpu().numpy()
    for w in rng.choice(n, 60, replace=False):
        assert abs(out["h0"][0][w] - brute.safe_wasserstein_oracle(e0[w], a0[w])) < 1e-6
        assert abs(out["h1"][0][w] - brute.safe_wasserstein_oracle(e1[w], a1[w])) < 1e-6
        assert abs(out["h1"][1][w] - brute.safe_wasserstein_oracle(e1[w], a1[mism_h[w]])) < 1e-6
    # triangle inequality on a sample of triples (EEG H1 diagrams)
    t = torch.from_numpy(rng.choice(n, (200, 3))).to(dev).int()
    dab, _ = engine.wasserstein_dev(eeg.h1, eeg.c1, eeg.h1, eeg.c1, t[:, 0].contiguous(), t[:, 1].contiguous(), ctx=ctx)
    dbc, _ = engine.wasserstein_dev(eeg.h1, eeg.c1, eeg.h1, eeg.c1, t[:, 1].contiguous(), t[:, 2].contiguous(), ctx=ctx)
    dac, _ = engine.wasserstein_dev(eeg.h1, eeg.c1, eeg.h1, eeg.c1, t[:, 0].contiguous(), t[:, 2].contiguous(), ctx=ctx)
    torch.cuda.synchronize()
    assert bool((dac <= dab + dbc + 1e-9).all())


def test_bench_two_rank_rehearsal_and_corpus_rows(tmp_path):
    """`python bench.py --gpus 2` starts its two ranks itself (here both on the one GPU of the box, all-gather over
    gloo: --share-gpu): recordings dealt by dist.shard_recordings, one all-gather of the (n_rec, 5 x 48) rows per
    pass, ONE JSON line from rank 0 with n_gpus = 2 and strong scaling over the same fixed corpus."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--recordings", "36", "--steps", "2",
           "--warmup", "1", "--no-cpu", "--features-steps", "1", "--dump-rows", str(tmp_path / "rows2.npy")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["unit"] == "windows/s"
    assert d["config"]["windows_per_pass"] == 36 * 5 * 15 and d["config"]["windows_per_gpu_per_pass"] == 18 * 5 * 15
    assert d["config"]["result_rows_finite_frac"] == 1.0
    assert d["features_pass"]["matrix_shape"] == [36, 220] and d["features_pass"]["matrix_finite"]
    assert d["roofline"]["frac"] > 0 and d["roofline"]["bound"] == "hbm"
    assert len(d["config"]["ms_per_step_per_rank"]) == 2 and d["config"]["allgather_ms"] is not None
    # the rows two ranks gather equal the rows of one rank on the same corpus, bit for bit
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--recordings", "36", "--steps", "1", "--warmup", "1",
                         "--no-cpu", "--no-extras", "--dump-rows", str(tmp_path / "rows1.npy")], env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stdout[-1500:] + r1.stderr[-1500:]
    assert np.array_equal(np.load(tmp_path / "rows1.npy"), np.load(tmp_path / "rows2.npy"))


def test_two_gpus_rccl_rows_equal_one_rank(tmp_path):
    """The RCCL branch (one process per GPU, device all_gather_into_tensor over xGMI, graphs replayed next to the
    collective): `bench.py --gpus 2` against `--gpus 1` on the same corpus -- the gathered rows must be equal bit for
    bit.  Needs two GPUs: skipped on the one-GPU test box, runs wherever the driver has more."""
    import json, os, subprocess, sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = {}
    for n in (1, 2):
        path = str(tmp_path / f"rows{n}.npy")
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--recordings", "64", "--steps", "2", "--warmup", "1",
               "--no-cpu", "--no-extras", "--dump-rows", path]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        assert d["n_gpus"] == n and d["config"]["result_rows_finite_frac"] == 1.0
        out[n] = (np.load(path), d)
    assert out[1][0].shape == out[2][0].shape == (64, 5 * 48)
    assert np.array_equal(out[1][0], out[2][0])
    assert out[2][1]["config"]["allgather_ms"] is not None and len(out[2][1]["config"]["ms_per_step_per_rank"]) == 2
