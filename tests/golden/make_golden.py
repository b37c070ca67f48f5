#!/usr/bin/env python3
"""
tests/golden/make_golden.py -- captures golden vectors from the REFERENCE's own
numpy/scipy-only functions.  Run once in the build container (where
/root/reference is mounted); the resulting ``reference_golden.npz`` is committed
and is the only thing that travels to the GPU box.

What is executed from the reference (nothing is copied into this repo):
  * scripts/utils.py, imported as a module.  Its two third-party imports that are
    not installed anywhere in this image (``ripser``, ``persim`` --
    requirements.txt:5-6) are satisfied by *recording* stand-ins that compute
    nothing: they store the arguments the reference hands to them.  That pins the
    reference's pre-processing in front of ripser (utils.py:127-131, :137-140) and
    in front of persim (utils.py:182-189), and the keyword arguments it passes.
  * the two function definitions of notebooks/2_graph_construction.ipynb cell 4
    (compute_correlation_matrix, correlation_to_distance; nb2:86-122), exec'd from
    the notebook JSON at generation time.
Values downstream of ripser/persim themselves cannot be captured (PARITY UNPINNED
there; see oracle/tda_oracle.c header).
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

REF = os.environ.get("TDA_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_golden.npz")

calls = {"ripser": [], "wasserstein": []}


def _fake_ripser(X, **kw):
    calls["ripser"].append((np.array(X, copy=True), dict(kw)))
    return {"dgms": [np.zeros((0, 2)), np.zeros((0, 2))]}


def _fake_wasserstein(a, b, **kw):
    calls["wasserstein"].append((np.array(a, copy=True), np.array(b, copy=True)))
    return 0.0


def main():
    m_r = types.ModuleType("ripser"); m_r.ripser = _fake_ripser
    m_p = types.ModuleType("persim"); m_p.wasserstein = _fake_wasserstein
    sys.modules["ripser"] = m_r
    sys.modules["persim"] = m_p
    sys.path.insert(0, os.path.join(REF, "scripts"))
    import utils as ref  # the reference module

    nb = json.load(open(os.path.join(REF, "notebooks", "2_graph_construction.ipynb")))
    src = "".join(nb["cells"][4]["source"])
    ns = {"np": np, "N_ELECTRODES": 47}
    exec(compile(src, "nb2-cell4", "exec"), ns)   # defines the two functions (+ a print-only self test)
    corr_fn, dist_fn = ns["compute_correlation_matrix"], ns["correlation_to_distance"]

    G = {}
    rng = np.random.default_rng(20240229)

    # ---- (1) corr -> dist, incl. zero-variance and duplicated channels ----
    W = np.empty((4, 47, 250))
    W[0] = rng.standard_normal((47, 250))
    A = rng.standard_normal((47, 8)); S = rng.standard_normal((8, 250))
    W[1] = A @ S + 0.5 * rng.standard_normal((47, 250))
    W[2] = W[1]; W[2, 5] = 3.25            # constant channel -> NaN -> 0 -> d = sqrt(2)
    W[3] = W[0]; W[3, 11] = W[3, 7]        # duplicated channel -> r = 1 -> d = 0
    W[3, 20] = -2.0 * W[3, 30]             # anti-correlated -> r = -1 -> d = 2
    import logging
    logging.disable(logging.WARNING)
    with np.errstate(all="ignore"):
        corr = np.stack([corr_fn(w) for w in W])
        dist = np.stack([dist_fn(c, method="euclidean") for c in corr])
    G["cd_windows"] = W; G["cd_corr"] = corr; G["cd_dist"] = dist

    # ---- (2) create_windows ----
    sig = rng.standard_normal(4606)
    wins = ref.create_windows(sig, 250, 62)
    G["cw_signal"] = sig; G["cw_windows"] = wins
    G["cw_empty_shape"] = np.array(ref.create_windows(sig[:100], 250, 62).shape)

    # ---- (3)+(4) tau and Takens on band-limited noise, one per band ----
    from scipy import signal as sp
    taus, tsig, pcs = [], [], {}
    for bi, (name, (lo, hi)) in enumerate(ref.FREQ_BANDS.items()):
        b, a = sp.butter(4, [lo / 125.0, hi / 125.0], btype="band")
        x = sp.filtfilt(b, a, rng.standard_normal(2000))[800:1050].copy()
        tau = ref.compute_tau(x, max_lag=125)
        taus.append(tau); tsig.append(x)
        pcs[name] = ref.takens_embedding(x, ref.TAKENS_DIM, tau, ref.TAKENS_SUBSAMPLE)
        G["tk_pc_" + name] = pcs[name]
    G["tau_signals"] = np.stack(tsig); G["tau_values"] = np.array(taus)
    const = np.full(250, 1.5)
    ramp = np.arange(250.0)
    G["tau_const"] = np.array([ref.compute_tau(const, max_lag=125)])
    G["tau_ramp"] = np.array([ref.compute_tau(ramp, max_lag=125), ref.compute_tau(ramp)])
    G["tk_empty_shape"] = np.array(ref.takens_embedding(tsig[0], 3, 125, 2).shape)
    G["tk_nosub"] = ref.takens_embedding(tsig[1], 3, 5, 1)

    # ---- (5) what the reference hands to ripser ----
    calls["ripser"].clear()
    for name in pcs:
        ref.compute_audio_persistence(pcs[name])
    G["ap_kwargs"] = np.array(json.dumps({k: float(v) if not isinstance(v, bool) else v
                                          for k, v in calls["ripser"][0][1].items()}))
    for name, (X, kw) in zip(pcs, calls["ripser"]):
        G["ap_pcnorm_" + name] = X
    flat = np.tile(np.array([[1.0, 2.0, 3.0]]), (5, 1)); flat[:, 1] += np.arange(5)
    calls["ripser"].clear(); ref.compute_audio_persistence(flat)
    G["ap_flat_in"] = flat; G["ap_flat_norm"] = calls["ripser"][0][0]
    small = ref.compute_audio_persistence(flat[:2])
    G["ap_small_h0"] = np.asarray(small[0], float); G["ap_small_h1"] = np.asarray(small[1], float)

    calls["ripser"].clear()
    D = dist[1].copy()
    D_asym = D + np.triu(rng.uniform(-1e-3, 1e-3, D.shape), 1)
    D_asym[3, 3] = 0.25; D_asym[7, 2] = -0.5
    keep = D_asym.copy()
    ref.compute_eeg_persistence(D_asym)
    assert np.array_equal(keep, D_asym)     # caller's matrix untouched (utils.py:137)
    G["ep_in"] = D_asym; G["ep_dm"] = calls["ripser"][0][0]
    G["ep_kwargs"] = np.array(json.dumps({k: (v if isinstance(v, bool) else float(v))
                                          for k, v in calls["ripser"][0][1].items()}))

    # ---- (6) sklearn's distance matrix = what ripser's point-cloud path builds ----
    from sklearn.metrics import pairwise_distances
    for name in pcs:
        G["pd_" + name] = pairwise_distances(G["ap_pcnorm_" + name], metric="euclidean")

    # ---- (7) extract_features ----
    dg = {
        "mixed": np.array([[0.0, 0.5], [0.0, 0.75], [0.1, 1.25], [0.0, np.inf], [0.3, 0.3]]),
        "single": np.array([[0.25, 1.0]]),
        "empty_finite": np.array([[0.0, np.inf], [0.5, np.inf]]),
        "zero_pers": np.array([[0.5, 0.5], [0.7, 0.7]]),
        "f32vals": np.sort(rng.random((37, 2)).astype(np.float32).astype(np.float64), axis=1),
        "big": np.sort(rng.random((150, 2)), axis=1),
    }
    keys = None
    for k, d in dg.items():
        f = ref.extract_features(d)
        keys = list(f.keys())
        G["ef_in_" + k] = d
        G["ef_out_" + k] = np.array([float(f[q]) for q in keys])
    G["ef_keys"] = np.array(json.dumps(keys))

    # ---- (8) safe_wasserstein's clean() ----
    calls["wasserstein"].clear()
    ref.safe_wasserstein(dg["mixed"], dg["empty_finite"])
    ref.safe_wasserstein(np.zeros((0, 2)), dg["single"])
    ref.safe_wasserstein(np.array([1.0, 2.0]), dg["single"])
    for i, (a, b) in enumerate(calls["wasserstein"]):
        G[f"sw_a{i}"] = np.asarray(a, float); G[f"sw_b{i}"] = np.asarray(b, float)

    def boom(a, b):
        raise RuntimeError("x")
    ref.wasserstein_distance = boom
    G["sw_exc"] = np.array([ref.safe_wasserstein(dg["single"], dg["single"])])

    # ---- (9) window selection (cmp:77-80, v2:394-398) ----
    G["sel_linspace"] = np.stack([np.linspace(0, n - 1, 15, dtype=int) for n in range(16, 90)])
    sel = []
    for name, band, n_windows, max_n in [("bb01_ut01", "delta", 71, 39), ("bb17_ut09", "gamma", 45, 39)]:
        seed = int(hashlib.md5(f"{name}-{band}-{42}".encode()).hexdigest()[:8], 16)
        sel.append(np.random.default_rng(seed).choice(n_windows, size=max_n, replace=False))
    G["sel_md5"] = np.stack(sel)

    # ---- constants ----
    G["const"] = np.array(json.dumps({
        "MAX_DIM": ref.MAX_DIM, "MAX_EDGE_LENGTH": ref.MAX_EDGE_LENGTH, "TAKENS_DIM": ref.TAKENS_DIM,
        "TAKENS_SUBSAMPLE": ref.TAKENS_SUBSAMPLE, "FREQ_BANDS": ref.FREQ_BANDS,
        "FS_AUDIO": ref.FS_AUDIO, "FS_EEG": ref.FS_EEG}))
    np.savez_compressed(OUT, **G)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(G), "arrays")


if __name__ == "__main__":
    main()
