"""
drivers.py -- batched counterparts of the reference's per-recording drivers, on the
reference's on-disk formats.

  select_windows_even / select_windows_md5   window selection, reproduced exactly (host side):
        np.linspace(0, n-1, 15, dtype=int)          scripts/tda_eeg_audio_comparison.py:77-80,
                                                    scripts/matched_vs_mismatched.py:52-55,77-80
        default_rng(md5(f"{dir}-{band}-{42}")[:8]).choice(n, size, replace=False)
                                                    scripts/tda_eeg_classification_v2.py:394-398
  process_file_features   scripts/tda_eeg_classification_v2.py:338-442  (220 features / recording)
  get_eeg_diagrams        scripts/matched_vs_mismatched.py:66-85
  get_audio_diagrams_from_windows / compute_cross_wasserstein   mvm:43-63, 87-95
  process_recording_arrays  scripts/tda_eeg_audio_comparison.py:63-122 given the band-passed audio windows
  process_recording / process_recording_file / run_analysis / detailed_rows
                            cmp:45-157 from the raw audio track (.mat) and the graphs/ tree to the rows of
                            results/eeg_audio_tda_detailed.csv
  validate_distance_matrix  scripts/tda_eeg_classification_v2.py:110-140 (first window of every band, v2:380-382)
  save_dataset              v2:670-688 incl. features/metadata.csv / metadata.json
"""
import hashlib
from pathlib import Path

import numpy as np

from . import engine
from .utils import FEATURE_KEYS, FREQ_BANDS, MAX_EDGE_LENGTH, TAKENS_DIM, TAKENS_SUBSAMPLE

MAX_WINDOWS = 15            # cmp:39, mvm:32
BANDS = list(FREQ_BANDS)    # v2:63 order: delta, theta, alpha, beta, gamma


def select_windows_even(n_win, max_windows=MAX_WINDOWS):
    if n_win > max_windows:
        return np.linspace(0, n_win - 1, max_windows, dtype=int)
    return np.arange(n_win)


def select_windows_md5(dirname, band, n_windows, max_n, random_state=42):
    max_n = min(int(max_n), n_windows)
    seed = int(hashlib.md5(f"{dirname}-{band}-{random_state}".encode()).hexdigest()[:8], 16)
    return np.random.default_rng(seed).choice(n_windows, size=max_n, replace=False)


def check_status(st, what, allow_degenerate=False):
    """Raises on any window status the drivers cannot publish: class overflow beyond the widest pass (2), H1 rows
    cut at h1_cap (1), a cloud with more than TDA_MAX_POINTS points (16).  The reference has no such limits (ripser
    grows its columns on the heap), so silently passing a cut diagram on would be a different answer."""
    st = np.asarray(st)
    bad = st & (1 | 2 | 16)
    if not allow_degenerate:
        bad = bad | (st & 4)
    if bad.any():
        w = int(np.nonzero(bad)[0][0])
        raise RuntimeError(f"{what}: window {w} reported status {int(st[w])} "
                           "(1: H1 rows cut at h1_cap, 2: class capacity exceeded, 16: cloud too large)")


def validate_distance_matrix(distance_matrix, name=""):
    """scripts/tda_eeg_classification_v2.py:110-140 -- (is_valid, issues) with the reference's checks, order and
    messages: 2-D, square, symmetric (rtol 1e-5, atol 1e-8), no value below -1e-10, zero diagonal (atol 1e-10), no
    NaN, no Inf.  The reference runs it on the FIRST window of every band only (v2:380) and only logs the outcome
    into the recording's metadata (v2:381-382): a host-side check of one 47 x 47 matrix per recording-band.
    This function MIRRORS v2:110-140 by contract -- the same seven checks in the same order with the same tolerances and
    the same (Spanish) message strings, because the messages are an output format (they land in metadata.csv /
    metadata.json, v2:684-688) and are pinned by a fixture generated from the reference (tests/golden)."""
    distance_matrix = np.asarray(distance_matrix)
    issues = []
    if distance_matrix.ndim != 2:
        issues.append(f"No es 2D: forma={distance_matrix.shape}")
        return False, issues
    n, m = distance_matrix.shape
    if n != m:
        issues.append(f"No es cuadrada: forma=({n}, {m})")
        return False, issues
    if not np.allclose(distance_matrix, distance_matrix.T, rtol=1e-5, atol=1e-8):
        max_diff = np.max(np.abs(distance_matrix - distance_matrix.T))
        issues.append(f"No simétrica: asimetría máxima={max_diff:.6f}")
    if np.any(distance_matrix < -1e-10):
        issues.append(f"Valores negativos presentes: min={np.min(distance_matrix):.6f}")
    diag = np.diag(distance_matrix)
    if not np.allclose(diag, 0, atol=1e-10):
        issues.append(f"Diagonal no cero: max={np.max(np.abs(diag)):.6f}")
    if np.any(np.isnan(distance_matrix)):
        issues.append("Contiene valores NaN")
    if np.any(np.isinf(distance_matrix)):
        issues.append("Contiene valores Inf")
    return len(issues) == 0, issues


def feature_names(bands=BANDS):
    """Column order of features/feature_names.txt (v2:429-436)."""
    names = []
    for band in bands:
        for f in FEATURE_KEYS:
            names += [f"{band}_h0_{f}_mean", f"{band}_h0_{f}_std", f"{band}_h1_{f}_mean", f"{band}_h1_{f}_std"]
    return names


def features_from_distances(dist_list, thresh=MAX_EDGE_LENGTH):
    """dist_list: list of (k_i, n, n) arrays, one per (recording, band) group, windows already
    selected.  One Rips launch + two feature launches + one aggregation launch for all groups.
    Returns (len(dist_list), 44) float64."""
    sizes = np.array([len(d) for d in dist_list])
    allw = np.concatenate([np.asarray(d, dtype=np.float64) for d in dist_list if len(d)], axis=0)
    h0, c0, h1, c1, st = engine.rips_dm_batch(allw, thresh=thresh, raw=True)
    if (st & 1).any():                         # more H1 rows than the default capacity: once more, large enough
        h0, c0, h1, c1, st = engine.rips_dm_batch(allw, thresh=thresh, raw=True, h1_cap=int(c1.max()))
    check_status(st, "features_from_distances")
    f0 = engine.features_batch(h0, c0)
    f1 = engine.features_batch(h1, c1)
    seg = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    return engine.aggregate_batch(f0, f1, seg)


def process_file_features(file_dir, freq_bands=BANDS, max_dim=1, max_edge_length=MAX_EDGE_LENGTH,
                          max_windows_per_band=None, window_sampling="random", random_state=42):
    """v2:338-442 -- returns (features dict in reference key order, metadata)."""
    file_dir = Path(file_dir)
    metadata = {"n_windows": {}, "n_windows_used": {}, "validation_issues": [],
                "window_sampling": window_sampling, "max_windows_per_band": max_windows_per_band}
    groups, bands_present = [], []
    for band in freq_bands:
        dist_file = file_dir / f"{band}_distances.npy"
        if not dist_file.exists():
            metadata["n_windows"][band] = 0
            continue
        try:
            dms = np.load(dist_file)
        except Exception as e:
            metadata["validation_issues"].append(f"{band}: error de carga - {e}")
            continue
        n_windows = dms.shape[0]
        metadata["n_windows"][band] = n_windows
        if n_windows == 0:
            continue
        is_valid, issues = validate_distance_matrix(dms[0], f"{band}[0]")             # v2:380-382
        if not is_valid:
            metadata["validation_issues"].extend([f"{band}: {i}" for i in issues])
        if max_windows_per_band is None:
            use = np.arange(n_windows)
        else:
            max_n = max_windows_per_band.get(band, n_windows) if isinstance(max_windows_per_band, dict) \
                else int(max_windows_per_band)
            max_n = min(max_n, n_windows)
            use = select_windows_md5(file_dir.name, band, n_windows, max_n, random_state) \
                if window_sampling == "random" else np.arange(max_n)
        metadata["n_windows_used"][band] = len(use)
        groups.append(dms[use]); bands_present.append(band)
    file_features = {}
    if groups:
        agg = features_from_distances(groups, max_edge_length)
        for g, band in enumerate(bands_present):
            k = 0
            for f in FEATURE_KEYS:
                for tag in ("h0", "h1"):
                    for stat in ("mean", "std"):
                        file_features[f"{band}_{tag}_{f}_{stat}"] = agg[g, k]
                        k += 1
    metadata["n_windows_total"] = int(sum(metadata["n_windows"].values()))
    metadata["n_windows_used_total"] = int(sum(metadata["n_windows_used"].values()))
    return file_features, metadata


def get_eeg_diagrams(graph_dir, bands=BANDS):
    """mvm:66-85 -- {band: [[H0, H1] per selected window]} from <band>_distances.npy."""
    graph_dir = Path(graph_dir)
    if not graph_dir.exists():
        return None
    result = {}
    for bname in bands:
        dist_file = graph_dir / f"{bname}_distances.npy"
        if not dist_file.exists():
            continue
        dms = np.load(str(dist_file))
        if dms.shape[0] == 0:
            continue
        idx = select_windows_even(dms.shape[0])
        h0, h1, st = engine.rips_dm_batch(dms[idx])
        check_status(st, f"get_eeg_diagrams({bname})")
        result[bname] = [[a, b] for a, b in zip(h0, h1)]
    return result


def get_audio_diagrams_from_windows(audio_wins):
    """mvm:50-62 for one band: selected windows, tau from the first one, Takens + Rips; windows
    whose cloud has < 3 points are skipped (mvm:60)."""
    n_win = len(audio_wins)
    if n_win == 0:
        return []
    idx = select_windows_even(n_win)
    sel = np.asarray(audio_wins, dtype=np.float64)[idx]
    tau = int(engine.tau_batch(sel[:1], max_lag=sel.shape[1] // 2)[0])
    h0, h1, npts, st = engine.takens_rips_batch(sel, tau, TAKENS_DIM, TAKENS_SUBSAMPLE)
    check_status(st, "get_audio_diagrams", allow_degenerate=True)
    return [[a, b] for a, b, p in zip(h0, h1, npts) if p >= 3]


def compute_cross_wasserstein(eeg_dgms_band, audio_dgms_band):
    """mvm:87-95 -- nanmean of W(H1, H1) over windows paired by index."""
    n = min(len(eeg_dgms_band), len(audio_dgms_band))
    if n == 0:
        return np.nan
    ra, ca = engine.pack_diagrams([d[1] for d in eeg_dgms_band[:n]])
    rb, cb = engine.pack_diagrams([d[1] for d in audio_dgms_band[:n]])
    w, st = engine.wasserstein_batch(ra, ca, rb, cb, want_status=True)
    w = np.where(st == 0, w, np.nan)
    return float(engine.segment_nanmean(w, np.array([0, n], np.int32))[0])


def process_recording_arrays(audio_band_windows, eeg_dists_by_band):
    """cmp:63-122 for one recording.  audio_band_windows: {band: (n_a, 250) band-passed audio
    windows}; eeg_dists_by_band: {band: (n_e, 47, 47)}.  Returns {band: {...}} with the keys of
    the reference's per-band result (wasserstein_h0/h1, n_windows, tau) plus the per-window H1
    feature time series used by its Spearman step (cmp:104-115)."""
    out = {}
    for bname in BANDS:
        if bname not in audio_band_windows or bname not in eeg_dists_by_band:
            continue
        aw = np.asarray(audio_band_windows[bname], dtype=np.float64)
        ed = np.asarray(eeg_dists_by_band[bname], dtype=np.float64)
        n_win = min(len(aw), ed.shape[0])
        if n_win == 0:
            continue
        idx = select_windows_even(n_win)
        tau = int(engine.tau_batch(aw[idx[:1]], max_lag=aw.shape[1] // 2)[0])          # cmp:83
        a0, ac0, a1, ac1, npts, ast = engine.takens_rips_batch(aw[idx], tau, TAKENS_DIM, TAKENS_SUBSAMPLE, raw=True)
        check_status(ast, f"process_recording({bname}) audio", allow_degenerate=True)
        keep = npts >= 3                                                               # cmp:90-91
        if not keep.any():
            continue
        e0, ec0, e1, ec1, est = engine.rips_dm_batch(ed[idx], raw=True)
        check_status(est, f"process_recording({bname}) eeg")
        k = np.nonzero(keep)[0].astype(np.int32)
        w0, s0 = engine.wasserstein_batch(e0, ec0, a0, ac0, k, k, want_status=True)
        w1, s1 = engine.wasserstein_batch(e1, ec1, a1, ac1, k, k, want_status=True)
        w0 = np.where(s0 == 0, w0, np.nan); w1 = np.where(s1 == 0, w1, np.nan)
        seg = np.array([0, len(k)], np.int32)
        fa = engine.features_batch(a1[k], ac1[k])
        fe = engine.features_batch(e1[k], ec1[k])
        out[bname] = {
            "wasserstein_h0": float(engine.segment_nanmean(w0, seg)[0]),
            "wasserstein_h1": float(engine.segment_nanmean(w1, seg)[0]),
            "n_windows": int(len(idx)), "tau": tau,
            "audio_h1_features": fa, "eeg_h1_features": fe,
        }
        r, p = engine.spearman_batch(fa, fe, seg)                       # cmp:104-114
        out[bname]["feature_correlations"] = {f: {"r": float(r[0, k]), "p": float(p[0, k])}
                                              for k, f in enumerate(engine.SPEARMAN_FEATURES)}
    return out


# --------------------------------------------------------------------------------------
# whole-corpus feature matrix (scripts/tda_eeg_classification_v2.py:445-606, 670-688)
# --------------------------------------------------------------------------------------
def compute_min_windows_per_band(graphs_dirs, freq_bands=BANDS):
    """v2:445-474 -- global per-band minimum window count (np.load with mmap: headers only)."""
    min_windows = {band: np.inf for band in freq_bands}
    for graphs_dir in graphs_dirs:
        graphs_dir = Path(graphs_dir)
        if not graphs_dir.exists():
            continue
        for file_dir in [d for d in graphs_dir.iterdir() if d.is_dir()]:
            for band in freq_bands:
                dist_file = file_dir / f"{band}_distances.npy"
                if not dist_file.exists():
                    continue
                try:
                    n_windows = np.load(dist_file, mmap_mode="r").shape[0]
                    if n_windows > 0:
                        min_windows[band] = min(min_windows[band], n_windows)
                except Exception:
                    continue
    return {b: (0 if v == np.inf else int(v)) for b, v in min_windows.items()}


def _entries(graphs_dir_slow, graphs_dir_fast, batch_start=0, batch_end=None):
    """v2:532-541 -- sorted slow recordings (label 0) then sorted fast ones (label 1), sliced."""
    slow = sorted([d for d in Path(graphs_dir_slow).iterdir() if d.is_dir()])
    fast = sorted([d for d in Path(graphs_dir_fast).iterdir() if d.is_dir()])
    entries = [(d, 0) for d in slow] + [(d, 1) for d in fast]
    total = len(entries)
    if batch_end is None or batch_end < 0:
        batch_end = total
    return entries[max(0, batch_start):min(batch_end, total)], total


def create_dataset(graphs_dir_slow, graphs_dir_fast, freq_bands=BANDS, max_dim=1, max_edge_length=MAX_EDGE_LENGTH,
                   equalize_windows=True, window_sampling="random", max_windows_per_band="min", random_state=42,
                   batch_start=0, batch_end=None, rank=0, world_size=1, gather=None):
    """v2:499-606.  Returns (X, y, subjects, feature_names, filenames, metadata) with the reference's
    row order.  All recordings of this call go through ONE Rips launch (+ features + aggregation).

    world_size > 1: recordings are dealt to ranks by dist.shard_recordings (descending window
    count), every rank computes its rows and `gather(local_rows, my_recs, shards, n_total)`
    (dist.all_gather_rows) assembles the full X on every rank -- this replaces the reference's
    BATCH_START/BATCH_END partial files and their merge (v2:55-60, 608-638)."""
    if equalize_windows and max_windows_per_band == "min":
        max_windows_per_band = compute_min_windows_per_band([graphs_dir_slow, graphs_dir_fast], freq_bands)
    elif not equalize_windows:
        max_windows_per_band = None
    entries, _ = _entries(graphs_dir_slow, graphs_dir_fast, batch_start, batch_end)
    n_rec = len(entries)
    names = feature_names(freq_bands)
    # window counts decide the sharding (mmap: headers only)
    n_win_rec = np.zeros(n_rec, dtype=np.int64)
    for i, (d, _) in enumerate(entries):
        for band in freq_bands:
            f = d / f"{band}_distances.npy"
            if f.exists():
                n_win_rec[i] += np.load(f, mmap_mode="r").shape[0]
    if world_size > 1:
        from . import dist as tdist
        shards = tdist.shard_recordings(n_win_rec, world_size)
        mine = shards[rank]
    else:
        shards, mine = None, np.arange(n_rec)
    groups, where, metadata = [], [], [None] * n_rec
    for i in mine:
        file_dir, label = entries[i]
        md = {"n_windows": {}, "n_windows_used": {}, "validation_issues": [], "window_sampling": window_sampling,
              "max_windows_per_band": max_windows_per_band}
        for bi, band in enumerate(freq_bands):
            dist_file = file_dir / f"{band}_distances.npy"
            if not dist_file.exists():
                md["n_windows"][band] = 0
                continue
            dms = np.load(dist_file)
            n_windows = dms.shape[0]
            md["n_windows"][band] = n_windows
            if n_windows == 0:
                continue
            is_valid, issues = validate_distance_matrix(dms[0], f"{band}[0]")           # v2:380-382
            if not is_valid:
                md["validation_issues"].extend([f"{band}: {i}" for i in issues])
            if max_windows_per_band is None:
                use = np.arange(n_windows)
            else:
                max_n = max_windows_per_band.get(band, n_windows) if isinstance(max_windows_per_band, dict) \
                    else int(max_windows_per_band)
                max_n = min(max_n, n_windows)
                use = select_windows_md5(file_dir.name, band, n_windows, max_n, random_state) \
                    if window_sampling == "random" else np.arange(max_n)
            md["n_windows_used"][band] = len(use)
            groups.append(dms[use]); where.append((i, bi))
        md["n_windows_total"] = int(sum(md["n_windows"].values()))
        md["n_windows_used_total"] = int(sum(md["n_windows_used"].values()))
        md["filename"] = file_dir.name; md["subject"] = file_dir.name.split("_")[0]; md["label"] = label
        metadata[i] = md
    X = np.full((n_rec, 44 * len(freq_bands)), np.nan)
    if groups:
        agg = features_from_distances(groups, max_edge_length)
        for g, (i, bi) in enumerate(where):
            X[i, 44 * bi:44 * (bi + 1)] = agg[g]
    if world_size > 1:
        import torch
        import torch.distributed as tdist_
        from . import dist as tdist
        gather = gather or tdist.all_gather_rows
        # the rows travel from this rank's GPU (RCCL moves device memory; gloo, in the rehearsals, is staged
        # through the host by all_gather_rows itself) and come back to the host once assembled
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        X = gather(torch.from_numpy(X[mine]).to(dev), mine, shards, n_rec).cpu().numpy()
        if tdist_.is_initialized():              # the per-recording metadata of the other ranks (v2:684-688 saves all)
            parts = [None] * world_size
            tdist_.all_gather_object(parts, [(int(i), metadata[i]) for i in mine])
            for part in parts:
                for i, md in part:
                    metadata[i] = md
    y = np.array([lab for _, lab in entries])
    filenames = [d.name for d, _ in entries]
    subjects = np.array([f.split("_")[0] for f in filenames])
    return X, y, subjects, names, filenames, [m for m in metadata if m is not None]


def save_dataset(features_dir, X, y, subjects, names, filenames, metadata=None):
    """v2:670-688 -- features/X.npy, y.npy, subjects.npy, feature_names.txt, filenames.txt and, when the
    per-recording metadata are given, metadata.csv (pandas.DataFrame(all_metadata).to_csv(index=False)) and
    metadata.json (indent 2, ensure_ascii False)."""
    import json
    features_dir = Path(features_dir)
    features_dir.mkdir(exist_ok=True, parents=True)
    np.save(features_dir / "X.npy", X)
    np.save(features_dir / "y.npy", y)
    np.save(features_dir / "subjects.npy", subjects)
    (features_dir / "feature_names.txt").write_text("".join(f"{n}\n" for n in names))
    (features_dir / "filenames.txt").write_text("".join(f"{n}\n" for n in filenames))
    if metadata:
        import pandas as pd
        pd.DataFrame(metadata).to_csv(features_dir / "metadata.csv", index=False)
        with open(features_dir / "metadata.json", "w") as f:
            json.dump(metadata, f, indent=2, ensure_ascii=False)


DETAILED_COLUMNS = ["filename", "condition", "subject", "band", "wasserstein_h0", "wasserstein_h1", "n_windows", "tau"] + \
    [f"corr_{feat}_{k}" for feat in engine.SPEARMAN_FEATURES for k in ("r", "p")]


def process_recording_file(filename, condition, data_dir, graphs_dir, fs_audio=44100):
    """cmp:45-124 on the reference's directory layout: data/<condition>/<filename>.mat (audio track `y`, averaged
    over its channels as load_audio does, utils.py:47-53) and graphs/<condition>/<stem>/<band>_distances.npy.
    Returns the reference's result dict {"filename", "condition", "subject", "bands": {...}} or None when a path is
    missing or no band survives (cmp:48-49, cmp:124)."""
    mat_path = Path(data_dir) / condition / filename
    graph_dir = Path(graphs_dir) / condition / filename.replace(".mat", "")
    if not mat_path.exists() or not graph_dir.exists():
        return None
    import scipy.io as sio
    y = sio.loadmat(str(mat_path))["y"]
    if y.ndim == 2:
        y = y.mean(axis=1)
    dists = {}
    for bname in BANDS:
        f = graph_dir / f"{bname}_distances.npy"
        if f.exists():                                                   # cmp:67-70
            dists[bname] = np.load(str(f))
    bands = process_recording(y.astype(np.float64), dists, fs_audio)
    for bd in bands.values():                                            # the reference keeps scalars only (cmp:116-122)
        bd.pop("audio_h1_features", None); bd.pop("eeg_h1_features", None)
    if not bands:
        return None
    return {"filename": filename, "condition": condition, "subject": filename.split("_")[0], "bands": bands}


def detailed_rows(all_results):
    """cmp:145-157 -- one row per (recording, band): the columns of results/eeg_audio_tda_detailed.csv."""
    rows = []
    for r in all_results:
        for bname, bd in r["bands"].items():
            row = {"filename": r["filename"], "condition": r["condition"], "subject": r["subject"], "band": bname,
                   "wasserstein_h0": bd["wasserstein_h0"], "wasserstein_h1": bd["wasserstein_h1"],
                   "n_windows": bd["n_windows"], "tau": bd["tau"]}
            for feat, vals in bd["feature_correlations"].items():
                row[f"corr_{feat}_r"] = vals["r"]
                row[f"corr_{feat}_p"] = vals["p"]
            rows.append(row)
    return rows


def run_analysis(data_dir, graphs_dir, out_csv=None, conditions=("slow", "fast")):
    """cmp:126-157, the per-recording half: every data/<condition>/*.mat in sorted order through
    process_recording_file, then the detailed table (written as results/eeg_audio_tda_detailed.csv when out_csv is
    given).  The statistics and plots behind it (cmp:159-349) are outside this engine."""
    import pandas as pd
    all_results = []
    for condition in conditions:
        d = Path(data_dir) / condition
        for fn in sorted(f.name for f in d.glob("*.mat")) if d.exists() else []:
            r = process_recording_file(fn, condition, data_dir, graphs_dir)
            if r:
                all_results.append(r)
    df = pd.DataFrame(detailed_rows(all_results), columns=DETAILED_COLUMNS)
    if out_csv:
        Path(out_csv).parent.mkdir(parents=True, exist_ok=True)
        df.to_csv(out_csv, index=False)
    return all_results, df


def process_recording(audio, eeg_dists_by_band, fs_audio=44100):
    """cmp:45-124 from the raw audio track (44.1 kHz, as load_audio returns it, utils.py:47-53) and the
    recording's EEG distance matrices: resample -> envelope -> per-band filter -> windows
    (preprocess.audio_to_band_windows, all filters on the GPU) -> process_recording_arrays."""
    from . import preprocess
    return process_recording_arrays(preprocess.audio_to_band_windows(audio, fs_audio), eeg_dists_by_band)
