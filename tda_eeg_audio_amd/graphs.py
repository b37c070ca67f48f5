"""
graphs.py -- drop-in for the functions of the reference's notebooks/2_graph_construction.ipynb
(cited by raw .ipynb JSON line, "nb2"): connectivity graphs = correlation and distance
matrices per window, and their on-disk format graphs/<cond>/<rec>/<band>_{correlations,distances}.npy
((n_win, 47, 47) float64, nb2:194-211), which is the hand-off format of the hot path.
"""
import logging
from pathlib import Path

import numpy as np

from . import engine

FREQ_BANDS = ["delta", "theta", "alpha", "beta", "gamma"]      # nb2:58
N_ELECTRODES = 47                                               # nb2:61


def compute_correlation_matrix(window_data):
    """nb2:86-97 -- np.corrcoef of one (n_ch, n_t) window, NaN -> 0 (corr_dist_kernel)."""
    w = np.asarray(window_data, dtype=np.float64)
    corr, _ = engine.corr_dist_batch(w[None])
    return corr[0]


def correlation_to_distance(corr_matrix, method="euclidean"):
    """nb2:100-122 -- distance from a correlation matrix (corr_to_dist_kernel); ValueError on an
    unknown method exactly as nb2:117."""
    c = np.asarray(corr_matrix, dtype=np.float64)
    return engine.corr_to_dist_batch(c[None], method)[0]


def process_file_graphs(file_dir, output_dir, freq_bands=FREQ_BANDS, distance_method="euclidean"):
    """nb2:158-218 -- all windows and bands of one recording; ONE kernel launch per band instead
    of the reference's per-window Python loop (nb2:198-207)."""
    file_dir, output_dir = Path(file_dir), Path(output_dir)
    file_output_dir = output_dir / file_dir.name
    file_output_dir.mkdir(exist_ok=True, parents=True)
    metadata = {"filename": file_dir.name, "bands": {}}
    for band_name in freq_bands:
        band_file = file_dir / f"{band_name}.npy"
        if not band_file.exists():
            continue
        windows = np.load(band_file)                 # (n_windows, n_electrodes, window_samples)
        n_windows = windows.shape[0]
        if n_windows:
            corr, dist = engine.corr_dist_batch(windows)
            n_nan = int(np.isnan(windows).any(axis=(1, 2)).sum())
            if n_nan:
                logging.warning("%d windows contain NaN samples", n_nan)
            if distance_method != "euclidean":
                dist = engine.corr_to_dist_batch(corr, distance_method)
        else:
            corr = np.zeros((0, windows.shape[1], windows.shape[1]))
            dist = corr.copy()
        np.save(file_output_dir / f"{band_name}_correlations.npy", corr)
        np.save(file_output_dir / f"{band_name}_distances.npy", dist)
        metadata["bands"][band_name] = {"n_windows": n_windows, "n_electrodes": windows.shape[1]}
    return metadata


def batch_process_graphs(input_dir, output_dir, freq_bands=FREQ_BANDS, distance_method="euclidean"):
    """nb2:256-295."""
    input_dir = Path(input_dir)
    file_dirs = sorted([d for d in input_dir.iterdir() if d.is_dir()])
    all_metadata, failed_files = [], []
    for file_dir in file_dirs:
        try:
            all_metadata.append(process_file_graphs(file_dir, output_dir, freq_bands, distance_method))
        except Exception as e:                        # nb2:284-290
            print(f"\nError processing {file_dir.name}: {str(e)}")
            failed_files.append(file_dir.name)
    return all_metadata, failed_files
