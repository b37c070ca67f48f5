"""
oracle/pipeline_ref.py -- the end-to-end per-window unit on the CPU ORACLE.

TEST INFRASTRUCTURE ONLY: used by __graft_entry__.smoke() as the checker and by bench.py's
cpu_baseline leg as the timed CPU port.  Mirrors tda_eeg_audio_amd/pipeline.py::run_step
(scripts/tda_eeg_audio_comparison.py:77-122 + scripts/tda_eeg_classification_v2.py:429-436).
"""
import numpy as np

RESULT_COLS = 4 + 44


def reference_step_cpu(eeg_win, audio_win, seg_off, max_lag=125):
    """The same unit on the CPU ORACLE -- test/benchmark infrastructure only (imports oracle/)."""
    from oracle import port
    n_win = eeg_win.shape[0]
    seg_off = np.asarray(seg_off)
    res = np.empty((len(seg_off) - 1, RESULT_COLS))
    for s in range(len(seg_off) - 1):
        a, b = seg_off[s], seg_off[s + 1]
        tau = port.compute_tau(audio_win[a], max_lag)
        w0, w1, f0, f1 = [], [], [], []
        for w in range(a, b):
            _, d = port.corr_dist(eeg_win[w])
            e = port.rips_dm(d)
            (au, P) = port.audio_persistence(audio_win[w], tau)
            f0.append(port.features(e[0])); f1.append(port.features(e[1]))
            if P < 3:                      # cmp:90-91: the window takes no part in the distances
                continue
            w0.append(port.wasserstein(_clean(e[0]), _clean(au[0])))
            w1.append(port.wasserstein(_clean(e[1]), _clean(au[1])))
            port.features(au[1])
        f0 = np.array(f0); f1 = np.array(f1)
        # cmp:101-102 drops the band when no window survived; here its two distances are NaN
        res[s, 0] = np.nanmean(w0) if w0 else np.nan
        res[s, 1] = np.nanmean(w1) if w1 else np.nan
        res[s, 2] = tau; res[s, 3] = b - a
        for f in range(11):
            res[s, 4 + 4 * f: 8 + 4 * f] = [f0[:, f].mean(), f0[:, f].std(), f1[:, f].mean(), f1[:, f].std()]
    return res


def _clean(d):
    d = np.asarray(d).reshape(-1, 2)
    m = np.isfinite(d).all(axis=1)
    d = d[m]
    return d if len(d) else np.zeros((1, 2))
