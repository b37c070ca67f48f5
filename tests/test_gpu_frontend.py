"""GPU tests of the front-end rows: zero-phase IIR filters against scipy itself (the
library the reference calls: scipy.signal.sosfiltfilt at nb1:258, scipy.signal.filtfilt at
utils.py:74), and the fused raw-EEG -> distance-matrices path against filter + window + oracle."""
import numpy as np
import pytest
from scipy import signal

from oracle import port
from tda_eeg_audio_amd import preprocess

pytestmark = pytest.mark.gpu


def test_sosfiltfilt_bit_identical_to_scipy(ctx):
    rng = np.random.default_rng(0)
    for n_sig, L in [(47, 4606), (47, 2663), (5, 28), (130, 1000), (1, 5741)]:
        x = rng.standard_normal((n_sig, L)).cumsum(axis=1) * 1e-2 + rng.standard_normal((n_sig, L))
        for lo, hi in preprocess.FREQ_BANDS.values():
            sos = preprocess.design_bandpass_filter(lo, hi, 250, 4)
            ref = signal.sosfiltfilt(sos, x, axis=1)
            got = preprocess.sosfiltfilt(sos, x, ctx=ctx)
            assert np.array_equal(got, ref), (n_sig, L, lo, hi, np.abs(got - ref).max())
    with pytest.raises(ValueError):
        preprocess.sosfiltfilt(preprocess.design_bandpass_filter(8, 13, 250), np.zeros((2, 27)), ctx=ctx)
    one = preprocess.apply_bandpass_filter(x[:3], 8, 13, 250, 4)
    assert np.array_equal(one, np.stack([signal.sosfiltfilt(preprocess.design_bandpass_filter(8, 13, 250, 4), r) for r in x[:3]]))


def test_filtfilt_bit_identical_to_scipy(ctx):
    rng = np.random.default_rng(1)
    env = np.abs(rng.standard_normal((9, 5200))).cumsum(axis=1) * 1e-3 + rng.random((9, 5200))
    for lo, hi in preprocess.FREQ_BANDS.values():
        b, a = signal.butter(4, [max(lo / 125, 0.001), min(hi / 125, 0.999)], btype="band")
        ref = signal.filtfilt(b, a, env, axis=1)
        got = preprocess.filtfilt(b, a, env, ctx=ctx)
        assert np.array_equal(got, ref), (lo, hi, np.abs(got - ref).max())
        assert np.array_equal(preprocess.bandpass_filter(env[0], 250, lo, hi), signal.filtfilt(b, a, env[0]))
    b, a = signal.butter(4, 50 / 125 * 0.9, btype="low")             # the envelope low-pass (utils.py:62-63)
    assert np.array_equal(preprocess.filtfilt(b, a, env, ctx=ctx), signal.filtfilt(b, a, env, axis=1))
    s = env[0]
    assert preprocess.bandpass_filter(s, 250, 200, 100) is s        # lo >= hi -> unchanged (utils.py:71-72)


def test_eeg_to_distances_equals_filter_window_loop(ctx):
    """preprocess_file + process_file_graphs semantics on a synthetic raw recording."""
    rng = np.random.default_rng(2)
    L = 4606
    t = np.arange(L) / 250.0
    eeg = rng.standard_normal((47, L)) + 0.5 * np.sin(2 * np.pi * 10 * t)[None, :] * rng.random((47, 1))
    out = preprocess.eeg_to_distances(eeg, 250, ctx=ctx)
    assert list(out) == list(preprocess.FREQ_BANDS)
    for name, (lo, hi) in preprocess.FREQ_BANDS.items():
        filt = np.stack([signal.sosfiltfilt(preprocess.design_bandpass_filter(lo, hi, 250, 4), ch) for ch in eeg])
        windows, times = preprocess.create_sliding_windows(filt, 1.0, 0.75, 250)
        assert windows.shape == (71, 47, 250) and times[0] == 0.5
        _, od = port.corr_dist_batch(windows)
        assert out[name].shape == (71, 47, 47) and np.array_equal(out[name], od), name
