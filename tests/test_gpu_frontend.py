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


def test_audio_front_end_vs_scipy(ctx):
    """resample_poly 44.1 kHz -> 250 Hz, Hilbert envelope + low-pass, per-band windows (cmp:53-65)."""
    rng = np.random.default_rng(3)
    n = 44100 * 6 + 1234
    t_ = np.arange(n) / 44100.0
    audio = (np.sin(2 * np.pi * 220 * t_) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t_)) + 0.1 * rng.standard_normal(n))
    ref_rs = signal.resample_poly(audio, 250, 44100)
    got_rs = preprocess.resample_audio(audio, 44100, 250, ctx=ctx)
    assert got_rs.shape == ref_rs.shape
    assert np.abs(got_rs - ref_rs).max() <= 1e-12 * max(1.0, np.abs(ref_rs).max())
    for m in (len(ref_rs), len(ref_rs) - 1):                    # even and odd lengths
        s = ref_rs[:m]
        ref_env = np.abs(signal.hilbert(s))
        got_env = preprocess.hilbert_envelope(s, ctx=ctx)
        assert np.abs(got_env - ref_env).max() <= 1e-12 * np.abs(ref_env).max()
    b, a = signal.butter(4, min(50, 125 * 0.9) / 125, btype="low")
    ref_env = signal.filtfilt(b, a, np.abs(signal.hilbert(ref_rs)))
    got_env = preprocess.compute_envelope(ref_rs, 250, ctx=ctx)
    assert np.abs(got_env - ref_env).max() <= 1e-11 * np.abs(ref_env).max()
    wins = preprocess.audio_to_band_windows(audio, 44100, ctx=ctx)
    assert list(wins) == list(preprocess.FREQ_BANDS)
    n_win = (len(ref_rs) - 250) // 62 + 1
    for name, (lo, hi) in preprocess.FREQ_BANDS.items():
        bb, aa = signal.butter(4, [max(lo / 125, 0.001), min(hi / 125, 0.999)], btype="band")
        # the 8th-order ba-form band-pass (utils.py:73-74) amplifies a 1e-12 input difference by up to
        # ~1e6 in the delta band (it is ill-conditioned by construction), so the band signals are
        # compared on the SAME envelope: there the recursion is bit-identical to scipy's
        ref_band = signal.filtfilt(bb, aa, got_env)
        assert wins[name].shape == (n_win, 250)
        assert np.array_equal(wins[name][0], ref_band[:250]) and np.array_equal(wins[name][-1], ref_band[(n_win - 1) * 62:(n_win - 1) * 62 + 250])


def test_fused_sliding_windows_equal_stacked_windows(ctx):
    """tda_eeg_window_sliding_dev: windows read in place from (n_rec, 47, L) band-passed recordings == the same kernel
    on the materialised (n_win, 47, 250) stack (nb1:314-381), with and without a window selection; and the whole
    front end, raw EEG -> zero-phase band-pass of all recordings in one launch -> fused sliding kernel, against
    scipy's sosfiltfilt + the oracle."""
    import torch
    from oracle import brute, port
    from tda_eeg_audio_amd import engine, preprocess
    dev = torch.device("cuda", ctx.device)
    rng = np.random.default_rng(21)
    n_rec, L = 5, 1200
    raw = rng.standard_normal((n_rec, 47, L)) + 0.4 * rng.standard_normal((n_rec, 1, L))
    sig_t = torch.from_numpy(raw).to(dev)
    per_rec = (L - 250) // 62 + 1
    stack = np.stack([raw[r][:, k * 62:k * 62 + 250] for r in range(n_rec) for k in range(per_rec)])
    ref = engine.eeg_window_dev(torch.from_numpy(stack).to(dev), ctx=ctx)
    out, got_per_rec = engine.eeg_window_sliding_dev(sig_t, ctx=ctx)
    torch.cuda.synchronize()
    assert got_per_rec == per_rec == 16 and out.n_win == n_rec * per_rec
    assert torch.equal(out.c0, ref.c0) and torch.equal(out.c1, ref.c1) and int(out.status.max()) == 0
    a0, a1 = out.to_lists(); b0, b1 = ref.to_lists()
    assert all(np.array_equal(x, y) for x, y in zip(a0, b0)) and all(np.array_equal(x, y) for x, y in zip(a1, b1))
    sel = np.array([3, 0, 17, 31, 79, 64, 65], np.int32)            # any order, any recording
    dist = torch.empty((len(sel), 47, 47), dtype=torch.float64, device=dev)
    pick, _ = engine.eeg_window_sliding_dev(sig_t, sel_t=torch.from_numpy(sel).to(dev), dist_t=dist, ctx=ctx)
    p0, p1 = pick.to_lists()
    for j, g in enumerate(sel):
        assert np.array_equal(p0[j], b0[g]) and np.array_equal(p1[j], b1[g])
        assert np.array_equal(dist[j].cpu().numpy(), port.corr_dist(stack[g])[1])
    # raw -> band-pass (all recordings, one launch) -> fused sliding kernel, against scipy + oracle
    sos = preprocess.design_bandpass_filter(8, 13, 250)
    filt = preprocess.sosfiltfilt(sos, raw.reshape(n_rec * 47, L), ctx=ctx).reshape(n_rec, 47, L)
    assert np.array_equal(filt, signal.sosfiltfilt(sos, raw, axis=-1))
    out2, _ = engine.eeg_window_sliding_dev(torch.from_numpy(filt).to(dev), ctx=ctx)
    f0, f1 = out2.to_lists()
    for g in (0, 15, 16, 47, 79):
        r, k = divmod(g, per_rec)
        o = port.rips_dm(port.corr_dist(filt[r][:, k * 62:k * 62 + 250])[1])
        assert np.array_equal(brute.sort_rows(f0[g]), brute.sort_rows(o[0])) and np.array_equal(brute.sort_rows(f1[g]), brute.sort_rows(o[1]))


def test_recording_pass_equals_stacked_windows(ctx):
    """process_recording whole, from host memory (recordings.RecordingPass: upload -> band-passes of EEG and envelope ->
    windows read in place -> tau / Takens / Rips / Wasserstein / rows -> host), shards of 3 with a short last shard,
    against (a) pipeline.run_step on the STACKED selected windows of the same band-passed signals (bit for bit) and
    (b) scipy's filters + the CPU oracle end to end for one recording-band."""
    import torch
    from oracle import pipeline_ref
    from tda_eeg_audio_amd import pipeline, preprocess, recordings, utils
    dev = torch.device("cuda", ctx.device)
    rng = np.random.default_rng(33)
    n_rec, L = 7, 1500
    raw = rng.standard_normal((n_rec, 47, L)) + 0.5 * rng.standard_normal((n_rec, 1, L))
    env = np.abs(rng.standard_normal((n_rec, L))).cumsum(axis=1) * 0.01 + np.abs(rng.standard_normal((n_rec, L)))
    rp = recordings.RecordingPass(L, 3, dev, ctx=ctx)
    rows = rp.run(torch.from_numpy(raw).pin_memory(), torch.from_numpy(env).pin_memory()).numpy()
    assert rows.shape == (n_rec, 5, 48) and np.isfinite(rows).all()
    per_rec = (L - 250) // 62 + 1
    pick = recordings.select_windows(per_rec)
    assert np.array_equal(pick, np.linspace(0, per_rec - 1, 15, dtype=int))
    seg_off = np.arange(0, n_rec * 15 + 1, 15, dtype=np.int32)
    ws = pipeline.Workspace(n_rec * 15, seg_off, dev)
    for b, (name, (lo, hi)) in enumerate(preprocess.FREQ_BANDS.items()):
        y = preprocess.apply_bandpass_filter(raw.reshape(n_rec * 47, L), lo, hi, 250).reshape(n_rec, 47, L)
        ya = np.stack([preprocess.bandpass_filter(env[r], 250, lo, hi) for r in range(n_rec)])
        eeg = np.stack([y[r][:, k * 62:k * 62 + 250] for r in range(n_rec) for k in pick])
        aud = np.stack([utils.create_windows(ya[r], 250, 62)[k] for r in range(n_rec) for k in pick])
        ref = pipeline.run_step(torch.from_numpy(eeg).to(dev), torch.from_numpy(aud).to(dev), ws, ctx=ctx).cpu().numpy()
        assert np.array_equal(rows[:, b], ref, equal_nan=True), name
        if name == "alpha":                     # scipy + oracle, one recording
            ys = signal.sosfiltfilt(preprocess.design_bandpass_filter(lo, hi, 250), raw[2], axis=-1)
            bb, aa = signal.butter(4, [lo / 125, hi / 125], btype="band")
            yas = signal.filtfilt(bb, aa, env[2])
            e1 = np.stack([ys[:, k * 62:k * 62 + 250] for k in pick])
            a1 = np.stack([yas[k * 62:k * 62 + 250] for k in pick])
            o = pipeline_ref.reference_step_cpu(e1, a1, np.array([0, 15], np.int32))
            assert np.abs(rows[2, b, :2] - o[0, :2]).max() < 1e-6 and np.array_equal(rows[2, b, 2:4], o[0, 2:4])
            assert np.allclose(rows[2, b, 4:], o[0, 4:], rtol=1e-9, atol=1e-12)
