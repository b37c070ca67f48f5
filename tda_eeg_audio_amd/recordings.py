"""
recordings.py -- process_recording (scripts/tda_eeg_audio_comparison.py:45-124) for a whole set of recordings, FROM HOST
MEMORY: raw EEG (n_rec, 47, L) float64 and the 250 Hz audio envelope (n_rec, L) float64 (compute_envelope of the
resampled audio, utils.py:56-63 -- preprocess.compute_envelope) in pinned host buffers go in, the (n_rec, 5, 48) result
rows [W_H0, W_H1, tau, n_windows, 44 aggregated EEG features] come back to the host.

Per shard of recordings (recordings of equal length):
    H2D of the shard (1.73 MB of EEG per recording instead of the 33 MB of its five window stacks)
    zero-phase band-pass of all 47 x n_rec EEG channels, all five bands in ONE launch   (nb1:236-263)
    zero-phase band-pass of the n_rec envelopes, all five bands in one launch, beside it  (utils.py:66-74, cmp:64)
    the selected 1 s windows of the band-passed envelopes                                (create_windows + np.linspace, cmp:65,77-80)
    ONE run_step over the (band, recording) groups of the shard: the EEG windows are read IN PLACE from the band-passed
    recordings (fused window kernel on sliding windows: neither the (n_win, 47, 250) stacks nor the distance matrices
    exist), tau from the first selected window of every group, Takens + Rips, finish, Wasserstein H0 / H1, rows
    D2H of the (n_shard, 5, 48) rows
and the upload of shard k + 1 overlaps the compute of shard k (two sets of buffers, a copy stream); a shard is verified
(class-overflow flags of its groups) after the next one has been queued.
The rows equal pipeline.run_step on the stacked windows of the same band-passed signals bit for bit
(tests/test_gpu_frontend.py::test_recording_pass_equals_stacked_windows).
"""
import numpy as np

from . import pipeline, preprocess
from ._lib import get_ctx

MAX_WINDOWS = 15            # cmp:39


def select_windows(n_win, max_windows=MAX_WINDOWS):
    """cmp:77-80."""
    return np.linspace(0, n_win - 1, max_windows, dtype=int) if n_win > max_windows else np.arange(n_win)


class RecordingPass:
    def __init__(self, n_samples, shard, device, ctx=None, n_ch=47, fs=250, bands=preprocess.FREQ_BANDS,
                 max_windows=MAX_WINDOWS, window_sec=1.0, overlap=0.75, n_sets=None):
        import os
        import torch
        from scipy import signal
        self.ctx = ctx or get_ctx()
        self.dev, self.S, self.L, self.n_ch, self.fs = device, int(shard), int(n_samples), n_ch, fs
        self.bands = list(dict(bands).values())
        self.win = int(window_sec * fs)
        self.step = int(self.win * (1 - overlap))                      # cmp:57-58: 62
        self.per_rec = (self.L - self.win) // self.step + 1 if self.L >= self.win else 0
        assert self.per_rec > 0, "recordings shorter than one window"
        self.pick = select_windows(self.per_rec, max_windows)
        self.k = len(self.pick)
        S, L, k, nb = self.S, self.L, self.k, len(self.bands)
        nyq = fs / 2                                                   # utils.py:66-74
        self.bas = [signal.butter(4, [max(lo / nyq, 0.001), min(hi / nyq, 0.999)], btype="band") for lo, hi in self.bands]
        edge = preprocess._sos_plan(preprocess.design_bandpass_filter(*self.bands[0], fs, preprocess.FILTER_ORDER))[2]
        f64 = dict(dtype=torch.float64, device=device)
        # the (band, recording) groups of a shard are the "recordings" of ONE batch: window (b * S + r) * per_rec + pick
        sel = ((np.arange(nb * S)[:, None]) * self.per_rec + self.pick[None, :]).astype(np.int32).ravel()
        self.sel_t = torch.from_numpy(sel).to(device)
        self.pick_t = torch.from_numpy(self.pick.astype(np.int64)).to(device)
        seg_off = np.arange(0, nb * S * k + 1, k, dtype=np.int32)
        # buffer sets = shards in flight (upload, filters, step, download of consecutive shards overlap)
        self.n_sets = int(n_sets or os.environ.get("TDA_REC_SETS", "2"))
        self.set = [dict(raw=torch.empty((S, n_ch, L), **f64), env=torch.empty((S, L), **f64),
                         y=torch.empty((nb, S * n_ch, L), **f64), ya=torch.empty((nb, S, L), **f64),
                         aw=torch.empty((nb * S * k, self.win), **f64), rows=torch.empty((S, nb, pipeline.RESULT_COLS), **f64),
                         ws=pipeline.Workspace(nb * S * k, seg_off, device, n_ch=n_ch),
                         work=torch.empty((nb, S * n_ch, L + 2 * edge), **f64), worka=torch.empty((nb, S, L + 2 * 3 * 9), **f64),
                         # a stream pair per buffer set: the filters of shard k + 1 (chains of dependent operations on few
                         # waves) run beside the Rips kernels of shard k (which fill the vector units)
                         main=torch.cuda.Stream(device=device), side=torch.cuda.Stream(device=device),
                         up=torch.cuda.Event(), done=torch.cuda.Event(), down=torch.cuda.Event()) for _ in range(self.n_sets)]
        self.copy = torch.cuda.Stream(device=device)                   # uploads
        self.back = torch.cuda.Stream(device=device)                   # rows back (its own stream: the download of shard k waits
                                                                       # for the compute of k, the upload of k + 1 must not)
        self.repairs = 0

    def _rips_step(self, st, retry):
        nb = len(self.bands)
        return pipeline.run_step(None, st["aw"], st["ws"], ctx=self.ctx, max_lag=self.win // 2, retry=retry,
                                 eeg_sliding=(st["y"].view(nb * self.S, self.n_ch, self.L), self.win, self.step, self.sel_t))

    def _shard_step(self, st):
        """Everything between the upload and the rows of one shard, on its main stream (the envelopes' filters on its side stream)."""
        import torch
        ctx, S, k, nb = self.ctx, self.S, self.k, len(self.bands)
        st["side"].wait_stream(st["main"])
        with torch.cuda.stream(st["side"]):
            preprocess.filtfilt_bank_dev(st["env"], self.bas, y_t=st["ya"], work_t=st["worka"], ctx=ctx)
            # create_windows + the selection: a strided view of the band-passed envelopes, gathered into the stack the
            # tau / Takens kernels read (2 KB per window: plumbing)
            st["aw"].view(nb * S, k, self.win).copy_(
                st["ya"].view(nb * S, self.L).unfold(1, self.win, self.step).index_select(1, self.pick_t))
        preprocess.bandpass_bank_dev(st["raw"].view(S * self.n_ch, self.L), self.bands, self.fs, y_t=st["y"], work_t=st["work"], ctx=ctx)
        st["main"].wait_stream(st["side"])
        res = self._rips_step(st, "one")                               # (nb * S, 48), band-major groups
        st["rows"].copy_(res.view(nb, S, pipeline.RESULT_COLS).transpose(0, 1))

    def run(self, raw_h, env_h, rows_h=None):
        """raw_h (n_rec, n_ch, L), env_h (n_rec, L): pinned float64 host tensors.  Returns rows_h (n_rec, n_bands, 48),
        pinned, complete when the call returns."""
        import torch
        n_rec = raw_h.shape[0]
        assert raw_h.shape[1:] == (self.n_ch, self.L) and env_h.shape == (n_rec, self.L)
        nb = len(self.bands)
        if rows_h is None:
            rows_h = torch.empty((n_rec, nb, pipeline.RESULT_COLS), dtype=torch.float64).pin_memory()
        S = self.S
        shards = [(s0, min(S, n_rec - s0)) for s0 in range(0, n_rec, S)]
        pend = []
        for i, (s0, n) in enumerate(shards):
            st = self.set[i % self.n_sets]
            with torch.cuda.stream(self.copy):
                if i >= self.n_sets:                # the shard before in this buffer set has read it and its rows are out
                    self.copy.wait_event(st["done"])
                    self.copy.wait_event(st["down"])
                st["raw"][:n].copy_(raw_h[s0:s0 + n], non_blocking=True)
                st["env"][:n].copy_(env_h[s0:s0 + n], non_blocking=True)
                if n < S:                           # a short last shard: the idle rows repeat its first recording
                    st["raw"][n:].copy_(st["raw"][:1].expand(S - n, -1, -1))
                    st["env"][n:].copy_(st["env"][:1].expand(S - n, -1))
                st["up"].record(self.copy)
            with torch.cuda.stream(st["main"]):
                st["main"].wait_event(st["up"])
                self._shard_step(st)
                st["done"].record(st["main"])
            with torch.cuda.stream(self.back):
                self.back.wait_event(st["done"])
                rows_h[s0:s0 + n].copy_(st["rows"][:n], non_blocking=True)
                st["down"].record(self.back)
            pend.append(i)
            if len(pend) >= self.n_sets:            # (the GPU has the later shards to work on while the host looks at this one;
                self._verify(pend.pop(0), rows_h, shards)      # its buffer set is the next to be reused)
        while pend:
            self._verify(pend.pop(0), rows_h, shards)
        self.back.synchronize()
        return rows_h

    def _verify(self, i, rows_h, shards):
        """Verify, then publish: a shard whose step left a class-overflow flag (run_step copies the flags of its groups
        to pinned memory) is run again with the full ladder -- rare -- and its rows replace the ones already copied."""
        import torch
        st = self.set[i % self.n_sets]
        s0, n = shards[i]
        st["down"].synchronize()
        fl = st["ws"].flags_host
        if bool((fl & 2).any()):
            self.repairs += 1
            nb = len(self.bands)
            with torch.cuda.stream(st["main"]):
                res = self._rips_step(st, "auto")
                st["rows"].copy_(res.view(nb, self.S, pipeline.RESULT_COLS).transpose(0, 1))
                rows_h[s0:s0 + n].copy_(st["rows"][:n])
                fl.copy_(st["ws"].seg_flags, non_blocking=True)
                st["main"].synchronize()
        # (groups of the idle rows of a short last shard repeat real recordings: their flags say nothing new)
        if bool(fl.any()):
            from ._lib import TdaError
            raise TdaError(f"window status bits {int(np.bitwise_or.reduce(fl.numpy())):#x} left in shard {i}: rows withheld")
