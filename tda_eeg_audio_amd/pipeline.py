"""
pipeline.py -- the end-to-end per-window unit of the reference, batched on one GPU.

Counterpart of the hot loop of process_recording (scripts/tda_eeg_audio_comparison.py:77-122):
for every selected window  corr->dist (nb2:198-207) -> Rips(EEG) (cmp:93) ; tau once per
recording-band from its first selected window (cmp:83) -> Takens -> Rips(audio) (cmp:89-92) ->
Wasserstein H0 and H1 (cmp:95-96) -> features of the H1 diagrams (cmp:98-99); then per
recording-band np.nanmean of the distances (cmp:117-118) and the mean/std feature aggregation of
process_file_features (scripts/tda_eeg_classification_v2.py:429-436).

Everything stays in HBM; each stage is one C-ABI launch on torch's current stream.  torch is
only the allocator / stream owner here.
"""
import numpy as np

from . import engine

RESULT_COLS = 4 + 44      # [w_h0, w_h1, tau, n_windows] + 44 aggregated EEG features per recording-band


class Workspace:
    """Pre-allocated device buffers for a batch of n_win windows grouped into recordings."""

    def __init__(self, n_win, seg_off, device, n_ch=47, h1_cap=engine.DEFAULT_H1_CAP):
        import torch
        self.n_win, self.device = n_win, device
        seg_off = np.asarray(seg_off, np.int32)
        assert seg_off[0] == 0 and seg_off[-1] == n_win
        self.n_seg = len(seg_off) - 1
        self.seg_off = torch.from_numpy(seg_off).to(device)
        rec_id = np.repeat(np.arange(self.n_seg), np.diff(seg_off))
        self.rec_id = torch.from_numpy(rec_id.astype(np.int64)).to(device)
        self.first_idx = torch.from_numpy(seg_off[:-1].astype(np.int64)).to(device)
        f64 = dict(dtype=torch.float64, device=device)
        self.eeg = engine.DeviceDiagrams(n_win, n_ch, h1_cap, device)
        self.aud = engine.DeviceDiagrams(n_win, 128, h1_cap, device)
        self.tau_seg = torch.empty(self.n_seg, dtype=torch.int32, device=device)
        self.tau_win = torch.empty(n_win, dtype=torch.int32, device=device)
        self.seg_flags = torch.zeros(self.n_seg, dtype=torch.int32, device=device)     # class-overflow bits per group
        self.flags_host = torch.zeros(self.n_seg, dtype=torch.int32).pin_memory()
        self.w0 = torch.empty(n_win, **f64); self.w1 = torch.empty(n_win, **f64)
        self.ws0 = torch.empty(n_win, dtype=torch.int32, device=device)
        self.ws1 = torch.empty(n_win, dtype=torch.int32, device=device)
        self.fe0 = torch.empty((n_win, 11), **f64); self.fe1 = torch.empty((n_win, 11), **f64)
        self.fa1 = torch.empty((n_win, 11), **f64)
        self.result = torch.empty((self.n_seg, RESULT_COLS), **f64)
        self.n_win_seg = torch.from_numpy(np.diff(seg_off).astype(np.float64)).to(device)
        self.side_stream = torch.cuda.Stream(device=device)
        import os
        # TDA_OVERLAP=1: EEG chain on a side stream.  Off by default since round 3: a step that forks takes two of the
        # four hardware queues, so only two lanes make progress at a time and their audio kernels end together,
        # leaving the CUs to the (latency-bound, nearly empty) widening passes -- measured with tools/share_ab.sh:
        # one stream per lane is +1.5 % on the full corpus and +9-12 % on the share a rank of eight holds
        self.overlap = os.environ.get("TDA_OVERLAP", "0") != "0"
        # fused EEG window kernel (corr -> dist -> Rips in one launch, the matrix stays in LDS); "0": the two-kernel
        # path through ws.dist (also taken when the channel count is outside the fused kernel's 33..48)
        self.fused = os.environ.get("TDA_FUSED_EEG", "1") != "0" and 33 <= n_ch <= 48
        self.dist = None if self.fused else torch.empty((n_win, n_ch, n_ch), **f64)


def run_step(eeg_win, audio_win, ws, ctx=None, max_lag=125, timers=None, retry="auto", eeg_sliding=None):
    """One pass of the hot path over the batch.  eeg_win (n_win,47,250) f64, audio_win (n_win,250)
    f64, both resident in HBM.  Returns ws.result (n_seg, 48).
    eeg_sliding = (sig_t (n_rec, 47, L), win_len, step, sel_t): the EEG windows are read IN PLACE from band-passed
    recordings instead (window sel_t[i] = r * n_win_per_rec + k; eeg_win is ignored) -- recordings.RecordingPass.
    `timers`: optional dict of
    (start,end) torch.cuda.Event pairs per stage, recorded on the launch stream.
    retry="auto": every Rips call launches its widening passes (exact by itself).  retry="first": first passes
    only; retry="one": first passes plus ONE widening pass each (TDA_RETRY_ONE_STEP: the wide rungs of the ladder
    wait for a nearly empty CU even when they have nothing to redo).  In both cases the class-overflow bits that are
    left are copied to ws.flags_host (pinned) at the end of the step and the caller re-runs the step with
    retry="auto" if any is set (pipeline.Lanes does: verify, then publish)."""
    import torch
    from . import _lib
    ctx = ctx or _lib.get_ctx()
    if retry != "auto":
        ctx.set_retry_policy(ctx.RETRY_FIRST_PASS if retry == "first" else ctx.RETRY_ONE_STEP)
    ctx.set_h1_order(ctx.ORDER_DEFERRED)         # one finishing pass for the three diagram sets of the batch
    try:
        return _run_step(eeg_win, audio_win, ws, ctx, max_lag, timers, retry, eeg_sliding)
    finally:
        ctx.set_h1_order(ctx.ORDER_IN_CALL)
        if retry != "auto":
            ctx.set_retry_policy(ctx.RETRY_AUTO)


def _run_step(eeg_win, audio_win, ws, ctx, max_lag, timers, retry, eeg_sliding=None):
    import torch

    def stage(name, fn):
        if timers is None or name not in timers:
            return fn()
        s, e = timers[name]
        s.record()
        r = fn()
        e.record()
        return r

    # The EEG chain (corr->dist->Rips) and the audio chain (tau->Takens->Rips) are independent until
    # the Wasserstein step; with ws.overlap the EEG kernels run on a side stream (see Workspace: not the default,
    # the other lanes fill the tails better than a fork inside the step does).
    main = torch.cuda.current_stream()
    side = ws.side_stream if ws.overlap else main
    if ws.overlap:
        side.wait_stream(main)
    with torch.cuda.stream(side):
        if eeg_sliding is not None:
            sig_t, win_len, step, sel_t = eeg_sliding
            stage("eeg_window", lambda: engine.eeg_window_sliding_dev(sig_t, win_len, step, sel_t=sel_t, out=ws.eeg, ctx=ctx))
        elif ws.fused and eeg_win.shape[2] <= 256:
            stage("eeg_window", lambda: engine.eeg_window_dev(eeg_win, ws.eeg, ctx=ctx))
        else:
            if ws.dist is None:
                ws.dist = torch.empty((ws.n_win, eeg_win.shape[1], eeg_win.shape[1]), dtype=torch.float64, device=ws.device)
            stage("corr_dist", lambda: engine.corr_dist_dev(eeg_win, ws.dist, None, ctx=ctx))
            stage("rips_eeg", lambda: engine.rips_dm_dev(ws.dist, ws.eeg, ctx=ctx))
    # tau from the first selected window of each recording-band (cmp:83), written per group and per window
    stage("tau", lambda: engine.tau_segments_dev(audio_win, ws.seg_off, max_lag, ws.tau_seg, ws.tau_win, ctx=ctx))
    stage("rips_audio", lambda: engine.takens_rips_dev(audio_win, ws.tau_win, ws.aud, ctx=ctx))
    if ws.overlap:
        main.wait_stream(side)
    # H1 rows of both Rips stages into ripser's order + extract_features of EEG H0 / EEG H1 / audio H1
    # (cmp:98-99, v2:415-416): one launch, one wavefront per diagram
    stage("finish", lambda: engine.diagram_finish_dev([(ws.eeg.h0, ws.eeg.c0, False, ws.fe0),
                                                        (ws.eeg.h1, ws.eeg.c1, True, ws.fe1),
                                                        (ws.aud.h1, ws.aud.c1, True, ws.fa1)], ctx=ctx))
    stage("wasserstein_h0", lambda: engine.wasserstein_dev(ws.eeg.h0, ws.eeg.c0, ws.aud.h0, ws.aud.c0,
                                                           out_t=ws.w0, status_t=ws.ws0, ctx=ctx))
    stage("wasserstein_h1", lambda: engine.wasserstein_dev(ws.eeg.h1, ws.eeg.c1, ws.aud.h1, ws.aud.c1,
                                                           out_t=ws.w1, status_t=ws.ws1, ctx=ctx))
    # per recording-band rows: nanmean of the distances (cmp:117-118), tau, window count, mean/std of the EEG
    # features (v2:429-436) -- one launch
    stage("reduce", lambda: engine.recording_rows_dev(ws.w0, ws.w1, ws.tau_seg, ws.fe0, ws.fe1, ws.seg_off, ws.result,
                                                      ws.eeg.status, ws.aud.status, ws.seg_flags, ctx=ctx))
    if retry != "auto":
        ws.flags_host.copy_(ws.seg_flags, non_blocking=True)
    return ws.result


class Batch:
    """Handle of a submitted batch: `result()` waits for it, repairs it if a window ran out of class bits, runs the
    `post` callback and returns its value (or the lane's result rows, valid until the lane is used again)."""

    def __init__(self, lanes, lane, inputs, post, event):
        self.lanes, self.lane, self.inputs, self.post, self.event = lanes, lane, inputs, post, event
        self.done, self.value, self.repaired = False, None, False

    def result(self):
        if not self.done:
            self.lanes._finalize(self.lane)
        return self.value


class Lanes:
    """`depth` independent (Workspace, stream pair) lanes.  Consecutive batches (recording-bands are
    independent, cmp:77-122 runs them one after the other) go to alternating lanes, so that the
    thin tail of one batch's grids -- 710 workgroups on 256 CUs is 1.4 rounds of the audio Rips
    kernel -- is filled by the head of the next batch instead of leaving CUs idle.  Every lane owns
    its buffers.

    graph=True: the launches of a step are captured once per (lane, input buffers) into a HIP graph
    and replayed afterwards, which takes the per-step host work from ~0.35 ms to ~0.06 ms.  Inputs are
    baked in by address, so feed the lanes from a fixed ring of device buffers.

    defer_retries=True: a step launches only the first pass of the two Rips stages.  Their widening
    passes redo just the windows that ran out of class bits -- rarely any -- but each is a launch
    that needs a large share of a CU before it can even look, and on a full GPU that stalls the lane
    (5-6 % of the throughput).  Instead the overflow bits of the batch travel to pinned host memory
    with the step; when the lane comes up again (`depth` batches later, so the wait is normally
    over) they are inspected, the batch is re-run with the full ladder if any is set, and only then
    is it published through `post` / `Batch.result()` (verify, then publish)."""

    def __init__(self, depth, n_win, seg_off, device, graph=False, defer_retries=True, **kw):
        import torch
        self.depth = max(1, int(depth))
        self.ws = [Workspace(n_win, seg_off, device, **kw) for _ in range(self.depth)]
        self.streams = [torch.cuda.Stream(device=device) for _ in range(self.depth)]
        self.k = 0
        self.graph = bool(graph)
        self.defer = defer_retries if defer_retries == "one" else bool(defer_retries)
        self.graphs = {}
        self.pending = [None] * self.depth
        self.repairs = 0
        self.before_step = None      # optional callable(lane index): runs before a step is launched or captured

    def _finalize(self, i):
        import torch
        b = self.pending[i]
        if b is None:
            return
        self.pending[i] = None
        st, ws = self.streams[i], self.ws[i]
        b.event.synchronize()
        eeg_win, audio_win, ctx, max_lag = b.inputs
        with torch.cuda.stream(st):
            if self.defer and bool(ws.flags_host.any()):
                if bool((ws.flags_host & 2).any()):                        # class overflow left by the short ladder: rare
                    run_step(eeg_win, audio_win, ws, ctx=ctx, max_lag=max_lag, retry="auto")
                    b.repaired = True
                    self.repairs += 1
                    ws.flags_host.copy_(ws.seg_flags, non_blocking=True)
                    st.synchronize()
                # verify before publishing: the full ladder ends in a pass without a capacity limit, so no overflow can be
                # left; a truncated H1 diagram (h1_cap), an oversized cloud or a kernel that gave up are not repairable
                # here -- the rows are not what the reference would give and must not go out
                if bool(ws.flags_host.any()):
                    from ._lib import TdaError
                    bits = int(np.bitwise_or.reduce(ws.flags_host.numpy()))
                    raise TdaError(f"window status bits {bits:#x} left in {int((ws.flags_host != 0).sum())} recording-band "
                                   "group(s) (1: H1 rows truncated, 2: class overflow, 8: not converged, 16: too many points): "
                                   "rows withheld")
            b.value = b.post(ws.result) if b.post is not None else ws.result
        b.done = True

    def submit(self, eeg_win, audio_win, ctx=None, max_lag=125, timers=None, post=None, sync_inputs=True, lane=None):
        """Enqueue one step on the next lane and return its Batch handle.  The batch that used the lane before
        is finalized first (see class docstring).  `post(result)` runs on the lane's stream when the batch is
        finalized (e.g. the all-gather of the result rows).  `timers` forces an eager (uncaptured) step.
        sync_inputs=False skips the wait on the caller's stream (inputs already complete in HBM)."""
        import torch
        i = self.k % self.depth if lane is None else int(lane) % self.depth      # lane=: a fixed batch -> lane map
        self.k += 1                                                              # (one captured graph per batch)
        self._finalize(i)
        st, ws = self.streams[i], self.ws[i]
        if sync_inputs:                                      # inputs produced on the caller's stream; pass False when
            st.wait_stream(torch.cuda.current_stream())     # they are resident and unchanged (costs ~0.07 ms per step)
        retry = ("one" if self.defer == "one" else "first") if self.defer else "auto"
        key = (i, eeg_win.data_ptr(), audio_win.data_ptr(), int(max_lag), id(ctx))
        if self.graph and timers is None:
            g = self.graphs.get(key)
            if g is None:
                if self.before_step is not None:
                    self.before_step(i)
                with torch.cuda.stream(st):                  # eager once: lazy initialisations stay out of the capture
                    run_step(eeg_win, audio_win, ws, ctx=ctx, max_lag=max_lag, retry=retry)
                st.synchronize()
                if self.before_step is not None:
                    self.before_step(i)
                g = torch.cuda.CUDAGraph()
                # thread_local: calls made by other threads meanwhile (e.g. the event queries of the RCCL watchdog
                # at N > 1) must not invalidate the capture
                with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                    run_step(eeg_win, audio_win, ws, ctx=ctx, max_lag=max_lag, retry=retry)
                self.graphs[key] = g
            with torch.cuda.stream(st):
                g.replay()
        else:
            if self.before_step is not None:
                self.before_step(i)
            with torch.cuda.stream(st):
                run_step(eeg_win, audio_win, ws, ctx=ctx, max_lag=max_lag, timers=timers, retry=retry)
        ev = torch.cuda.Event()
        ev.record(st)
        b = Batch(self, i, (eeg_win, audio_win, ctx, max_lag), post, ev)
        self.pending[i] = b
        return b

    def drain(self):
        """Finalize every batch still in flight and make the caller's stream wait for the lanes."""
        import torch
        for i in range(self.depth):
            self._finalize(i)
        cur = torch.cuda.current_stream()
        for st in self.streams:
            cur.wait_stream(st)


class CorpusPass:
    """One rank's share of a corpus-shaped pass: the loops of run_analysis / process_recording
    (scripts/tda_eeg_audio_comparison.py:131-138, 63-122) turned inside out.  The rank's recordings are
    resident in HBM as ONE batch per band -- eeg[b]: (n_rec * wpr, 47, 250), aud[b]: (n_rec * wpr, 250),
    window (r * wpr + w) = the w-th selected window of local recording r -- and a pass is one `run_step` per band,
    fed through `Lanes` (band b always on lane b % depth, so each band's launches are captured once).  The
    (n_rec, n_bands, 48) block of result rows is all-gathered ONCE per pass (dist.all_gather_rows: RCCL over xGMI
    on the GPU box), which replaces the reference's serial loop over recordings and the partial-file merge of
    scripts/tda_eeg_classification_v2.py:608-638."""

    def __init__(self, eeg, aud, wpr, device, ctx, depth=3, graph=True, my_recs=None, shards=None, n_total=None,
                 gather=True, seg_off=None, merge_bands=False, **lane_kw):
        """merge_bands: all bands of the rank's share as ONE batch per pass (band-major groups) instead of one batch
        per band -- five times the windows per launch, which matters when the share is small (a rank of eight holds
        2,655 windows per band: 5.2 rounds of the audio kernel)."""
        import torch
        self.ctx = ctx
        self.n_bands = len(eeg)
        n_win = eeg[0].shape[0]
        assert all(e.shape[0] == n_win for e in eeg) and all(a.shape[0] == n_win for a in aud)
        if seg_off is None:                       # wpr selected windows per recording-band
            assert n_win % wpr == 0
            seg_off = np.arange(0, n_win + 1, wpr, dtype=np.int32)
        seg_off = np.asarray(seg_off, np.int32)
        self.n_rec = len(seg_off) - 1
        if merge_bands and self.n_bands > 1:
            base = getattr(eeg[0], "_base", None)
            step_b = eeg[0].numel() * eeg[0].element_size()
            if base is not None and base.is_contiguous() and all(e.is_contiguous() and e.data_ptr() == eeg[0].data_ptr() + b * step_b
                                                                   for b, e in enumerate(eeg)) \
                    and base.numel() == self.n_bands * eeg[0].numel() and base.data_ptr() == eeg[0].data_ptr():
                eeg_all = base.view(self.n_bands * n_win, *eeg[0].shape[1:])          # the bands already lie back to back
            else:
                eeg_all = torch.cat(list(eeg))
            aud_all = torch.cat(list(aud))
            seg_all = np.concatenate([[0]] + [seg_off[1:] + b * n_win for b in range(self.n_bands)]).astype(np.int32)
            self.batches = [(eeg_all, aud_all, list(range(self.n_bands)))]
            seg_off, n_win = seg_all, self.n_bands * n_win
        else:
            self.batches = [(eeg[b], aud[b], [b]) for b in range(self.n_bands)]
        self.eeg, self.aud = [b[0] for b in self.batches], [b[1] for b in self.batches]
        self.n_win = n_win                        # windows per batch (= per launch of every stage)
        # big batches: ONE widening pass rides along with every Rips call (a few windows in ten thousand need it, and
        # it fits beside the other kernels); the wide rungs of the ladder, which would wait for a nearly empty CU
        # even with nothing to redo, run only for a batch whose flags are still set when it is verified
        lane_kw.setdefault("defer_retries", "one")
        self.lanes = Lanes(depth, n_win, seg_off, device, graph=graph, **lane_kw)
        self.blocks = [torch.empty((self.n_rec, self.n_bands, RESULT_COLS), dtype=torch.float64, device=device)
                       for _ in range(3)]
        self.my_recs, self.shards, self.n_total = my_recs, shards, n_total
        self.gather = gather and shards is not None and len(shards) > 1
        self.passes = 0
        self.gathered = None
        self._open = {}
        self._gather_events = []

    def _post(self, k, i, result):
        """Runs on the lane's stream when batch i of pass k has been verified: its rows join the pass' block; the
        last batch to arrive (any order) all-gathers the block."""
        import torch
        blk = self.blocks[k % len(self.blocks)]
        bands = self.batches[i][2]
        if len(bands) == 1:
            blk[:, bands[0]].copy_(result)
        else:                                     # groups are band-major: (band, recording) -> (recording, band)
            blk.copy_(result.view(len(bands), self.n_rec, RESULT_COLS).transpose(0, 1))
        st = self._open.setdefault(k, {"left": len(self.batches), "events": []})
        ev = torch.cuda.Event()
        ev.record()
        st["events"].append(ev)
        st["left"] -= 1
        if st["left"] == 0:
            cur = torch.cuda.current_stream()
            for e in st["events"]:
                cur.wait_event(e)
            del self._open[k]
            flat = blk.view(self.n_rec, self.n_bands * RESULT_COLS)
            if self.gather:
                from . import dist as tdist
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.gathered = tdist.all_gather_rows(flat, self.my_recs, self.shards, self.n_total)
                e1.record()
                self._gather_events = (self._gather_events + [(e0, e1)])[-16:]
            else:
                self.gathered = flat.clone() if len(self.batches) == 1 else flat
        return None

    def step(self, timers=None):
        """Enqueue one pass (one batch per band, or one in all).  timers: optional {batch index: stage-event dict} --
        those batches are launched eagerly with per-stage events (see run_step)."""
        import functools
        k = self.passes
        self.passes += 1
        for i, (e, a, _) in enumerate(self.batches):
            # one batch per pass: passes alternate over the lanes; otherwise batch i always on lane i mod depth
            lane = (k if len(self.batches) == 1 else i)
            self.lanes.submit(e, a, ctx=self.ctx, post=functools.partial(self._post, k, i), sync_inputs=False,
                              lane=lane, timers=(timers or {}).get(i))

    def prime(self):
        """Untimed set-up: every lane launches its step once eagerly and captures it, so that no later pass pays for
        a capture (with one batch per pass the lanes take turns, and the first `depth` passes would each capture)."""
        for _ in range(self.lanes.depth if len(self.batches) == 1 else 1):
            self.step()
        return self.finish()

    def allgather_ms(self):
        """Mean duration of the all-gather of a pass (events on the lane's stream), None on one rank."""
        import torch
        if not self._gather_events:
            return None
        torch.cuda.synchronize()
        return round(float(np.mean([a.elapsed_time(b) for a, b in self._gather_events])), 4)

    def finish(self):
        """Verify and publish everything in flight; returns the rows of the last pass:
        (n_total or n_rec, n_bands * 48), recordings in the reference's order."""
        self.lanes.drain()
        return self.gathered


def run_features_step(eeg_win, ws, ctx=None):
    """The EEG half alone, as process_file_features needs it (scripts/tda_eeg_classification_v2.py:404-436,
    BASELINE.json configs[2]): corr->dist -> Rips -> 11 features of H0 and of H1 -> mean/std over the windows of
    every (recording, band) group.  Returns ws.feat44 (n_seg, 44)."""
    import torch
    from . import _lib
    ctx = ctx or _lib.get_ctx()
    ctx.set_h1_order(ctx.ORDER_DEFERRED)
    try:
        if ws.fused and eeg_win.shape[2] <= 256:
            engine.eeg_window_dev(eeg_win, ws.eeg, ctx=ctx)
        else:
            if ws.dist is None:
                ws.dist = torch.empty((ws.n_win, eeg_win.shape[1], eeg_win.shape[1]), dtype=torch.float64, device=ws.device)
            engine.corr_dist_dev(eeg_win, ws.dist, None, ctx=ctx)
            engine.rips_dm_dev(ws.dist, ws.eeg, ctx=ctx)
    finally:
        ctx.set_h1_order(ctx.ORDER_IN_CALL)
    engine.diagram_finish_dev([(ws.eeg.h0, ws.eeg.c0, False, ws.fe0), (ws.eeg.h1, ws.eeg.c1, True, ws.fe1)], ctx=ctx)
    if not hasattr(ws, "feat44"):
        ws.feat44 = torch.empty((ws.n_seg, 44), dtype=torch.float64, device=ws.device)
    engine.aggregate_dev(ws.fe0, ws.fe1, ws.seg_off, ws.feat44, ctx=ctx)
    return ws.feat44


STAGES = ["eeg_window", "corr_dist", "rips_eeg", "tau", "rips_audio", "finish", "wasserstein_h0", "wasserstein_h1", "reduce"]
