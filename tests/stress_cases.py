"""Randomised parity stress of the HIP kernels against the CPU oracle (sorted multisets, bit-exact): audio-like
windows of all bands, EEG-like and white-noise matrices at three thresholds, tie-heavy metrics of odd sizes, point
clouds of 3..124 points with duplicates (fewer edges than one chunk, last chunks that end inside the edge list),
both first-pass class widths, the fused EEG window kernel, and Wasserstein on quantised / near-diagonal /
equal-birth diagrams.  Used by tests/test_gpu_stress.py (a seeded cut, in the driver-run suite) and by
tools/stress_parity.py (any scale / seed; TDA_STRESS_DEBUG=1 loads the LDS guard build).  TEST INFRASTRUCTURE."""
import ctypes as C
import os

import numpy as np

from oracle import brute, port

same = lambda a, b: np.array_equal(brute.sort_rows(a), brute.sort_rows(b))
LDS_GUARD = 0x100          # status bit of the guard build: a sentinel word between two LDS regions was overwritten


class Stress:
    def __init__(self, ctx, engine, synth, seed=2024, scale=1, guard=False, log=print):
        self.ctx, self.engine, self.synth, self.scale, self.guard, self.log = ctx, engine, synth, scale, guard, log
        self.rng = np.random.default_rng(seed)
        self.bad = 0
        self.dbg = (C.c_double * 2048)()

    def cloud_intact(self, n):
        if not self.guard:
            return True
        self.ctx.lib.tda_debug_read(self.dbg)
        v = np.frombuffer(self.dbg, dtype=np.float64)
        return np.array_equal(v[:n], v[1024:1024 + n])

    def fail(self, *msg):
        self.bad += 1
        self.log("FAIL", *msg)

    # ---- audio windows, every band
    def audio(self, n_per_band=150):
        e = self.engine
        for band in ("delta", "theta", "alpha", "beta", "gamma"):
            aw = self.synth.audio_windows(n_per_band * self.scale, band, seed=int(self.rng.integers(1 << 30)))
            tau = int(e.tau_batch(aw[:1], 125, ctx=self.ctx)[0])
            h0, h1, npts, st = e.takens_rips_batch(aw, tau, ctx=self.ctx)
            for w in range(len(aw)):
                (o, _) = port.audio_persistence(aw[w], tau)
                if not (same(h0[w], o[0]) and same(h1[w], o[1]) and (st[w] & ~4) == 0):
                    self.fail("audio", band, w, "status", st[w], len(h1[w]), len(o[1]))

    # ---- EEG-like and white-noise distance matrices, several thresholds; the fused kernel on the same windows
    def matrices(self, n=120):
        import torch
        e = self.engine
        for kind in ("latent", "white"):
            W = self.synth.eeg_windows(n * self.scale, seed=int(self.rng.integers(1 << 30)), kind=kind)
            dist = e.corr_dist_batch(W, want_corr=False, ctx=self.ctx)
            oracle = {}
            for th in (2.0, 1.2, 0.9):
                h0, h1, st = e.rips_dm_batch(dist, thresh=th, ctx=self.ctx)
                for w in range(len(W)):
                    o = oracle[(th, w)] = port.rips_dm(dist[w], thresh=th)
                    if not (same(h0[w], o[0]) and same(h1[w], o[1]) and st[w] == 0):
                        self.fail("dm", kind, th, w, "status", st[w], len(h1[w]), len(o[1]))
            wt = torch.from_numpy(W).to(torch.device("cuda", self.ctx.device))
            for th in (2.0, 0.9):
                out = e.eeg_window_dev(wt, thresh=th, ctx=self.ctx)
                f0, f1 = out.to_lists()
                st = out.status.cpu().numpy()
                for w in range(len(W)):
                    o = oracle[(th, w)]
                    if not (same(f0[w], o[0]) and same(f1[w], o[1]) and st[w] == 0):
                        self.fail("fused", kind, th, w, "status", st[w], len(f1[w]), len(o[1]))

    # ---- random metrics with heavy ties, odd sizes
    def ties(self, reps=6):
        e = self.engine
        for n in (3, 5, 17, 33, 64, 65, 90, 128):
            for rep in range(reps * self.scale):
                d = self.rng.integers(1, 6, size=(n, n)).astype(np.float64) / 4.0
                d = np.minimum(d, d.T); np.fill_diagonal(d, 0.0)
                h0, h1, st = e.rips_dm_batch(d[None], thresh=2.0, h1_cap=4096, ctx=self.ctx)
                o = port.rips_dm(d, thresh=2.0)
                if not (same(h0[0], o[0]) and same(h1[0], o[1]) and st[0] == 0):
                    self.fail("ties n", n, rep, "status", st[0], len(h0[0]), len(o[0]), len(h1[0]), len(o[1]))

    # ---- random clouds incl. duplicates (small clouds many times)
    def clouds(self, small=12, large=6):
        e = self.engine
        for P in list(range(3, 41)) + [47, 64, 65, 80, 100, 124]:
            for rep in range((small if P <= 40 else large) * self.scale):
                dim = 3 if rep % 4 else 2
                pc = self.rng.random((P, dim))
                if rep % 2: pc[P // 2:] = pc[:P - P // 2]          # duplicate points
                th = 2.0 if rep % 3 else 0.6
                h0, h1, st = e.cloud_rips_batch(pc[None], normalise=True, thresh=th, h1_cap=4096, ctx=self.ctx)
                intact = self.cloud_intact(dim * P)
                o = port.rips_f32(port.cloud_dm(port.minmax_normalise(pc)).astype(np.float32), thresh=th)
                ok = same(h0[0], o[0]) and same(h1[0], o[1]) and (st[0] & ~4) == 0 and intact
                if not intact:
                    self.log("LDS CLOBBER cloud P", P, rep)
                if not ok:
                    self.fail("cloud P", P, rep, "status", st[0], len(h0[0]), len(o[0]), len(h1[0]), len(o[1]))
                    os.makedirs("gpurun_out", exist_ok=True)
                    np.savez(f"gpurun_out/fail_cloud_{P}_{rep}.npz", pc=pc, h0=h0[0], h1=h1[0], o0=o[0], o1=o[1])

    def rips_all(self, words=((2, 1), (1, 1))):
        for w in words:
            self.ctx.set_class_words(*w)
            try:
                self.audio(); self.matrices(); self.ties(); self.clouds()
            finally:
                self.ctx.set_class_words(2, 1)
            self.log("class words", w, "mismatches so far:", self.bad)

    # ---- Wasserstein: quantised coordinates (many equal costs), near-diagonal points, empty / single-row diagrams,
    # equal births (1-D path) and general position, sizes up to the buffers the pipeline uses
    def wasserstein(self, rounds=4, bar=1e-7):
        e = self.engine
        for rnd in range(rounds * self.scale):
            As, Bs = [], []
            for k in range(250):
                m, n = int(self.rng.integers(0, 64)), int(self.rng.integers(0, 128))
                q = (0, 4, 16, 1 << 20)[k % 4]                          # 0: continuous

                def dgm(sz, equal_birth):
                    x = self.rng.random((sz, 2))
                    if q: x = np.round(x * q) / q
                    x = np.sort(x, axis=1)
                    if k % 5 == 0: x[:, 1] = x[:, 0] + x[:, 1] * 1e-3       # hugging the diagonal
                    if equal_birth: x[:, 0] = 0.0; x = x[np.argsort(x[:, 1], kind="stable")]
                    return x
                eb = k % 3 == 0
                As.append(dgm(m, eb)); Bs.append(dgm(n, eb))
            ra, ca = e.pack_diagrams(As, cap=64); rb, cb = e.pack_diagrams(Bs, cap=128)
            out, st = e.wasserstein_batch(ra, ca, rb, cb, ctx=self.ctx, want_status=True)
            ref = np.array([brute.safe_wasserstein_oracle(a, b) for a, b in zip(As, Bs)])
            err = np.abs(out - ref)
            # 1e-7: a tenth of the north_star bar.  Coincident points of the two diagrams cost 0 or ~7e-9 depending on
            # how the rounding residue of sklearn's |x|^2 - 2 x.y + |y|^2 falls (FMA or not in the BLAS behind it)
            nb = int((err > bar).sum() + (st != 0).sum())
            if nb:
                i = int(err.argmax())
                os.makedirs("gpurun_out", exist_ok=True)
                np.savez(f"gpurun_out/fail_wasserstein_{rnd}_{i}.npz", a=As[i], b=Bs[i], gpu=out[i], ref=ref[i])
                self.log("FAIL wasserstein round", rnd, "max err", err.max(), "at", i, "status!=0:", int((st != 0).sum()))
            self.bad += nb
