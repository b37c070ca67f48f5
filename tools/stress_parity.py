"""Randomised parity stress of the Rips kernels against the CPU oracle (sorted multisets, bit-exact):
audio-like windows of all bands, random point clouds of odd sizes, random metrics with heavy ties, thresholds that cut
the filtration, duplicate points.  Not part of the test-suite (runs a few thousand oracle windows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from oracle import brute, port
from tda_eeg_audio_amd import _lib

# TDA_STRESS_DEBUG=1: use the guard build (make -C tda_eeg_audio_amd/csrc DEBUG_PTS=1), which copies the
# normalised cloud out of LDS before and after the sweep -- a stray LDS write shows even when no diagram row moves
DEBUG = os.environ.get("TDA_STRESS_DEBUG") == "1"
if DEBUG:
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "libtdaeeg_dbg.so")
from tda_eeg_audio_amd import engine, synth

ctx = _lib.get_ctx(0)
dbg = (C.c_double * 2048)()


def cloud_intact(n):
    if not DEBUG:
        return True
    ctx.lib.tda_debug_read(dbg)
    v = np.frombuffer(dbg, dtype=np.float64)
    return np.array_equal(v[:n], v[1024:1024 + n])


same = lambda a, b: np.array_equal(brute.sort_rows(a), brute.sort_rows(b))
bad = 0
SCALE = int(os.environ.get("TDA_STRESS_SCALE", "1"))           # repetitions multiplier
rng = np.random.default_rng(int(os.environ.get("TDA_STRESS_SEED", "2024")))
for words in ((2, 1), (1, 1)):
    ctx.set_class_words(*words)
    # audio windows, every band
    for band in ("delta", "theta", "alpha", "beta", "gamma"):
        aw = synth.audio_windows(150 * SCALE, band, seed=int(rng.integers(1 << 30)))
        tau = int(engine.tau_batch(aw[:1], 125, ctx=ctx)[0])
        h0, h1, npts, st = engine.takens_rips_batch(aw, tau, ctx=ctx)
        for w in range(len(aw)):
            (o, _) = port.audio_persistence(aw[w], tau)
            ok = same(h0[w], o[0]) and same(h1[w], o[1]) and (st[w] & ~4) == 0
            if not ok: print("FAIL audio", band, w, "status", st[w], len(h1[w]), len(o[1]), flush=True)
            bad += not ok
    # EEG-like and white-noise distance matrices, several thresholds
    for kind in ("latent", "white"):
        W = synth.eeg_windows(120 * SCALE, seed=int(rng.integers(1 << 30)), kind=kind)
        dist = engine.corr_dist_batch(W, want_corr=False, ctx=ctx)
        for th in (2.0, 1.2, 0.9):
            h0, h1, st = engine.rips_dm_batch(dist, thresh=th, ctx=ctx)
            for w in range(len(W)):
                o = port.rips_dm(dist[w], thresh=th)
                ok = same(h0[w], o[0]) and same(h1[w], o[1]) and st[w] == 0
                if not ok: print("FAIL dm", kind, th, w, "status", st[w], len(h1[w]), len(o[1]), flush=True)
                bad += not ok
    # random metrics with heavy ties, odd sizes
    for n in (3, 5, 17, 33, 64, 65, 90, 128):
        for rep in range(6 * SCALE):
            d = rng.integers(1, 6, size=(n, n)).astype(np.float64) / 4.0
            d = np.minimum(d, d.T); np.fill_diagonal(d, 0.0)
            h0, h1, st = engine.rips_dm_batch(d[None], thresh=2.0, h1_cap=4096, ctx=ctx)
            o = port.rips_dm(d, thresh=2.0)
            ok = same(h0[0], o[0]) and same(h1[0], o[1]) and st[0] == 0
            if not ok: print("FAIL ties n", n, rep, "status", st[0], len(h0[0]), len(o[0]), len(h1[0]), len(o[1]), flush=True)
            bad += not ok
    # random clouds incl. duplicates
    # (small clouds many times: fewer edges than one chunk, and last chunks that end inside the edge list)
    for P in list(range(3, 41)) + [47, 64, 65, 80, 100, 124]:
        for rep in range((12 if P <= 40 else 6) * SCALE):
            dim = 3 if rep % 4 else 2
            pc = rng.random((P, dim))
            if rep % 2: pc[P // 2:] = pc[:P - P // 2]          # duplicate points
            th = 2.0 if rep % 3 else 0.6
            h0, h1, st = engine.cloud_rips_batch(pc[None], normalise=True, thresh=th, h1_cap=4096, ctx=ctx)
            intact = cloud_intact(dim * P)
            o = port.rips_f32(port.cloud_dm(port.minmax_normalise(pc)).astype(np.float32), thresh=th)
            ok = same(h0[0], o[0]) and same(h1[0], o[1]) and (st[0] & ~4) == 0 and intact
            if not intact: print("LDS CLOBBER cloud P", P, rep, flush=True)
            if not ok:
                print("FAIL cloud P", P, rep, "status", st[0], len(h0[0]), len(o[0]), len(h1[0]), len(o[1]), flush=True)
                os.makedirs("gpurun_out", exist_ok=True)
                np.savez(f"gpurun_out/fail_cloud_{P}_{rep}.npz", pc=pc, h0=h0[0], h1=h1[0], o0=o[0], o1=o[1])
            bad += not ok
    print("class words", words, "mismatches so far:", bad, flush=True)
# Wasserstein: quantised coordinates (many equal costs), near-diagonal points, empty / single-row diagrams, equal
# births (1-D path) and general position, sizes up to the buffers the pipeline uses
wbad = 0
for rnd in range(4 * SCALE):
    As, Bs = [], []
    for k in range(250):
        m, n = int(rng.integers(0, 64)), int(rng.integers(0, 128))
        q = (0, 4, 16, 1 << 20)[k % 4]                          # 0: continuous
        def dgm(sz, equal_birth):
            x = rng.random((sz, 2))
            if q: x = np.round(x * q) / q
            x = np.sort(x, axis=1)
            if k % 5 == 0: x[:, 1] = x[:, 0] + x[:, 1] * 1e-3       # hugging the diagonal
            if equal_birth: x[:, 0] = 0.0; x = x[np.argsort(x[:, 1], kind="stable")]
            return x
        eb = k % 3 == 0
        As.append(dgm(m, eb)); Bs.append(dgm(n, eb))
    ra, ca = engine.pack_diagrams(As, cap=64); rb, cb = engine.pack_diagrams(Bs, cap=128)
    out, st = engine.wasserstein_batch(ra, ca, rb, cb, ctx=ctx, want_status=True)
    ref = np.array([brute.safe_wasserstein_oracle(a, b) for a, b in zip(As, Bs)])
    err = np.abs(out - ref)
    # 1e-7: a tenth of the north_star bar.  Coincident points of the two diagrams cost 0 or ~7e-9 depending on how the
    # rounding residue of sklearn's |x|^2 - 2 x.y + |y|^2 falls (FMA or not in the BLAS behind it): not pinned
    nb = int((err > 1e-7).sum() + (st != 0).sum())
    if nb:
        i = int(err.argmax())
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez(f"gpurun_out/fail_wasserstein_{rnd}_{i}.npz", a=As[i], b=Bs[i], gpu=out[i], ref=ref[i])
    if nb: print("FAIL wasserstein round", rnd, "max err", err.max(), "at", int(err.argmax()), "status!=0:", int((st != 0).sum()), flush=True)
    wbad += nb
print("wasserstein mismatches:", wbad, "(bar 1e-7)", flush=True)
bad += wbad
ctx.set_class_words(2, 1)
print("STRESS", "OK" if bad == 0 else f"FAILED ({bad})")
sys.exit(1 if bad else 0)
