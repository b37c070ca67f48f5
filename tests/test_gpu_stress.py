"""A seeded cut of the randomised parity stress (tests/stress_cases.py) in the driver-run GPU suite, and one run of the
LDS guard build (sentinel words between the LDS regions of the Rips kernels; libtdaeeg_dbg.so, built by
__graft_entry__.build()) in a child process.  ripser / persim cannot be pinned by reference fixtures (SURVEY.md
section 8c), so breadth of independent cross-checks against the oracle is the lever on parity here."""
import os
import subprocess
import sys

import pytest

import stress_cases
from tda_eeg_audio_amd import engine, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def stress(ctx):
    msgs = []
    s = stress_cases.Stress(ctx, engine, synth, seed=2024, scale=1, log=lambda *a: msgs.append(" ".join(map(str, a))))
    yield s
    assert s.bad == 0, "\n".join(msgs[-20:])


@pytest.mark.parametrize("words", [(2, 1), (1, 1)])
def test_stress_audio_windows_all_bands(stress, ctx, words):
    ctx.set_class_words(*words)
    try:
        stress.audio(n_per_band=150)
    finally:
        ctx.set_class_words(2, 1)


@pytest.mark.parametrize("words", [(2, 1), (1, 1), (0, 1)])
def test_stress_matrices_thresholds_and_fused_kernel(stress, ctx, words):
    ctx.set_class_words(*words)
    try:
        stress.matrices(n=120)
    finally:
        ctx.set_class_words(2, 1)


@pytest.mark.parametrize("words", [(2, 1), (1, 1)])
def test_stress_tie_heavy_metrics_and_small_clouds(stress, ctx, words):
    ctx.set_class_words(*words)
    try:
        stress.ties(reps=6)
        stress.clouds(small=12, large=6)
    finally:
        ctx.set_class_words(2, 1)


def test_stress_wasserstein_quantised(stress):
    stress.wasserstein(rounds=4)


def test_lds_guard_build_clean():
    """The guard build: 16 sentinel bytes between every two LDS regions of RipsLayout, written before the sweep and
    checked after it (status bit 0x100), plus the point cloud compared before / after.  Run once, on the inputs that
    found the overrun of round 1 (short last chunks) and on tie-heavy metrics."""
    lib = os.path.join(ROOT, "tda_eeg_audio_amd", "libtdaeeg_dbg.so")
    assert os.path.exists(lib), "guard build missing: python -c 'import __graft_entry__ as g; g.build()'"
    env = dict(os.environ, TDA_STRESS_DEBUG="1", TDA_STRESS_WHAT="clouds", TDA_STRESS_SCALE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_parity.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "STRESS OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bounded_polls_report_instead_of_hanging():
    """The cross-wave hand-overs of the sweep poll LDS words (the turn of phase a, the done flags of phase c).  Every
    poll is bounded (TDA_POLL_LIMIT): with a dependency broken on purpose (guard build, tda_debug_inject) the windows
    come back with TDA_WIN_NOT_CONVERGED instead of hanging the GPU."""
    lib = os.path.join(ROOT, "tda_eeg_audio_amd", "libtdaeeg_dbg.so")
    assert os.path.exists(lib), "guard build missing: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probes", "poll_fault_inject.py")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "POLL GUARD OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
