"""CPU tests: the C-ABI library loads and exports every symbol include/tdaeeg.h declares (no
compute calls), the product package never touches oracle/, and the host-side logic (window
selection, sharding, packing, the mirrors' pure-index functions) matches the reference."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "tdaeeg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tda_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from tda_eeg_audio_amd import _lib
    names = _declared()
    assert len(names) >= 28
    lib = _lib.load()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in tdaeeg.h but not exported by libtdaeeg.so"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes signature"
    assert set(_lib.SYMBOLS) <= set(names)
    assert lib.tda_version() >= 100


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    from tda_eeg_audio_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.TdaError):
        _lib.Context(0)
    from tda_eeg_audio_amd import utils
    with pytest.raises(_lib.TdaError):
        utils.compute_eeg_persistence(np.zeros((4, 4)))


def test_product_package_never_uses_oracle():
    pkg = os.path.join(ROOT, "tda_eeg_audio_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
                assert "libtda_oracle" not in txt, f


def test_window_selection_matches_reference(golden):
    from tda_eeg_audio_amd import drivers
    for k, n in enumerate(range(16, 90)):
        assert np.array_equal(drivers.select_windows_even(n), golden["sel_linspace"][k])
    assert np.array_equal(drivers.select_windows_even(9), np.arange(9))
    assert np.array_equal(drivers.select_windows_md5("bb01_ut01", "delta", 71, 39), golden["sel_md5"][0])
    assert np.array_equal(drivers.select_windows_md5("bb17_ut09", "gamma", 45, 39), golden["sel_md5"][1])


def test_pure_index_mirrors_match_reference(golden):
    from tda_eeg_audio_amd import utils
    assert np.array_equal(utils.create_windows(golden["cw_signal"], 250, 62), golden["cw_windows"])
    assert utils.create_windows(golden["cw_signal"][:100], 250, 62).shape == tuple(golden["cw_empty_shape"])
    for name, s, t in zip(utils.FREQ_BANDS, golden["tau_signals"], golden["tau_values"]):
        assert np.array_equal(utils.takens_embedding(s, 3, int(t), 2), golden["tk_pc_" + name])
    assert utils.takens_embedding(golden["tau_signals"][0], 3, 125, 2).shape == tuple(golden["tk_empty_shape"])
    assert np.array_equal(utils.takens_embedding(golden["tau_signals"][1], 3, 5, 1), golden["tk_nosub"])
    import json
    c = json.loads(str(golden["const"]))
    assert (utils.MAX_DIM, utils.MAX_EDGE_LENGTH, utils.TAKENS_DIM, utils.TAKENS_SUBSAMPLE) == \
        (c["MAX_DIM"], c["MAX_EDGE_LENGTH"], c["TAKENS_DIM"], c["TAKENS_SUBSAMPLE"])
    assert {k: tuple(v) for k, v in c["FREQ_BANDS"].items()} == utils.FREQ_BANDS
    assert (utils.FS_AUDIO, utils.FS_EEG) == (c["FS_AUDIO"], c["FS_EEG"])


def test_feature_names_match_reference_file():
    from tda_eeg_audio_amd import drivers
    ref = os.path.join("/root/reference", "features", "feature_names.txt")
    names = drivers.feature_names()
    assert len(names) == 220 and names[0] == "delta_h0_n_features_mean" and names[3] == "delta_h1_n_features_std"
    assert names[-1] == "gamma_h1_persistence_entropy_std"
    if os.path.exists(ref):          # only in the build container; the GPU box has no reference
        assert names == open(ref).read().split()


def test_pack_diagrams_and_sharding():
    from tda_eeg_audio_amd import dist, engine
    rows, cnt = engine.pack_diagrams([np.zeros((0, 2)), np.array([[0.0, 1.0], [0.5, np.inf]]), np.array([1.0, 2.0])])
    assert rows.shape == (3, 2, 2) and list(cnt) == [0, 2, 0]
    nwin = np.array([89] * 710 + [45] * 706)
    shards = dist.shard_recordings(nwin, 8)
    allr = np.sort(np.concatenate(shards))
    assert np.array_equal(allr, np.arange(1416)) and all(len(s) == 177 for s in shards)
    loads = np.array([nwin[s].sum() for s in shards])
    assert loads.max() - loads.min() <= 89
    assert [len(s) for s in dist.shard_recordings(np.arange(5), 8)] == [1, 1, 1, 1, 1, 0, 0, 0]


_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from tda_eeg_audio_amd import dist as tdist
rank, world, local = tdist.init_from_env(backend="gloo")
n_total = 11
nwin = np.array([80, 45, 71, 39, 89, 50, 60, 41, 77, 49, 66])
shards = tdist.shard_recordings(nwin, world)
mine = shards[rank]
# fake per-recording result rows: row r = [r, 10 r, ...] so that the gathered matrix is checkable
local_rows = torch.tensor([[float(r) * (k + 1) for k in range(6)] for r in mine], dtype=torch.float64).reshape(len(mine), 6)
out = tdist.all_gather_rows(local_rows, mine, shards, n_total)
exp = torch.tensor([[float(r) * (k + 1) for k in range(6)] for r in range(n_total)], dtype=torch.float64)
assert torch.equal(out, exp), (rank, out)
dist.barrier()
if rank == 0:
    print("GATHER_OK")
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 3])
def test_all_gather_rows_gloo(world, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = 29600 + world + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GATHER_OK" in r.stdout


def test_validate_distance_matrix_matches_reference():
    """drivers.validate_distance_matrix against the answers of the reference's own function
    (scripts/tda_eeg_classification_v2.py:110-140; fixture made by tests/golden/make_golden_drivers.py)."""
    import json
    from tda_eeg_audio_amd import drivers
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_golden_drivers.npz"))
    answers = json.loads(str(g["vdm_answers"]))
    assert len(answers) == 11
    with np.errstate(all="ignore"):
        for k, exp in answers.items():
            ok, issues = drivers.validate_distance_matrix(g[f"vdm_{k}"], k)
            assert ok == exp["valid"] and issues == exp["issues"], (k, issues, exp)


def test_detailed_rows_have_the_reference_csv_columns():
    """cmp:145-157: the header of the reference's results/eeg_audio_tda_detailed.csv (repeated here as data)."""
    from tda_eeg_audio_amd import drivers, engine
    header = ("filename,condition,subject,band,wasserstein_h0,wasserstein_h1,n_windows,tau,corr_mean_persistence_r,"
              "corr_mean_persistence_p,corr_total_persistence_r,corr_total_persistence_p,corr_persistence_entropy_r,"
              "corr_persistence_entropy_p,corr_max_persistence_r,corr_max_persistence_p,corr_n_features_r,corr_n_features_p")
    assert drivers.DETAILED_COLUMNS == header.split(",")
    res = [{"filename": "bb01_ut01.mat", "condition": "slow", "subject": "bb01", "bands": {
        "delta": {"wasserstein_h0": 13.25, "wasserstein_h1": 0.81, "n_windows": 15, "tau": 84,
                  "feature_correlations": {f: {"r": 0.1 * i, "p": 0.5} for i, f in enumerate(engine.SPEARMAN_FEATURES)}}}}]
    rows = drivers.detailed_rows(res)
    assert len(rows) == 1 and list(rows[0]) == drivers.DETAILED_COLUMNS
    assert rows[0]["corr_n_features_r"] == pytest.approx(0.4) and rows[0]["tau"] == 84


def test_oracle_segment_step_equals_python_restatement():
    """orc_segment_step (the C unit bench.py's cpu_baseline times) == oracle/pipeline_ref.py, incl. a group whose
    Takens clouds have fewer than 3 points: its windows leave the distances (cmp:90-91), the band gets NaN."""
    from oracle import pipeline_ref, port
    from tda_eeg_audio_amd import synth
    eeg = synth.eeg_windows(10, seed=3, windows_per_recording=5)
    aud = synth.audio_windows(10, "theta", seed=5)
    t = np.arange(250) / 250.0
    aud[5:] = 1.0 - t[None, :] * 0.5                    # slow ramp: no zero crossing of the autocorrelation below lag 125
    ref = pipeline_ref.reference_step_cpu(eeg, aud, [0, 5, 10])
    rows = np.stack([port.segment_step(eeg[:5], aud[:5]), port.segment_step(eeg[5:], aud[5:])])
    assert np.array_equal(ref, rows, equal_nan=True)
    assert ref[1, 2] > 60 and np.isfinite(ref[1, :2]).all()       # a long delay, small clouds, still >= 3 points
    # tau = 124 -> P = 1 < 3: every window skipped, distances NaN, features still aggregated
    port_tau = port.compute_tau
    try:
        port.compute_tau = lambda s, max_lag=None: 124
        r = pipeline_ref.reference_step_cpu(eeg[:5], aud[:5], [0, 5])
    finally:
        port.compute_tau = port_tau
    assert np.isnan(r[0, :2]).all() and r[0, 2] == 124 and np.isfinite(r[0, 4:]).all()


def test_bench_launch_contract_without_gpu():
    """bench.py --gpus N: WORLD_SIZE != N is an error (exit 2), and a plain `--gpus 2` on a box with fewer GPUs
    refuses before starting any rank -- it never silently measures one GPU."""
    import torch
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1 but --gpus 2" in r.stderr
    if torch.cuda.device_count() < 2:
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 2 and "GPU(s) visible" in r.stderr


def test_mfma_kernels_keep_accumulators_in_vgprs(tmp_path):
    """Condition under which the f64 MFMA code of the product is known to be exact (csrc/corr_dist_dev.h, COMPILER
    FAULT): hipcc mis-places the wait in front of the first v_accvgpr_read after a v_mfma_f64 when the accumulators
    live in AGPRs.  No kernel of the shipped code objects that runs MFMAs may use AGPRs (the filter kernels, which have
    no MFMA, may use them as plain extra registers)."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    lib = os.path.join(ROOT, "tda_eeg_audio_amd", "libtdaeeg.so")
    if not (os.path.exists(objdump) and os.path.exists(readelf) and os.path.exists(lib)):
        pytest.skip("ROCm binutils or the built library are not here")
    shutil.copy(lib, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", "lib.so"], cwd=tmp_path, check=True, capture_output=True)
    seen = 0
    for f in sorted(os.listdir(tmp_path)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([readelf, "--notes", f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        counts = re.findall(r"\.agpr_count:\s+(\d+)\s+(?:.*\n)*?\s+\.name:\s+(\S+)", notes)
        for n, name in counts:
            if "eeg_window_kernel" in name or "corr_dist_kernel" in name:       # the kernels that run v_mfma_f64
                seen += 1
                assert int(n) == 0, f"{name} uses {n} AGPRs"
    assert seen >= 10, seen          # every variant of the two MFMA kernels was looked at
