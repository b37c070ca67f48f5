"""
tda_eeg_audio_amd -- MI355X-native persistent-homology feature engine for the per-window hot
path of Ignaciagothe/tda-eeg-audio (corr->distance, Takens, Rips H0/H1, features, Wasserstein).

The arithmetic lives in hand-written HIP kernels behind the C ABI of include/tdaeeg.h
(tda_eeg_audio_amd/libtdaeeg.so).  ``utils`` and ``graphs`` mirror the reference's function
names (scripts/utils.py, notebooks/2_graph_construction.ipynb) so its drivers run unchanged.
"""
from . import _lib  # noqa: F401
