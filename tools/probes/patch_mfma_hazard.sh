#!/bin/bash
# Proof of cause for the fault of eeg_window_kernel at __launch_bounds__(256, 1) (see csrc/corr_dist_dev.h):
# builds rips.hip with -DTDA_EEG_WIDE_WAVES=1, shows the mis-placed wait in the device assembly, patches
# `s_nop 15; s_nop 3` in front of the first v_accvgpr_read after the last v_mfma_f64 of a tile and links
# tda_eeg_audio_amd/libtdaeeg_w1.so (as compiled) and libtdaeeg_w1p.so (patched).  On the GPU box:
#     python tools/probes/wide_waves_repro.py libtdaeeg_w1.so     -> distances off by up to 4e-2 in every window
#     python tools/probes/wide_waves_repro.py libtdaeeg_w1p.so    -> bit-exact
# Run from the repo root (CPU only: hipcc cross-compiles).
set -e -o pipefail
ROOT=$(pwd)
CSRC=$ROOT/tda_eeg_audio_amd/csrc
make -C $CSRC VARIANT=w1 EXTRA=-DTDA_EEG_WIDE_WAVES=1 -j4 > /dev/null
W=$(mktemp -d)
cd $W
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -DTDA_EEG_WIDE_WAVES=1"
/opt/rocm/bin/hipcc $FLAGS --save-temps -c $CSRC/rips.hip -o rips.o 2> /dev/null
/opt/rocm/bin/hipcc -### $FLAGS --save-temps -c $CSRC/rips.hip -o rips.o 2>&1 | grep '^ "' > cmds.txt
S=rips-hip-amdgcn-amd-amdhsa-gfx950.s
echo "as compiled (first read of the accumulators 4 wait states behind the MFMA, 18 are required):"
grep -n -B1 -A3 "^	s_nop 0$" $S | grep -A4 -B1 "v_accvgpr_read_b32 v[0-9]*, a7" | head -12 || true
python3 - <<'PY'
import re
s = open("rips-hip-amdgcn-amd-amdhsa-gfx950.s").read()
pat = re.compile(r"\ts_nop 0\n\tv_accvgpr_read_b32 (v\d+), (a\d+)\n\ts_nop 11\n")
n = len(pat.findall(s))
s = pat.sub(lambda m: "\ts_nop 15\n\ts_nop 3\n\tv_accvgpr_read_b32 %s, %s\n\ts_nop 11\n" % m.groups(), s)
open("rips-hip-amdgcn-amd-amdhsa-gfx950.s", "w").write(s)
print("patched", n, "places")
PY
# device: assemble, link, bundle; host: compile with the new bundle embedded, assemble
for i in 4 5 6 8 9 10; do sed -n "${i}p" cmds.txt > step.sh; bash step.sh 2> /dev/null; done
cd $CSRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtdaeeg_w1p.so capi.w1.o $W/rips.o corr_dist.w1.o features.w1.o wasserstein.w1.o filters.w1.o
rm -rf $W
echo "built tda_eeg_audio_amd/libtdaeeg_w1.so and libtdaeeg_w1p.so"
