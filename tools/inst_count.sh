#!/bin/bash
# Vector / scalar / LDS wave-instructions per window of the two big kernels (run on the GPU box from the repo root):
#   tools/inst_count.sh [tag]
# One rocprofv3 --pmc pass over a reduced corpus (236 recordings, one batch per band, eager launches).  The counts are
# exact and repeatable to the instruction, which the wall clock is not: this is the yardstick for instruction-count work.
set -e -o pipefail
TAG=${1:-count}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/pmc_cnt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_cnt -o cnt -- \
  python3 bench.py --no-cpu --no-extras --recordings 236 --steps 1 --warmup 1 --lanes 1 --no-graph --per-band > /dev/null 2> $OUT/pmc_cnt.err
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
for path in glob.glob(os.path.join(out, "pmc_cnt", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").strip()
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
lines = []
# window pairs of the run = workgroups of the audio first pass (8 waves each); every kernel's counts are divided by it
# (the finishing pass and the H1 Wasserstein distances take two launches each since round 3: their sums are what counts)
n_pairs = next(c["SQ_WAVES"] / 8 for name, c in acc.items() if name.startswith("rips_cloud_kernel<512, 1, unsigned int, false") and c.get("SQ_WAVES"))
for k in ("rips_cloud_kernel<512, 1, unsigned int, false", "eeg_window_kernel<3, false, 1, false", "wasserstein_kernel<2",
          "wasserstein_kernel<1", "wasserstein_kernel<4", "diagram_finish_kernel", "rips_cloud_kernel<512, 1, unsigned long long, true"):
    for name, c in acc.items():
        if name.startswith(k) and c.get("SQ_WAVES"):
            lines.append(f"{name[:60]:60s} windows={int(n_pairs):7d} valu/win={c['SQ_INSTS_VALU']/n_pairs:9.1f} salu/win={c['SQ_INSTS_SALU']/n_pairs:9.1f} "
                         f"lds/win={c['SQ_INSTS_LDS']/n_pairs:8.1f}")
print("\n".join(lines))
open(os.path.join(out, f"{tag}_inst_count.txt"), "w").write("\n".join(lines) + "\n")
PY
rm -rf $OUT/pmc_cnt
