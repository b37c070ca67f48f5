#!/bin/bash
# tools/inst_split.py under rocprofv3: cumulative vector instructions per window after each phase of the audio kernel
set -e -o pipefail
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; rm -rf $OUT/pmc_split
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_split -o s -- python3 tools/inst_split.py > /dev/null 2> $OUT/pmc_split.err
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
rows = defaultdict(dict)
for path in glob.glob(os.path.join(out, "pmc_split", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "rips_cloud_kernel<512, 1, unsigned int, false" in r["Kernel_Name"] or "rips_cloud_kernel<384, 1, unsigned int, false" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
stops = ["keys", "ranking", "a:mask", "b:cand-barrier", "b:walk", "b:publish", "c:deps", "d:closure", "d:list", "d:reduce", "d:table", "all"]
if os.environ.get("TDA_SPLIT_PHASES_OF"):
    k = int(os.environ["TDA_SPLIT_PHASES_OF"])
    stops = [f"chunk {k - 1}", "a:mask", "b:cand-barrier", "b:walk", "b:publish", "c:deps", "d:closure", "d:list", "d:reduce", "d:table",
             f"chunk {k} end"]
elif os.environ.get("TDA_SPLIT_CHUNKS"):
    stops = ["keys", "ranking"] + [f"chunk {c}" for c in range(int(os.environ["TDA_SPLIT_CHUNKS"]))] + ["all"]
ids = sorted(rows)
txt = []
for bi, band in enumerate(["beta", "delta"]):
    prev = 0.0
    for si, name in enumerate(stops):
        c = rows[ids[bi * len(stops) + si]]
        n = c["SQ_WAVES"] / 8.0
        v = c["SQ_INSTS_VALU"] / n
        txt.append(f"{band:6s} after {name:16s} valu/win={v:9.1f} (+{v - prev:8.1f})  salu/win={c['SQ_INSTS_SALU']/n:9.1f} lds/win={c['SQ_INSTS_LDS']/n:8.1f}")
        prev = v if name != "d:table" else prev
print("\n".join(txt))
open(os.path.join(out, "inst_split.txt"), "w").write("\n".join(txt) + "\n")
PY
rm -rf $OUT/pmc_split
