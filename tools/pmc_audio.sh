#!/bin/bash
# SQ / instruction-cache counters of the audio Rips stage alone (tools/audio_stage_bench.py), one --pmc pass per group.
#   tools/pmc_audio.sh <tag> [env assignments, e.g. TDA_CLOUD_WIDE_FIRST=1]
set -e -o pipefail
TAG=$1; shift || true
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
G2="SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_WAVES"
G3="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
i=1
for G in "$G1" "$G2" "$G3"; do
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/g$i -o p -- python3 tools/audio_stage_bench.py > $OUT/run$i.txt 2> $OUT/err$i.txt || echo "pass $i failed"
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "rips_cloud_kernel" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
res = {}
for k, c in acc.items():
    w = c.get("SQ_WAVE_CYCLES", 0) or 1
    res[k] = {kk: vv for kk, vv in c.items()}
    res[k]["_wait_any_frac"] = c.get("SQ_WAIT_ANY", 0) / w
    res[k]["_wait_inst_frac"] = c.get("SQ_WAIT_INST_ANY", 0) / w
    res[k]["_active_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0) / w
    res[k]["_valu_frac"] = c.get("SQ_ACTIVE_INST_VALU", 0) / w
    if c.get("SQC_ICACHE_REQ"):
        res[k]["_icache_miss_rate"] = c.get("SQC_ICACHE_MISSES", 0) / c["SQC_ICACHE_REQ"]
    if c.get("SQ_LDS_IDX_ACTIVE"):
        res[k]["_lds_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for k, v in res.items():
    print(k[:70], {a: round(b, 4) for a, b in v.items() if a.startswith("_")}, "VALU", v.get("SQ_INSTS_VALU"), "LDS", v.get("SQ_INSTS_LDS"), "SALU", v.get("SQ_INSTS_SALU"), "IFETCH", v.get("SQ_IFETCH"), "waves", v.get("SQ_WAVES"))
PY
rm -rf $OUT/g1 $OUT/g2 $OUT/g3
