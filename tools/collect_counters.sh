#!/bin/bash
# SQ counters of the Rips kernels -- the resource that binds them is LDS / issue, not HBM.
#   tools/collect_counters.sh <tag>      (on the GPU box, from the repo root)
# Separate --pmc passes (8 SQ slots per pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel-trace only, the
# program directly after `--`, one band batch in flight so that launches map 1:1 to stages.  A reduced corpus
# (236 recordings = 3,540 windows per band batch, 13.8 rounds of the audio kernel) keeps a pass short; the
# counters are per-launch sums, and the ratios reported do not depend on the batch size.
set -e -o pipefail
TAG=${1:-r02}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --no-cpu --no-extras --recordings 236 --steps 2 --warmup 1 --lanes 1 --no-graph"
rocprofv3 -L > $OUT/${TAG}_counters_available.txt 2>&1 || true
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_FLAT SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CU_CYCLES"
i=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_sq$i -o sq -- python3 $ARGS > $OUT/pmc_sq$i.json 2> $OUT/pmc_sq$i.err || echo "pass $i failed (see $OUT/pmc_sq$i.err)"
  echo "SQ pass $i done"
  i=$((i+1))
done
python3 tools/summarize_counters.py $OUT $TAG
