#!/bin/bash
# Collect the rocprofv3 evidence that profiles/ holds (run on the GPU box from the repo root):
#   tools/collect_profiles.sh <tag>        e.g. r02
# 1. per-kernel time: --kernel-trace --stats of the default bench workload (corpus, 10 timed passes; no secondary legs, so
#    that the launches of the dominant kernel are exactly the ones bench.py's own probe averages: kernel_ms_all_launches)
# 2. HBM bytes: --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (kernel-trace only, as the MI355X guide
#    prescribes), one band batch in flight and eager launches so that launches map 1:1 to stages
# 3. SQ counters of the Rips kernels (what binds them is LDS / issue, not HBM): three more --pmc passes, 8 SQ slots each
# Passes 2-3 run a reduced corpus, one batch per band (236 recordings = 3,540 windows per launch, 6.9 rounds of the audio
# kernel at two workgroups per CU):
# the counters are per-launch sums and the figures reported are per window or ratios.
# The program sits directly after `--` in every pass.  Summaries go to gpurun_out/<tag>_*; copy what is to be judged
# into profiles/.
set -e -o pipefail
TAG=${1:-r03}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
SMALL="bench.py --no-cpu --no-extras --recordings 236 --steps 2 --warmup 1 --lanes 1 --no-graph --per-band"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- python3 bench.py --no-cpu --no-extras --steps 10 --warmup 2 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/prof_stats.err
echo "stats pass done"
# 1b. the same with ONE pass in flight and eager launches: the dominant kernel's launches do not overlap the kernels of
#     other passes, so its average duration in this CSV fits inside ms_per_step and `roofline.frac` can be recomputed
#     from the tracked file alone (alg_bytes_per_launch / avg duration / 8 TB/s)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats1 -o stats1 -- python3 bench.py --lanes 1 --no-graph --no-cpu --no-extras --steps 10 --warmup 2 > $OUT/${TAG}_bench_lanes1_under_rocprof.json 2> $OUT/prof_stats1.err
echo "single-lane stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -o fetch -- python3 $SMALL > /dev/null 2> $OUT/prof_fetch.err
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -o write -- python3 $SMALL > /dev/null 2> $OUT/prof_write.err
echo "WRITE_SIZE pass done"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_FLAT SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CU_CYCLES"
i=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_sq$i -o sq -- python3 $SMALL > /dev/null 2> $OUT/pmc_sq$i.err || echo "SQ pass $i failed (see $OUT/pmc_sq$i.err)"
  echo "SQ pass $i done"
  i=$((i+1))
done
python3 tools/summarize_profiles.py $OUT $TAG
python3 tools/summarize_counters.py $OUT $TAG
# the raw traces are large; the summaries above are what travels back
rm -rf $OUT/prof_stats/*/ $OUT/prof_stats1/*/ $OUT/prof_fetch $OUT/prof_write $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_sq3
find $OUT/prof_stats $OUT/prof_stats1 -name "*trace*" -delete 2>/dev/null || true
