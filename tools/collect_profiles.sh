#!/bin/bash
# Collect the rocprofv3 evidence that profiles/ holds (run on the GPU box from the repo root):
#   tools/collect_profiles.sh <tag>        e.g. r01
# 1. per-kernel time: --kernel-trace --stats of the default bench command
# 2. HBM bytes: --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (kernel-trace only, as the
#    MI355X guide prescribes), one batch in flight so that launches map 1:1 to stages
# Summaries go to gpurun_out/<tag>_* ; copy what is to be judged into profiles/.
set -e -o pipefail
TAG=${1:-r01}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- python3 bench.py --no-cpu --steps 40 --warmup 4 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/prof_stats.err
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -o fetch -- python3 bench.py --no-cpu --steps 4 --warmup 1 --lanes 1 > /dev/null 2> $OUT/prof_fetch.err
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -o write -- python3 bench.py --no-cpu --steps 4 --warmup 1 --lanes 1 > /dev/null 2> $OUT/prof_write.err
echo "WRITE_SIZE pass done"
python3 tools/summarize_profiles.py $OUT $TAG
