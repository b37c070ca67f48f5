#!/bin/bash
# The raw-recordings leg against the number of buffer sets (shards in flight) and the shard size, one call.
# bash tools/recordings_ab.sh
for sets in 2 3 4; do
    for shard in 118 177 236 354; do
        echo -n "sets $sets shard $shard: "
        TDA_REC_SETS=$sets timeout -k 10 200 python tools/recordings_bench.py 1416 $shard 2>&1 | tail -1 | cut -c1-120
    done
done
