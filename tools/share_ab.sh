#!/bin/bash
# A/B of the step's stream layout (TDA_OVERLAP=1: EEG chain forked onto a side stream) and of the lanes in flight, on
# the share a rank of eight holds (177 recordings), the full corpus, configs[1] and the raw-recordings leg.
# bash tools/share_ab.sh        (GPU box, repo root)
mkdir -p gpurun_out
run() {
    env "$1" timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 40 $2 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$1 $2: failed"; tail -3 gpurun_out/ab.err; return; }
    python -c "
import json;d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]);print('$1 $2:',round(d['value']),round(d['ms_per_step'],3),'ms')"
}
for rec in 177 1416; do
    for ov in 0 1; do
        for lanes in 2 3 5; do
            run TDA_OVERLAP=$ov "--recordings $rec --lanes $lanes"
        done
    done
done
for ov in 0 1; do
    run TDA_OVERLAP=$ov "--workload batch710"
    run TDA_OVERLAP=$ov "--per-band"
    env TDA_OVERLAP=$ov timeout -k 10 300 python tools/recordings_bench.py 2>&1 | tail -2
done
