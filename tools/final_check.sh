#!/bin/bash
# What a round ends with, in one GPU call: the GPU suite, the guard-build stress run, the bench and its small-share and
# configs[1] modes; stops at the first GPU fault.   bash tools/final_check.sh
mkdir -p gpurun_out
fault() { if grep -q "Memory access fault" "$1" 2>/dev/null; then echo "GPU FAULT in $1"; exit 1; fi; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/fc_pytest.txt 2>&1; tail -2 gpurun_out/fc_pytest.txt; fault gpurun_out/fc_pytest.txt
timeout -k 10 300 python bench.py --no-cpu --steps 40 > gpurun_out/fc_bench.json 2> gpurun_out/fc_bench.err; fault gpurun_out/fc_bench.err
python -c "
import json;d=json.loads(open('gpurun_out/fc_bench.json').read().strip().splitlines()[-1]);print('corpus:',round(d['value']),round(d['ms_per_step'],3),'ms; raw recordings',round(d['recordings_from_host']['value']),'; EEG-only',round(d['features_pass']['value']))"
timeout -k 10 100 python bench.py --no-cpu --no-extras --recordings 177 --steps 40 2> gpurun_out/fc_177.err | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('an eighth:',round(d['value']),round(d['ms_per_step'],3),'ms')"; fault gpurun_out/fc_177.err
timeout -k 10 100 python bench.py --no-cpu --no-extras --workload batch710 --steps 20 2> gpurun_out/fc_710.err | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('batch710:',round(d['value']),round(d['ms_per_step'],3),'ms')"; fault gpurun_out/fc_710.err
TDA_STRESS_DEBUG=1 timeout -k 10 400 python tools/stress_parity.py 2>&1 | tail -1
