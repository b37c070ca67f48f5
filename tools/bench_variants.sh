#!/bin/bash
# Sanity pass over bench.py's other modes (one short run each): value, ms per pass, repaired windows, finite rows.
# Run from the repo root on the GPU box:  bash tools/bench_variants.sh
mkdir -p gpurun_out
for a in "--workload batch710" "--per-band" "--lanes 2 --no-graph" "--recordings 177"; do
    timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 10 $a > gpurun_out/bv.json 2> gpurun_out/bv.err
    echo "$a: rc=$?"
    python - <<'PY'
import json
d = json.loads(open("gpurun_out/bv.json").read().strip().splitlines()[-1])
c = d["config"]
print("   ", round(d["value"]), d["unit"], round(d["ms_per_step"], 2), "ms", c.get("windows_repaired"), c.get("result_rows_finite_frac"))
PY
done
TDA_FUSED_EEG=0 timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 5 > gpurun_out/bv.json 2> gpurun_out/bv.err
echo "two-kernel EEG: rc=$?"
python -c "
import json;d=json.loads(open('gpurun_out/bv.json').read().strip().splitlines()[-1]);print('   ',round(d['value']),round(d['ms_per_step'],2),'ms')"
