#!/bin/bash
# A/B of two builds of the library on the bench's pass, alternating, in one call (boxes differ by ~10 %: never compare
# across calls).   bash tools/lib_ab.sh libtdaeeg_prev.so libtdaeeg.so [bench flags]
A=$1; B=$2; shift 2
mkdir -p gpurun_out
for rep in 1 2 3; do
    for l in $A $B; do
        TDA_LIB=$l timeout -k 10 200 python bench.py --no-cpu --no-extras --steps 40 "$@" > gpurun_out/lab.json 2> gpurun_out/lab.err || { echo "$l failed"; tail -3 gpurun_out/lab.err; continue; }
        python -c "
import json;d=json.loads(open('gpurun_out/lab.json').read().strip().splitlines()[-1]);print('$l',round(d['value']),round(d['ms_per_step'],3),'ms')"
    done
done
