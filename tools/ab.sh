#!/bin/bash
# A/B of a variant library (tools/variants.sh) against the current build, on the GPU box:  tools/ab.sh <variant> [kernel slot] [config]
# Three alternating runs of the headline step; prints steps/s and the named kernel's average launch (HIP events).
v=$1; k=${2:-render_bwd}; cfg=${3:-c2}
for i in 1 2 3; do
  for lib in var/$v.so libmom4d.so; do
    MOM4D_LIB=iclr2025_3d-mom_amd/lib/$lib MOM4D_LIB_LAX=1 python bench.py --config $cfg --no-cpu-baseline --no-extra --steps ${STEPS:-300} --warmup ${WARM:-50} --roofline-kernel $k 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print('$lib', round(d['value'],1), 'steps/s;', '$k', round(r.get('avg_launch_us',0),1), 'us')
"
  done
done
