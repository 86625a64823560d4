#!/bin/bash
# tools/abn.sh <kernel slot> <variant> [<variant> ...]: two rounds over the current build and every named variant library
k=$1; shift
for i in 1 2; do
  for lib in libmom4d.so "$@"; do
    [ "$lib" = libmom4d.so ] && path=iclr2025_3d-mom_amd/lib/libmom4d.so || path=iclr2025_3d-mom_amd/lib/var/$lib.so
    MOM4D_LIB=$path MOM4D_LIB_LAX=1 python bench.py --no-cpu-baseline --no-extra --steps 300 --warmup 50 --roofline-kernel $k 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print('$lib', round(d['value'],1), 'steps/s;', '$k', round(r.get('avg_launch_us',0),1), 'us')
"
  done
done
