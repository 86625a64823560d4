#!/bin/bash
# tools/ab_lib.sh VARIANT.so [config] [rounds]: the bench's main leg with the shipped library and with lib/var/VARIANT.so, alternating, one call
v=$1; cfg=${2:-c2}; n=${3:-3}
L=iclr2025_3d-mom_amd/lib
extra=""; [ "$cfg" != "c2" ] && extra="--config $cfg --steps 40 --warmup 10"
[ "$cfg" = "c2" ] && extra="--steps 200 --warmup 50"
for r in $(seq $n); do for lib in $L/libmom4d.so $L/var/$v; do echo -n "$(basename $lib) "; MOM4D_LIB=$lib python bench.py --no-extra --no-cpu-baseline $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],4))"; done; done
