#!/bin/bash
# tools/ab_env.sh VAR A B [rounds]: the default bench's headline (200 steps after 50) with VAR=A and VAR=B alternating in one call
var=$1; a=$2; b=$3; n=${4:-3}
for r in $(seq $n); do for m in $a $b; do echo -n "$var=$m "; env $var=$m python bench.py --no-extra --no-cpu-baseline --steps 200 --warmup 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4))"; done; done
