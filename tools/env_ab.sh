#!/bin/bash
# tools/env_ab.sh VAR=VALUE [steps]: three alternating runs of the headline step with and without an environment setting
kv=$1; n=${2:-400}
for i in 1 2 3; do
  for on in 0 1; do
    if [ $on = 1 ]; then export "$kv"; else unset "${kv%%=*}"; fi
    python bench.py --no-cpu-baseline --no-extra --steps $n --warmup 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$on' == '1' and '$kv' or 'default', round(d['value'],1), 'steps/s')
"
  done
done
