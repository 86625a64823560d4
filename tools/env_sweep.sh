#!/bin/bash
# tools/env_sweep.sh VAR "v1 v2 ..." [config] [steps] [warmup]: the headline step under each value of an environment variable ("-" = unset), twice
var=$1; vals=$2; cfg=${3:-c2}; n=${4:-300}; w=${5:-50}
for i in 1 2; do
  for v in $vals; do
    if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
    python bench.py --config $cfg --no-cpu-baseline --no-extra --steps $n --warmup $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', round(d['value'],1), 'steps/s')
"
  done
done
