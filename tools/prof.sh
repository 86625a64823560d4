#!/bin/bash
# Per-kernel time of a bench.py run under rocprofv3 (kernel trace + stats), top kernels printed; the stats CSV is kept
# under gpurun_out/<tag>/ for copying into profiles/.   usage: tools/prof.sh <tag> [bench.py args...]
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 $OLDPWD/bench.py --no-extra --no-cpu-baseline "$@" > $out/bench.json 2> $out/err.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp "$f" $out/kernel_stats.csv 2>/dev/null
python3 - "$out/kernel_stats.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"\(.*", "", name)[:60]
    print(f'{name:60s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:9.1f} us  {float(r["Percentage"]):5.2f} %')
PY
tail -c 400 $out/bench.json | head -c 0
python3 -c "
import json,sys
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
print('steps/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4))
"
