#!/bin/bash
# SQ-counter pass(es) of a short bench.py run: tools/pmc_sq.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...]
# -> gpurun_out/<tag>/pass<i>_counter_collection.csv (+ a per-kernel mean table on stdout via tools/pmc_sq.py).
# --pmc passes carry only --kernel-trace (never a runtime / hip / hsa trace), the program directly after `--`.
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pass$i -o run -- python3 $root/bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 > $out/pass$i.json 2> $out/pass$i.err
  f=$(find $out/pass$i -name "*counter_collection.csv" | head -1)
  cp "$f" $out/pass${i}_counter_collection.csv
  rm -rf $out/pass$i
done
python3 $root/tools/pmc_sq.py $out/pass*_counter_collection.csv
