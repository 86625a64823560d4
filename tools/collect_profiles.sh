#!/bin/bash
# Collects every profile the round commits, in one GPU-box call:  tools/collect_profiles.sh <prefix, e.g. r04_a> [config, default c2]
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench.py command        -> profiles/<prefix>_kernel_stats.csv (+ the bench line)
#   2. two separate --pmc passes FETCH_SIZE / WRITE_SIZE (only --kernel-trace)  -> profiles/<prefix>_pmc_traffic.{json,_per_kernel.csv}
#   3. one --pmc pass with the SQ counters + GRBM_GUI_ACTIVE                    -> profiles/<prefix>_sq.json, <prefix>_mfma_util.json
# The program is always directly after `--`; counter passes never carry a runtime / hip / hsa trace.
# Another config (c3, c5: BASELINE configs[2], [4]) is profiled with `--config`; its steps are longer, so fewer are taken.
set -u
pre=$1
cfg=${2:-c2}
root=$PWD
out=$root/gpurun_out/$pre
mkdir -p $out $root/profiles
cd /tmp && export TMPDIR=/tmp
if [ "$cfg" = "c2" ]; then main_args="--no-cpu-baseline --no-side-legs"; pmc_args="--steps 20 --warmup 5 --no-cpu-baseline --no-extra"
else main_args="--config $cfg --no-cpu-baseline --no-extra --steps 40 --warmup 10"; pmc_args="--config $cfg --steps 12 --warmup 6 --no-cpu-baseline --no-extra"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 $root/bench.py $main_args > $out/bench_default.json 2> $out/stats.err
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${pre}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o run -- python3 $root/bench.py $pmc_args > $out/pmc_$c.json 2> $out/pmc_$c.err
  cp $(find $out/pmc_$c -name "*counter_collection.csv" | head -1) $out/${c}_counter_collection.csv
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -o run -- python3 $root/bench.py $pmc_args > $out/pmc_sq.json 2> $out/pmc_sq.err
cp $(find $out/pmc_sq -name "*counter_collection.csv" | head -1) $out/sq_counter_collection.csv
cp $(find $out/pmc_sq -name "*kernel_trace.csv" | head -1) $out/sq_kernel_trace.csv
rm -rf $out/stats $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_sq
cd $root
name=$(python3 -c "import bench; print(bench.CONFIGS['$cfg']['name'])")
python3 tools/pmc_traffic.py $out/FETCH_SIZE_counter_collection.csv $out/WRITE_SIZE_counter_collection.csv $out/$pre $cfg 2>&1 | tail -25
python3 tools/pmc_sq.py $out/sq_counter_collection.csv --json $out/${pre}_sq.json --workload "$name" | head -14
if [ "$cfg" = "c2" ]; then python3 tools/pmc_mfma.py $out/sq_counter_collection.csv $out/sq_kernel_trace.csv $out/$pre 2>&1 | tail -6; fi
tail -c 2500 $out/bench_default.json
# the raw per-dispatch counter tables are tens of MB per config; only the summaries travel back
rm -f $out/*_counter_collection.csv $out/sq_kernel_trace.csv
