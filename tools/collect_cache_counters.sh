#!/bin/bash
# Cache / TLB / wait counters per kernel, for the question "what holds the gathers on cache-resident planes to 2.6-2.8 TB/s at scale?"
# (VERDICT r4 item 6):  tools/collect_cache_counters.sh <prefix, e.g. r05_c5> [config, default c5]
# Separate --pmc passes with --kernel-trace only; the program directly after `--`.  Summary -> profiles/<prefix>_cache.json
set -u
pre=$1
cfg=${2:-c5}
root=$PWD
out=$root/gpurun_out/$pre
mkdir -p $out $root/profiles
cd /tmp && export TMPDIR=/tmp
args="--config $cfg --steps 8 --warmup 4 --no-cpu-baseline --no-extra"
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o run -- python3 $root/bench.py $args > $out/p$i.json 2> $out/p$i.err
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp $f $out/pass${i}_counter_collection.csv; else echo "pass $i ($set): no counter file"; tail -3 $out/p$i.err; fi
  rm -rf $out/p$i
done
cd $root
name=$(python3 -c "import bench; print(bench.CONFIGS['$cfg']['name'])")
python3 tools/pmc_sq.py $out/pass*_counter_collection.csv --json profiles/${pre}_cache.json --workload "$name" | grep -i "deform_field_fwd_b3\|hexplane_bwd6_gather\|hexplane_bwd5_scatter\|deform_bwd_b3g\|adam_kernel" | cut -c1-900
rm -f $out/pass*_counter_collection.csv
