#!/bin/bash
# Kernel trace of ONE rank through the distributed path on RCCL (camera-batch or tile-row) and of the unsharded step, summarised by
# tools/gap_stats.py: which launches and which idle stretches the path adds.   tools/dist_one_rank_trace.sh <camera|tile-row> <out dir>
# (rocprofv3 wraps the rank directly: the launcher variables are exported here, so bench.py does not spawn.)
mode=${1:-camera}
out=$PWD/${2:-gpurun_out/dist_trace}
root=$PWD
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--no-extra --no-cpu-baseline --steps 150 --warmup 30"
rocprofv3 --kernel-trace --output-format csv -d $out/plain -o run -- python3 $root/bench.py $args > /dev/null 2> $out/plain.err
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 LOCAL_WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 MOM_FORCE_DIST=1
rocprofv3 --kernel-trace --output-format csv -d $out/dist -o run -- python3 $root/bench.py --shard $mode $args > /dev/null 2> $out/dist.err
cd $root
for w in plain dist; do
  f=$(find $out/$w -name "*kernel_trace.csv" | head -1)
  echo "== $w ($mode)"; python3 tools/gap_stats.py $f 100 | head -45 > $out/gap_$w.txt; head -45 $out/gap_$w.txt
  rm -rf $out/$w
done
