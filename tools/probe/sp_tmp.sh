#!/bin/bash
bash tools/collect_profiles.sh r06_c3 c3 > gpurun_out/r06_c3_collect.log 2>&1
bash tools/collect_profiles.sh r06_c5 c5 > gpurun_out/r06_c5_collect.log 2>&1
MOM_BENCH_FULL=gpurun_out/r06_bench_full.json python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err
bash tools/dist_one_rank.sh gpurun_out/r06_dist > gpurun_out/r06_dist.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_smoke.log 2>&1
tail -2 gpurun_out/r06_smoke.log
tail -c 1500 gpurun_out/r06_bench_line.json
