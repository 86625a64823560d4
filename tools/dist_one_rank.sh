#!/bin/bash
# One rank through the distributed code path on RCCL (MOM_BENCH_SPAWN=1 -> launch.py spawns one rank with MOM_FORCE_DIST=1), both shard
# modes, on the direct RCCL transport (mom_comm_*) and through torch.distributed (MOM_COMM=torch), plus the sharded-Adam variant, against
# the unsharded step: what the path itself costs before any wire time.
#   tools/dist_one_rank.sh gpurun_out/r05_2
out=${1:-gpurun_out/dist}
mkdir -p $out
args="--no-extra --no-cpu-baseline --steps 200 --warmup 50"
python bench.py $args > $out/plain.json 2>/dev/null
for m in camera tile-row; do
  MOM_BENCH_SPAWN=1 python bench.py --shard $m $args > $out/dist_$m.json 2> $out/dist_$m.err; echo "$m rc $?"
  MOM_COMM=torch MOM_BENCH_SPAWN=1 python bench.py --shard $m $args > $out/dist_${m}_torch.json 2>/dev/null
done
MOM_BENCH_SPAWN=1 python bench.py --shard camera --shard-adam $args > $out/dist_camera_sharded.json 2> $out/dist_camera_sharded.err; echo "sharded rc $?"
python - <<PY
import json
rows = {}
for n in ("plain", "dist_camera", "dist_tile-row", "dist_camera_torch", "dist_tile-row_torch", "dist_camera_sharded"):
    try:
        d = json.load(open("$out/" + n + ".json"))
        rows[n] = {"steps_per_s": round(d["value"], 1), "parallelism": d["config"]["parallelism"], "ranks_seen": d["config"]["ranks_seen"]}
    except Exception as e:
        rows[n] = {"failed": str(e)}
json.dump(rows, open("$out/dist_one_rank.json", "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
