#!/bin/bash
# tools/probe/pk_bisect/run_variants.sh <launches> name1 name2 ...: the real-kernel reproducer (tools/probe/pk_hazard_real.py) on each variant library
cd ${GRAFT_REPO_ROOT:-.}
n=$1; shift
for v in "$@"; do
  MOM4D_LIB=$PWD/iclr2025_3d-mom_amd/lib/var/pk_$v.so MOM4D_LIB_LAX=1 timeout 300 python tools/probe/pk_hazard_real.py $n 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
pos = d['wrong_feature_elements_by_position_in_tile_of_32']
print('$v', 'wrong launches', d['wrong_launches'], 'of', d['launches'], '| wrong elements', sum(pos), '| positions in tile', [i for i, c in enumerate(pos) if c], '| mlp', d['launches_with_wrong_mlp_outputs'])
"
done
