#!/bin/bash
V=iclr2025_3d-mom_amd/lib/var
for r in 1 2 3; do for v in q0 q1; do MOM4D_LIB=$PWD/$V/$v.so python tools/kbench.py render_fwd render_bwd 2>/dev/null | tail -1; done; done
for v in q0 q1; do KBENCH_CONFIG=c1 MOM4D_LIB=$PWD/$V/$v.so python tools/kbench.py render_fwd render_bwd 2>/dev/null | tail -1; done
MOM4D_LIB=$PWD/$V/q1.so python -m pytest tests/test_raster_gpu.py tests/test_tile_rows_gpu.py tests/test_golden_gpu.py tests/test_fused_step_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -5
